// Plan-time parameter math (host only, double precision).
//
// Restates the scalar work of _PlanNUFFT (reference src/plan.jl:467-541) for the backwards
// Kaiser-Bessel kernel: oversampled size rule, shape parameter, piecewise-polynomial fit,
// Fourier coefficients, index map, plus the MI355X LDS tile search that replaces
// block_dims_gpu_shmem (src/gpu_common.jl:19-92).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>

#include "device_common.h"
#include "nufft_internal.h"

namespace nufft {

// Julia nextprod((2, 3, 5), n): smallest 2^a 3^b 5^c >= n (src/plan.jl:492-494).
int64_t nextprod235(int64_t n) {
    if (n < 1) n = 1;
    int64_t best = std::numeric_limits<int64_t>::max();
    for (int64_t p2 = 1;; p2 *= 2) {
        for (int64_t p23 = p2;; p23 *= 3) {
            int64_t p = p23;
            while (p < n) p *= 5;
            best = std::min(best, p);
            if (p23 >= n) break;
        }
        if (p2 >= n) break;
    }
    return best;
}

// src/plan.jl:485-498
int64_t oversampled_size(int64_t N, double sigma, bool real_first_dim) {
    if (real_first_dim) return 2 * nextprod235((int64_t)std::floor(sigma * (double)((N + 1) / 2)));
    return nextprod235((int64_t)std::floor(sigma * (double)N));
}

// src/Kernels/kaiser_bessel_backwards.jl:123-136
double bkb_beta(int M, double sigma_d) {
    const double a = M * (2.0 - 1.0 / sigma_d);
    const double gamma = std::max(0.995, std::sqrt(1.0 - 0.3 / (a * a)));
    return M_PI * a * gamma;
}

// src/Kernels/kaiser_bessel_backwards.jl:99-102
double bkb_function(double y, double beta) {
    const double z = 1.0 - y * y;
    const double s = std::sqrt(z > 0.0 ? z : 0.0);
    const double bs = beta * s;
    const double ratio = (bs == 0.0) ? 1.0 : std::sinh(bs) / bs;
    return ratio * (beta / M_PI);
}

// Modified Bessel function I0 (stands in for Bessels.besseli0, called at
// src/Kernels/kaiser_bessel_backwards.jl:143).  Power series: all terms positive, no cancellation.
double bessel_i0(double x) {
    const double q = 0.25 * x * x;
    double term = 1.0, sum = 1.0;
    for (int k = 1; k < 1000; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}

// KaiserBesselKernel, src/Kernels/kaiser_bessel.jl:151-165
double kb_beta(int M, double sigma_d) {
    const double a = M * (2.0 - 1.0 / sigma_d);
    const double gamma = std::sqrt(1.0 - 0.8 / (a * a));
    return M_PI * a * gamma;
}

// phi(y) = I0(beta sqrt(1 - y^2)), src/Kernels/kaiser_bessel.jl:128-130
double kb_function(double y, double beta) {
    const double z = 1.0 - y * y;
    return bessel_i0(beta * std::sqrt(z > 0.0 ? z : 0.0));
}

// GaussianKernel: ell / dx, src/Kernels/gaussian.jl:107-116
double gaussian_ell(int M, double sigma_d) {
    return std::sqrt(sigma_d * M / (2.0 * sigma_d - 1.0) / M_PI);
}

// Solves the (npoly x npoly) Vandermonde system by Gaussian elimination with partial pivoting
// (solve_polynomial_coefficients!, src/Kernels/piecewise_polynomial.jl:23-41).
static void solve_dense(std::vector<double>& A, std::vector<double>& b, int n) {
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r)
            if (std::fabs(A[r * n + col]) > std::fabs(A[piv * n + col])) piv = r;
        if (piv != col) {
            for (int c = 0; c < n; ++c) std::swap(A[col * n + c], A[piv * n + c]);
            std::swap(b[col], b[piv]);
        }
        const double d = A[col * n + col];
        for (int r = col + 1; r < n; ++r) {
            const double f = A[r * n + col] / d;
            if (f == 0.0) continue;
            for (int c = col; c < n; ++c) A[r * n + c] -= f * A[col * n + c];
            b[r] -= f * b[col];
        }
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = b[r];
        for (int c = r + 1; c < n; ++c) s -= A[r * n + c] * b[c];
        b[r] = s / A[r * n + r];
    }
}

// solve_piecewise_polynomial_coefficients, src/Kernels/piecewise_polynomial.jl:50-74 with
// Npoly = M + 4 (src/Kernels/kaiser_bessel_backwards.jl:98).
void poly_coefficients(int M, double beta, double (*f)(double, double), std::vector<double>& cs) {
    const int L = 2 * M;
    const int np = M + 4;
    cs.assign((size_t)np * L, 0.0);
    std::vector<double> xs(np), A((size_t)np * np), ys(np);
    for (int i = 1; i <= np; ++i) xs[i - 1] = std::cos(M_PI * ((double)i - 0.5) / (double)np);
    const double delta = 1.0 / (double)L;
    for (int j = 1; j <= L; ++j) {
        const double h = 1.0 - 2.0 * ((double)j - 0.5) / (double)L;
        for (int i = 0; i < np; ++i) {
            double pw = 1.0;
            for (int k = 0; k < np; ++k) {
                A[(size_t)i * np + k] = pw;
                pw *= xs[i];
            }
            ys[i] = f(h + xs[i] * delta, beta);
        }
        solve_dense(A, ys, np);
        for (int k = 0; k < np; ++k) cs[(size_t)k * L + (j - 1)] = ys[k];
    }
}

void bkb_poly_coefficients(int M, double beta, std::vector<double>& cs) { poly_coefficients(M, beta, bkb_function, cs); }
// Npoly = M + 4 as well, src/Kernels/kaiser_bessel.jl:127-130
void kb_poly_coefficients(int M, double beta, std::vector<double>& cs) { poly_coefficients(M, beta, kb_function, cs); }

// init_wavenumbers, src/plan.jl:558-566
void wavenumbers(int64_t N, bool r2c, std::vector<double>& ks) {
    if (r2c) {
        ks.resize((size_t)(N / 2 + 1));
        for (int64_t i = 0; i <= N / 2; ++i) ks[(size_t)i] = (double)i;
    } else {
        ks.resize((size_t)N);
        for (int64_t i = 0; i < N; ++i) ks[(size_t)i] = (double)(i >= (N + 1) / 2 ? i - N : i);
    }
}

// evaluate_fourier_func, src/Kernels/kaiser_bessel_backwards.jl:138-145
void fourier_coefficients(const std::vector<double>& ks, int M, int64_t Nover, double beta,
                          std::vector<double>& phihat) {
    const double w = M * (2.0 * M_PI / (double)Nover);
    phihat.resize(ks.size());
    for (size_t i = 0; i < ks.size(); ++i) {
        const double q = w * ks[i];
        const double s = std::sqrt(beta * beta - q * q);
        phihat[i] = w * bessel_i0(s);
    }
}

// evaluate_fourier_func of the other kernels: src/Kernels/kaiser_bessel.jl:167-174,
// src/Kernels/gaussian.jl:118-123 (param = tau), src/Kernels/bspline.jl:121-129.
void fourier_coefficients_kernel(int kernel, const std::vector<double>& ks, int M, int64_t Nover, double param,
                                 std::vector<double>& phihat) {
    if (kernel == NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL) { fourier_coefficients(ks, M, Nover, param, phihat); return; }
    const double dx = 2.0 * M_PI / (double)Nover;
    phihat.resize(ks.size());
    for (size_t i = 0; i < ks.size(); ++i) {
        const double k = ks[i];
        if (kernel == NUFFT_KERNEL_KAISER_BESSEL) {
            const double w = M * dx, q = w * k;
            const double s = std::sqrt(param * param - q * q);
            phihat[i] = 2.0 * w * std::sinh(s) / s;
        } else if (kernel == NUFFT_KERNEL_GAUSSIAN) {
            phihat[i] = std::exp(-param * k * k / 4.0) * std::sqrt(M_PI * param);
        } else {
            const double kh = k * dx / 2.0;
            phihat[i] = (k == 0.0 ? 1.0 : std::pow(std::sin(kh) / kh, 2 * M)) * dx;
        }
    }
}

// non_oversampled_indices!, src/NonuniformFFTs.jl:318-348 (0-based result)
void non_oversampled_indices(const std::vector<double>& ks, int64_t n_axis, bool fftshift,
                             std::vector<int64_t>& indmap) {
    const int64_t Nk = (int64_t)ks.size();
    indmap.resize((size_t)Nk);
    const bool r2c = ks.back() > 0;
    if (r2c) {
        for (int64_t i = 0; i < Nk; ++i) indmap[(size_t)i] = i;
    } else if (Nk % 2 == 0) {
        const int64_t h = Nk / 2;
        for (int64_t i = 0; i < h; ++i) {
            if (fftshift) {
                indmap[(size_t)i] = n_axis - h + i;
                indmap[(size_t)(h + i)] = i;
            } else {
                indmap[(size_t)i] = i;
                indmap[(size_t)(h + i)] = n_axis - h + i;
            }
        }
    } else {
        const int64_t h = (Nk - 1) / 2;
        if (fftshift) {
            for (int64_t i = 0; i < h; ++i) indmap[(size_t)i] = n_axis - h + i;
            for (int64_t i = 0; i <= h; ++i) indmap[(size_t)(h + i)] = i;
        } else {
            for (int64_t i = 0; i <= h; ++i) indmap[(size_t)i] = i;
            for (int64_t i = 0; i < h; ++i) indmap[(size_t)(h + 1 + i)] = n_axis - h + i;
        }
    }
}

// Candidate tile edges along one axis: multiples of the bin size that leave room for the clipped halo
// (n + 2M - 1 <= Ñ, so that a point has at most one periodic image next to the tile), or the whole axis.
static void edge_candidates(int64_t N, int b, int M, int cap, std::vector<int>& out) {
    out.clear();
    for (int n = b; n < N && n <= cap; n += b)
        if (n + 2 * M - 1 <= N) out.push_back(n);
    if (N <= cap || out.empty()) out.push_back((int)N);      // single tile spanning the axis (wrap mode)
}

// LDS row stride (in Float64 reals) of the spreading tile.  A wave instruction of the accumulation
// touches consecutive rows with `stencil_inner` contiguous reals each; the rows land on disjoint banks
// when the stride is congruent to the stencil width modulo the 128-byte bank period.
int lds_row_stride(int inner_elems, int stencil_inner, int real_bytes) {
    const bool no_pad = option_present("NUFFT_LDS_NO_PAD");
    return no_pad ? inner_elems : padded_row_stride(inner_elems, stencil_inner, real_bytes);
}

static void fill_shape(TileShapeHost& t, int D, int M, int ncomp, int real_bytes, const int64_t* Nover, const int n[3], bool padded, int bin_log2) {
    const int halo = padded ? 2 * M - 1 : 0;
    int P[3] = {1, 1, 1};
    t.ntiles = 1;
    for (int d = 0; d < 3; ++d) {
        t.n[d] = d < D ? n[d] : 1;
        t.nt[d] = d < D ? (int)((Nover[d] + n[d] - 1) / n[d]) : 1;
        P[d] = d < D ? n[d] + halo : 1;
        t.ntiles *= t.nt[d];
    }
    t.row_stride = (!padded && D >= 2) ? lds_row_stride(ncomp * P[0], ncomp * 2 * M, 8) : ncomp * P[0];
    t.rows[0] = P[1];
    t.rows[1] = P[2];
    t.plane_stride = padded ? t.row_stride * P[1] : spread_plane_stride(t.row_stride, P[1], D, ncomp);
    t.elems = (int64_t)t.plane_stride * P[2];
    // upper bound of the contiguous runs of sorted points a tile works through (the kernels' own arithmetic)
    const int b = 1 << bin_log2;
    int nbv[3] = {1, 1, 1};
    for (int d = 0; d < D; ++d) nbv[d] = (int)((Nover[d] + b - 1) / b);
    t.max_items = (int)tile_items_bound(!padded, D, M, b, t.n, nbv);
}

// Exact maximum, over all tile positions, of the runs of the sorted array a tile looks up — the arithmetic of the kernels
// themselves (bin_segments / the bin box of an interpolation tile, tile_kernels.h) replayed on the host for every tile
// coordinate of every axis (the count is a product over the axes, so the maximum is the product of the per-axis maxima).
// A plan whose table (tile_items_bound + kItemTarget) could not hold it is refused at creation; the kernels never see one.
static int host_bin_rows(int lo, int hi, int N, int blog, int nb, int* pieces) {
    *pieces = 1;
    if (hi - lo >= N) return nb;
    if (lo >= 0 && hi <= N) return ((hi - 1) >> blog) - (lo >> blog) + 1;
    const int lo2 = lo < 0 ? lo + N : lo, hi2 = lo < 0 ? hi : hi - N;
    const int a_first = lo2 >> blog, b_last = (hi2 - 1) >> blog;
    if (b_last + 1 >= a_first) return nb;
    *pieces = 2;
    return (nb - a_first) + (b_last + 1);
}
long exact_tile_runs(bool spreading, int D, int M, const int64_t* Nover, int bin_log2, const TileShapeHost& t) {
    long runs = 1;
    for (int d = 0; d < D; ++d) {
        const int N = (int)Nover[d], nb = (N + (1 << bin_log2) - 1) >> bin_log2;
        int worst = 0;
        for (int k = 0; k < t.nt[d]; ++k) {
            const int org = k * t.n[d], neff = std::min(t.n[d], N - org);
            int pieces = 1, rows;
            if (spreading) rows = host_bin_rows(org - M, org + neff + M - 1, N, bin_log2, nb, &pieces);
            else rows = ((org + neff - 1) >> bin_log2) - (org >> bin_log2) + 1;
            // dimension 1 contributes its pieces (one run per piece), the others their bin rows
            worst = std::max(worst, d == 0 ? (spreading ? pieces : 1) : rows);
        }
        runs *= worst;
    }
    return runs;
}

// exact LDS bytes of a candidate tile (tile + work-item table + window strips); -1: too many work items
static int64_t candidate_lds(bool spreading, int D, int M, int ncomp, int real_bytes, int nwaves, int64_t elems,
                             const int n[3], const int64_t* Nover, int bin_log2) {
    const int b = 1 << bin_log2;
    int nbv[3] = {1, 1, 1};
    for (int d = 0; d < D; ++d) nbv[d] = (int)((Nover[d] + b - 1) / b);
    const int nn[3] = {n[0], n[1], n[2]};
    const long items = tile_items_bound(spreading, D, M, b, nn, nbv);
    if (items > kMaxTileItems || elems > (int64_t)1 << 24) return -1;
    return lds_layout((int)elems, spreading ? 8 : real_bytes, real_bytes, D, M, ncomp, nwaves, (int)items,
                      spreading ? spread_strip_pad(D, ncomp) : 0).total;
}

bool choose_tiles(int D, int M, int ncomp, int real_bytes, const int64_t* Nover, int lds_budget_bytes,
                  int spread_waves, int interp_waves, const int* forced_sp, const int* forced_ip, int bin_log2,
                  TileGeom& g) {
    const int halo = 2 * M - 1;
    g.nbins = 1;
    for (int d = 0; d < 3; ++d) {
        g.blog[d] = d < D ? bin_log2 : 0;
        g.nb[d] = d < D ? (int)((Nover[d] + (1 << bin_log2) - 1) >> bin_log2) : 1;
        g.nbins *= g.nb[d];
    }
    const int b = 1 << bin_log2;
    std::vector<int> cand[3];
    for (int d = 0; d < 3; ++d) {
        if (d < D) edge_candidates(Nover[d], b, M, D == 1 ? 8192 : 96, cand[d]);
        else cand[d].assign(1, 1);
    }
    // --- spreading tile: interior only, Float64 accumulation; cost = point visits per point ---
    {
        auto fits = [&](int64_t elems, const int n[3]) {
            const int64_t tot = candidate_lds(true, D, M, ncomp, real_bytes, spread_waves, elems, n, Nover, bin_log2);
            return tot >= 0 && tot <= lds_budget_bytes;
        };
        double best = std::numeric_limits<double>::infinity();
        int bn[3] = {0, 0, 0};
        if (forced_sp && forced_sp[0] > 0) {
            for (int d = 0; d < 3; ++d) {
                int n = d < D ? (forced_sp[d] > 0 ? forced_sp[d] : forced_sp[0]) : 1;
                if (d < D) {
                    n = std::max(b, n / b * b);
                    if (n + halo > Nover[d]) n = (int)Nover[d];
                }
                bn[d] = n;
            }
            if (!fits((int64_t)spread_plane_stride(D >= 2 ? lds_row_stride(ncomp * bn[0], ncomp * 2 * M, 8) : ncomp * bn[0], bn[1], D, ncomp) * bn[2], bn)) return false;
        } else {
            for (int n3 : cand[2]) for (int n2 : cand[1]) for (int n1 : cand[0]) {
                const int64_t elems = (int64_t)spread_plane_stride(D >= 2 ? lds_row_stride(ncomp * n1, ncomp * 2 * M, 8) : ncomp * n1, n2, D, ncomp) * n3;
                const int n[3] = {n1, n2, n3};
                if (!fits(elems, n)) continue;
                double cost = 1.0;
                for (int d = 0; d < D; ++d)
                    if (n[d] < Nover[d]) cost *= (double)(n[d] + halo) / n[d];
                // every tile also pays a fixed cost (zeroing + storing the LDS tile): prefer enough work per tile
                cost -= 1e-6 * n1;
                if (cost < best) { best = cost; bn[0] = n1; bn[1] = n2; bn[2] = n3; }
            }
            if (bn[0] == 0) return false;
        }
        fill_shape(g.sp, D, M, ncomp, real_bytes, Nover, bn, false, bin_log2);
        if (exact_tile_runs(true, D, M, Nover, bin_log2, g.sp) > g.sp.max_items) return false;
    }
    // --- interpolation tile: padded, grid precision; cost = halo amplification of the tile load ---
    {
        auto fits = [&](int64_t elems, const int n[3]) {
            const int64_t tot = candidate_lds(false, D, M, ncomp, real_bytes, interp_waves, elems, n, Nover, bin_log2);
            return tot >= 0 && tot <= lds_budget_bytes;
        };
        double best = std::numeric_limits<double>::infinity();
        int bn[3] = {0, 0, 0};
        if (forced_ip && forced_ip[0] > 0) {
            for (int d = 0; d < 3; ++d) {
                int n = d < D ? (forced_ip[d] > 0 ? forced_ip[d] : forced_ip[0]) : 1;
                if (d < D) {
                    n = std::max(b, n / b * b);
                    if (n >= Nover[d]) n = (int)Nover[d];
                }
                bn[d] = n;
            }
            int64_t e = ncomp;
            for (int d = 0; d < D; ++d) e *= bn[d] + halo;
            if (!fits(e, bn)) return false;
        } else {
            std::vector<int> ic[3];
            for (int d = 0; d < 3; ++d) {
                ic[d].clear();
                if (d >= D) { ic[d].push_back(1); continue; }
                for (int n = b; n < Nover[d] && n <= (D == 1 ? 8192 : 96); n += b) ic[d].push_back(n);
                if (Nover[d] <= (D == 1 ? 8192 : 96) || ic[d].empty()) ic[d].push_back((int)Nover[d]);
            }
            for (int n3 : ic[2]) for (int n2 : ic[1]) for (int n1 : ic[0]) {
                const int n[3] = {n1, n2, n3};
                int64_t elems = ncomp;
                double cost = 1.0;
                for (int d = 0; d < D; ++d) {
                    elems *= n[d] + halo;
                    cost *= (double)(n[d] + halo) / n[d];
                }
                if (!fits(elems, n)) continue;
                cost -= 1e-6 * n1;
                if (cost < best) { best = cost; bn[0] = n1; bn[1] = n2; bn[2] = n3; }
            }
            if (bn[0] == 0) return false;
        }
        fill_shape(g.ip, D, M, ncomp, real_bytes, Nover, bn, true, bin_log2);
        if (exact_tile_runs(false, D, M, Nover, bin_log2, g.ip) > g.ip.max_items) return false;
    }
    return true;
}

}  // namespace nufft
