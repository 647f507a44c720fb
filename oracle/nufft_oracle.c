/*
 * nufft_oracle.c — plain-C restatement of the reference's *blocked CPU* spreading and
 * interpolation (Float64), used (a) to cross-check the numpy oracle at sizes numpy cannot reach
 * and (b) as the timed "port" CPU baseline of bench.py.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle/nufft_oracle.py for the pinning
 * statement).  Nothing under nonuniformffts.jl_amd/ links or loads it.
 *
 * Reference files restated (relative to /root/reference):
 *   src/blocking/cpu.jl:73-185        counting sort of the points by block (assign_blocks_cpu!,
 *                                     sortperm_cpu!); block = (16,16,16) in 3-D (src/plan.jl:437-449
 *                                     with default_block_size = 4096, src/NonuniformFFTs.jl:58),
 *                                     clamped to N - M (src/blocking/cpu.jl:49-51)
 *   src/spreading/cpu_blocked.jl:94-168   per-thread padded block buffer (block_dims + 2M), points
 *                                     spread without wrapping (:38-64,:285-327), then the block is
 *                                     added to the global array with periodic wrap (:170-266); the
 *                                     merge uses atomics (use_atomics = true variant, :150-163)
 *   src/interpolation/cpu_blocked.jl:95-206  copy padded block (copy_to_block!), gather per point
 *   src/Kernels/Kernels.jl:121-126    point_to_cell: r = (x / 2π) * N
 *   src/blocking/blocking.jl:12-21    to_unit_cell_cpu (while loops)
 *   src/Kernels/kaiser_bessel_backwards.jl:147-175, src/Kernels/piecewise_polynomial.jl:76-92
 *                                     window evaluation (FastApproximation = Horner, Direct = sinh)
 *
 * Arrays are column-major ("dimension 1 fastest") exactly as in the reference.
 * Build: see oracle/Makefile (gcc -O3 -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXM 16
#define TWO_PI 6.28318530717958647692

typedef struct {
    int D, M, evalmode, ncomp;       /* ncomp: 1 real values, 2 complex (interleaved re,im) */
    int64_t N[3];                    /* oversampled grid */
    const double* coefs;             /* [D][M+4][2M] */
    double beta[3];
} oracle_geom;

/* Coordinate arithmetic of a Float32 plan (oracle_set_coord_f32(1); OraclePlan.coord_dtype in nufft_oracle.py): the
 * points (exactly representable Float32 numbers) are folded and located with Float32 operations, as the reference does
 * for T = Float32 (src/blocking/blocking.jl:12-21, src/Kernels/Kernels.jl:121-126 evaluated in T); windows and sums stay
 * Float64. */
static int g_coord_f32 = 0;
void oracle_set_coord_f32(int on) { g_coord_f32 = on; }

/* to_unit_cell_cpu, src/blocking/blocking.jl:12-21 */
static inline double fold(double x) {
    if (g_coord_f32) {
        volatile float xf = (float)x;
        const float L = (float)TWO_PI;
        while (xf < 0) xf += L;
        while (xf >= L) xf -= L;
        return (double)xf;
    }
    while (x < 0) x += TWO_PI;
    while (x >= TWO_PI) x -= TWO_PI;
    return x;
}

/* point_to_cell (0-based), src/Kernels/Kernels.jl:121-126 */
static inline int64_t cell_of(double xf, int64_t N, double* r) {
    if (g_coord_f32) {
        volatile float q = (float)xf / (float)TWO_PI;
        volatile float rf = q * (float)N;
        int64_t i = (int64_t)rf;
        if (i >= N) i = N - 1;
        volatile float X = rf - (float)i;          /* the cell fraction is formed in Float32 too */
        *r = (double)i + (double)X;
        return i;
    }
    *r = (xf / TWO_PI) * (double)N;
    int64_t i = (int64_t)(*r);
    if (i >= N) i = N - 1;
    return i;
}

/* 2M window values for cell fraction X; vals[j] belongs to grid node i - M + 1 + j (0-based). */
static inline void window(const oracle_geom* g, int d, double X, double* vals) {
    const int M = g->M, L = 2 * M;
    if (g->evalmode == 0) { /* Direct, kaiser_bessel_backwards.jl:158-175 */
        const double beta = g->beta[d];
        for (int j = 0; j < L; ++j) {
            const double y = ((double)(M - 1 - j) + X) / (double)M;
            const double z = 1.0 - y * y;
            const double s = sqrt(z > 0 ? z : 0);
            const double bs = beta * s;
            vals[j] = (s == 0.0 ? 1.0 : sinh(bs) / bs) * (beta / M_PI);
        }
    } else { /* FastApproximation, piecewise_polynomial.jl:76-92 */
        const int np = M + 4;
        const double* cs = g->coefs + (size_t)d * np * L;
        const double x = 2.0 * X - 1.0;
        for (int j = 0; j < L; ++j) vals[j] = cs[(size_t)(np - 1) * L + j];
        for (int k = np - 2; k >= 0; --k)
            for (int j = 0; j < L; ++j) vals[j] = x * vals[j] + cs[(size_t)k * L + j];
    }
}

typedef struct {
    int64_t nb[3], bd[3];   /* blocks per dim, block dims */
    int64_t nblocks;
    int64_t* offsets;       /* [nblocks + 1] */
    int64_t* perm;          /* [Np] */
} blocking;

static void default_block_dims(int D, const int64_t* N, int M, int64_t* bd) {
    /* get_block_dims(Ñs, 4096): powers of two round-robin, src/plan.jl:437-449; clamp, cpu.jl:49-51 */
    for (int d = 0; d < 3; ++d) bd[d] = 1;
    int64_t prod = 1;
    int i = 0;
    while (prod < 4096) {
        bd[i] <<= 1;
        prod <<= 1;
        i = (i + 1 == D) ? 0 : i + 1;
    }
    for (int d = 0; d < D; ++d) {
        int64_t lim = N[d] - M;
        if (lim < 1) lim = 1;
        if (bd[d] > lim) bd[d] = lim;
    }
}

static void build_blocking(const oracle_geom* g, int64_t Np, const double* const* x, blocking* b) {
    default_block_dims(g->D, g->N, g->M, b->bd);
    b->nblocks = 1;
    for (int d = 0; d < 3; ++d) {
        b->nb[d] = d < g->D ? (g->N[d] + b->bd[d] - 1) / b->bd[d] : 1;
        b->nblocks *= b->nb[d];
    }
    b->offsets = (int64_t*)calloc((size_t)b->nblocks + 1, sizeof(int64_t));
    b->perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)(Np > 0 ? Np : 1));
    int64_t* blk = (int64_t*)malloc(sizeof(int64_t) * (size_t)(Np > 0 ? Np : 1));
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < Np; ++p) {
        int64_t lin = 0, mul = 1;
        for (int d = 0; d < g->D; ++d) {
            double r;
            const int64_t i = cell_of(fold(x[d][p]), g->N[d], &r);
            lin += mul * (i / b->bd[d]);
            mul *= b->nb[d];
        }
        blk[p] = lin;
    }
    for (int64_t p = 0; p < Np; ++p) b->offsets[blk[p] + 1]++;
    for (int64_t k = 0; k < b->nblocks; ++k) b->offsets[k + 1] += b->offsets[k];
    int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)b->nblocks);
    memcpy(cur, b->offsets, sizeof(int64_t) * (size_t)b->nblocks);
    for (int64_t p = 0; p < Np; ++p) b->perm[cur[blk[p]]++] = p;
    free(cur);
    free(blk);
}

static void free_blocking(blocking* b) {
    free(b->offsets);
    free(b->perm);
}

static inline int64_t wrap(int64_t i, int64_t N) {
    while (i < 0) i += N;
    while (i >= N) i -= N;
    return i;
}

/* Type-1 spreading of C components onto zeroed grids u[c] (each N1*N2*N3*ncomp doubles). */
int oracle_spread_blocked(int D, const int64_t* N, int M, int evalmode, int ncomp, const double* coefs,
                          const double* betas, int64_t Np, const double* const* x, int C,
                          const double* const* v, double* const* u) {
    if (M > MAXM || D < 1 || D > 3) return 1;
    oracle_geom g;
    g.D = D; g.M = M; g.evalmode = evalmode; g.ncomp = ncomp; g.coefs = coefs;
    for (int d = 0; d < 3; ++d) { g.N[d] = d < D ? N[d] : 1; g.beta[d] = d < D ? betas[d] : 0; }
    blocking b;
    build_blocking(&g, Np, x, &b);
    const int L = 2 * M;
    int64_t P[3];
    for (int d = 0; d < 3; ++d) P[d] = d < D ? b.bd[d] + L : 1;   /* padded block, cpu.jl:54 */
    const size_t pelems = (size_t)(P[0] * P[1] * P[2]) * ncomp;
    const char* ua = getenv("ORACLE_USE_ATOMICS");
    const int use_atomics = ua && ua[0] == '1';
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * pelems);
#pragma omp for schedule(dynamic, 1)
        for (int64_t blk = 0; blk < b.nblocks; ++blk) {
            const int64_t pa = b.offsets[blk], pb = b.offsets[blk + 1];
            if (pa == pb) continue;
            int64_t t[3], rem = blk;
            for (int d = 0; d < 3; ++d) { t[d] = rem % b.nb[d]; rem /= b.nb[d]; }
            for (int c = 0; c < C; ++c) {
                memset(buf, 0, sizeof(double) * pelems);
                for (int64_t q = pa; q < pb; ++q) {
                    const int64_t p = b.perm[q];
                    double w[3][2 * MAXM];
                    int64_t s[3] = {0, 0, 0};
                    for (int d = 0; d < D; ++d) {
                        double r;
                        const int64_t i = cell_of(fold(x[d][p]), g.N[d], &r);
                        window(&g, d, r - (double)i, w[d]);
                        s[d] = i - t[d] * b.bd[d];   /* local cell; stencil occupies s+1 .. s+2M in the padded block */
                    }
                    const int n3 = D >= 3 ? L : 1, n2 = D >= 2 ? L : 1;
                    for (int j3 = 0; j3 < n3; ++j3) {
                        const double w3 = D >= 3 ? w[2][j3] : 1.0;
                        for (int j2 = 0; j2 < n2; ++j2) {
                            const double w23 = (D >= 2 ? w[1][j2] : 1.0) * w3;
                            double* row = buf + (((size_t)(s[2] + (D >= 3 ? j3 + 1 : 0)) * P[1] + (s[1] + (D >= 2 ? j2 + 1 : 0))) * P[0] + s[0] + 1) * ncomp;
                            if (ncomp == 1) {
                                const double vv = v[c][p];
                                for (int j1 = 0; j1 < L; ++j1) row[j1] += vv * (w23 * w[0][j1]);
                            } else {
                                const double vr = v[c][2 * p], vi = v[c][2 * p + 1];
                                for (int j1 = 0; j1 < L; ++j1) {
                                    const double ww = w23 * w[0][j1];
                                    row[2 * j1] += vr * ww;
                                    row[2 * j1 + 1] += vi * ww;
                                }
                            }
                        }
                    }
                }
                /* add_from_block! with periodic wrap (cpu_blocked.jl:170-266).  Default of the reference
                 * (cpu_use_atomics = false, :156-163): plain adds under one lock; ORACLE_USE_ATOMICS=1 selects
                 * the atomics variant instead. */
                int64_t o[3];
                for (int d = 0; d < 3; ++d) o[d] = t[d] * b.bd[d] - M;   /* local index l (0-based) -> global o + l */
                if (use_atomics) {
                    for (int64_t l3 = 0; l3 < P[2]; ++l3) {
                        const int64_t g3 = D >= 3 ? wrap(o[2] + l3, g.N[2]) : 0;
                        for (int64_t l2 = 0; l2 < P[1]; ++l2) {
                            const int64_t g2 = D >= 2 ? wrap(o[1] + l2, g.N[1]) : 0;
                            const double* src = buf + ((size_t)(l3 * P[1] + l2) * P[0]) * ncomp;
                            double* dstrow = u[c] + ((size_t)(g3 * g.N[1] + g2) * g.N[0]) * ncomp;
                            for (int64_t l1 = 0; l1 < P[0]; ++l1) {
                                const int64_t g1 = wrap(o[0] + l1, g.N[0]);
                                for (int k = 0; k < ncomp; ++k) {
                                    const double val = src[l1 * ncomp + k];
                                    if (val != 0.0) {
#pragma omp atomic
                                        dstrow[g1 * ncomp + k] += val;
                                    }
                                }
                            }
                        }
                    }
                } else {
#pragma omp critical(oracle_merge)
                    {
                        for (int64_t l3 = 0; l3 < P[2]; ++l3) {
                            const int64_t g3 = D >= 3 ? wrap(o[2] + l3, g.N[2]) : 0;
                            for (int64_t l2 = 0; l2 < P[1]; ++l2) {
                                const int64_t g2 = D >= 2 ? wrap(o[1] + l2, g.N[1]) : 0;
                                const double* src = buf + ((size_t)(l3 * P[1] + l2) * P[0]) * ncomp;
                                double* dstrow = u[c] + ((size_t)(g3 * g.N[1] + g2) * g.N[0]) * ncomp;
                                /* contiguous pieces of the row between wrap points */
                                int64_t l1 = 0;
                                while (l1 < P[0]) {
                                    const int64_t g1 = wrap(o[0] + l1, g.N[0]);
                                    int64_t len = g.N[0] - g1;
                                    if (len > P[0] - l1) len = P[0] - l1;
                                    double* dst = dstrow + g1 * ncomp;
                                    const double* sp = src + l1 * ncomp;
                                    for (int64_t e = 0; e < len * ncomp; ++e) dst[e] += sp[e];
                                    l1 += len;
                                }
                            }
                        }
                    }
                }
            }
        }
        free(buf);
    }
    free_blocking(&b);
    return 0;
}

/* Type-2 interpolation from C grids u[c] to values v[c] (without the Δx prefactor applied per
 * dimension in the reference: cpu_nonblocked.jl:45-48 multiplies each 1-D window by Δx_d; here the
 * product is applied once at the end, as src/interpolation/gpu.jl:55-56 does). */
int oracle_interp_blocked(int D, const int64_t* N, int M, int evalmode, int ncomp, const double* coefs,
                          const double* betas, int64_t Np, const double* const* x, int C,
                          const double* const* u, double* const* v) {
    if (M > MAXM || D < 1 || D > 3) return 1;
    oracle_geom g;
    g.D = D; g.M = M; g.evalmode = evalmode; g.ncomp = ncomp; g.coefs = coefs;
    for (int d = 0; d < 3; ++d) { g.N[d] = d < D ? N[d] : 1; g.beta[d] = d < D ? betas[d] : 0; }
    blocking b;
    build_blocking(&g, Np, x, &b);
    const int L = 2 * M;
    int64_t P[3];
    for (int d = 0; d < 3; ++d) P[d] = d < D ? b.bd[d] + L : 1;
    const size_t pelems = (size_t)(P[0] * P[1] * P[2]) * ncomp;
    double prefactor = 1.0;
    for (int d = 0; d < D; ++d) prefactor *= TWO_PI / (double)g.N[d];
#pragma omp parallel
    {
        double* buf = (double*)malloc(sizeof(double) * pelems);
#pragma omp for schedule(dynamic, 1)
        for (int64_t blk = 0; blk < b.nblocks; ++blk) {
            const int64_t pa = b.offsets[blk], pb = b.offsets[blk + 1];
            if (pa == pb) continue;
            int64_t t[3], rem = blk;
            for (int d = 0; d < 3; ++d) { t[d] = rem % b.nb[d]; rem /= b.nb[d]; }
            int64_t o[3];
            for (int d = 0; d < 3; ++d) o[d] = t[d] * b.bd[d] - M;
            for (int c = 0; c < C; ++c) {
                /* copy_to_block!, interpolation/cpu_blocked.jl:156-206 */
                for (int64_t l3 = 0; l3 < P[2]; ++l3) {
                    const int64_t g3 = D >= 3 ? wrap(o[2] + l3, g.N[2]) : 0;
                    for (int64_t l2 = 0; l2 < P[1]; ++l2) {
                        const int64_t g2 = D >= 2 ? wrap(o[1] + l2, g.N[1]) : 0;
                        double* dst = buf + ((size_t)(l3 * P[1] + l2) * P[0]) * ncomp;
                        const double* srcrow = u[c] + ((size_t)(g3 * g.N[1] + g2) * g.N[0]) * ncomp;
                        for (int64_t l1 = 0; l1 < P[0]; ++l1) {
                            const int64_t g1 = wrap(o[0] + l1, g.N[0]);
                            for (int k = 0; k < ncomp; ++k) dst[l1 * ncomp + k] = srcrow[g1 * ncomp + k];
                        }
                    }
                }
                for (int64_t q = pa; q < pb; ++q) {
                    const int64_t p = b.perm[q];
                    double w[3][2 * MAXM];
                    int64_t s[3] = {0, 0, 0};
                    for (int d = 0; d < D; ++d) {
                        double r;
                        const int64_t i = cell_of(fold(x[d][p]), g.N[d], &r);
                        window(&g, d, r - (double)i, w[d]);
                        s[d] = i - t[d] * b.bd[d];
                    }
                    double accr = 0.0, acci = 0.0;
                    const int n3 = D >= 3 ? L : 1, n2 = D >= 2 ? L : 1;
                    for (int j3 = 0; j3 < n3; ++j3) {
                        const double w3 = D >= 3 ? w[2][j3] : 1.0;
                        for (int j2 = 0; j2 < n2; ++j2) {
                            const double w23 = (D >= 2 ? w[1][j2] : 1.0) * w3;
                            const double* row = buf + (((size_t)(s[2] + (D >= 3 ? j3 + 1 : 0)) * P[1] + (s[1] + (D >= 2 ? j2 + 1 : 0))) * P[0] + s[0] + 1) * ncomp;
                            if (ncomp == 1) {
                                for (int j1 = 0; j1 < L; ++j1) accr += row[j1] * (w23 * w[0][j1]);
                            } else {
                                for (int j1 = 0; j1 < L; ++j1) {
                                    const double ww = w23 * w[0][j1];
                                    accr += row[2 * j1] * ww;
                                    acci += row[2 * j1 + 1] * ww;
                                }
                            }
                        }
                    }
                    if (ncomp == 1) v[c][p] = accr * prefactor;
                    else { v[c][2 * p] = accr * prefactor; v[c][2 * p + 1] = acci * prefactor; }
                }
            }
        }
        free(buf);
    }
    free_blocking(&b);
    return 0;
}

/* Type-1 epilogue: out[k3][k2][k1] = uhat[i3[k3]][i2[k2]][i1[k1]] * norm / (phihat1[k1] phihat2[k2] phihat3[k3]) — truncation to the
 * requested modes + deconvolution + normalisation in one threaded pass (exec_type1!, src/NonuniformFFTs.jl:372-379,394-401: the
 * reference does the same copy per output index inside its threaded loop).  Arrays in C order with dimension 1 fastest;
 * ns[d] = size of the oversampled spectrum, no[d] = modes kept, idx[d][k] = source index, inv[d][k] = 1 / phihat_d[k]. */
int oracle_deconv_truncate(int D, const int64_t* ns, const int64_t* no, const int64_t* const* idx, const double* const* inv,
                           double norm, const double* uhat, double* out) {
    const int64_t n1 = no[0], n2 = D >= 2 ? no[1] : 1, n3 = D >= 3 ? no[2] : 1;
    const int64_t s1 = ns[0], s2 = D >= 2 ? ns[1] : 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t k3 = 0; k3 < n3; ++k3) {
        for (int64_t k2 = 0; k2 < n2; ++k2) {
            const int64_t i3 = D >= 3 ? idx[2][k3] : 0, i2 = D >= 2 ? idx[1][k2] : 0;
            const double f23 = norm * (D >= 3 ? inv[2][k3] : 1.0) * (D >= 2 ? inv[1][k2] : 1.0);
            const double* src = uhat + 2 * ((i3 * s2 + i2) * s1);
            double* dst = out + 2 * ((k3 * n2 + k2) * n1);
            for (int64_t k1 = 0; k1 < n1; ++k1) {
                const double f = f23 * inv[0][k1];
                const int64_t i1 = idx[0][k1];
                dst[2 * k1] = src[2 * i1] * f;
                dst[2 * k1 + 1] = src[2 * i1 + 1] * f;
            }
        }
    }
    return 0;
}

/* threaded zero fill of a plan-owned work array (the reference's fill!(us, 0) at the start of exec_type1!, src/NonuniformFFTs.jl:161-167) */
void oracle_zero(double* p, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) p[i] = 0.0;
}

void oracle_set_num_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
