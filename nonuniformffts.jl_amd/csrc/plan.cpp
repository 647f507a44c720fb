// C ABI of libnufft_mi355x.so: plan lifetime, set_points, exec_type1 / exec_type2 and the
// stage-level entry points.  See include/nufft_mi355x.h for the reference functions each entry
// point replaces.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "kernels.h"
#include "nufft_internal.h"

namespace nufft {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

static int fail(int code, const std::string& msg) {
    set_error(msg);
    return code;
}

#define NUFFT_HIP(expr)                                                                        \
    do {                                                                                       \
        hipError_t e__ = (expr);                                                               \
        if (e__ != hipSuccess)                                                                 \
            return nufft::fail(e__ == hipErrorOutOfMemory ? NUFFT_ERR_ALLOC : NUFFT_ERR_HIP,   \
                               std::string(#expr) + ": " + hipGetErrorString(e__));            \
    } while (0)

#define NUFFT_ROCFFT(expr)                                                                     \
    do {                                                                                       \
        rocfft_status s__ = (expr);                                                            \
        if (s__ != rocfft_status_success)                                                      \
            return nufft::fail(NUFFT_ERR_ROCFFT, std::string(#expr) + ": rocfft status " + std::to_string((int)s__)); \
    } while (0)

static std::once_flag g_rocfft_once;

static size_t real_bytes(const nufft_plan* p) { return p->dtype == NUFFT_F32 ? 4 : 8; }
static size_t value_bytes(const nufft_plan* p) { return real_bytes(p) * (p->is_complex ? 2 : 1); }

static TileShape make_shape(const nufft::TileShapeHost& h) {
    TileShape t{};
    for (int d = 0; d < 3; ++d) {
        t.n[d] = h.n[d];
        t.nt[d] = h.nt[d];
    }
    t.row_stride = h.row_stride;
    t.plane_stride = h.plane_stride;
    t.elems = (int)h.elems;
    t.ntiles = (int)h.ntiles;
    t.max_items = h.max_items;
    return t;
}

static Geom make_geom(const nufft_plan* p) {
    Geom g{};
    for (int d = 0; d < 3; ++d) {
        g.Nover[d] = (int)p->Nover[d];
        g.blog[d] = p->tile.blog[d];
        g.nb[d] = p->tile.nb[d];
    }
    g.nbins = (int)p->tile.nbins;
    g.sp = make_shape(p->tile.sp);
    g.ip = make_shape(p->tile.ip);
    return g;
}

struct DeviceGuard {
    int prev = -1;
    bool active = false;
    explicit DeviceGuard(int dev) {
        if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) {
            active = hipSetDevice(dev) == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (active) (void)hipSetDevice(prev);
    }
};

// Every device allocation of a plan goes through here: workspace_bytes and the per-buffer breakdown (nufft_workspace_breakdown) are
// what the registry holds, so they cannot drift apart.  `name`: the row of DESIGN.md section 3 the buffer belongs to.
static int dev_alloc(nufft_plan* p, void** ptr, size_t bytes, const char* name = "tables") {
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(ptr, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        return fail(NUFFT_ERR_ALLOC, std::string("hipMalloc(") + std::to_string(bytes) + ", " + name + "): " + hipGetErrorString(e));
    }
    p->allocs[*ptr] = {name, (int64_t)bytes};
    p->workspace_bytes += (int64_t)bytes;
    return NUFFT_OK;
}
template <typename P>
static void dev_free(nufft_plan* p, P*& ptr) {
    if (!ptr) return;
    auto it = p->allocs.find(static_cast<void*>(ptr));
    if (it != p->allocs.end()) { p->workspace_bytes -= it->second.second; p->allocs.erase(it); }
    (void)hipFree(ptr);
    ptr = nullptr;
}

template <typename T>
static int upload(nufft_plan* p, void** dst, const std::vector<double>& src) {
    std::vector<T> tmp(src.size());
    for (size_t i = 0; i < src.size(); ++i) tmp[i] = (T)src[i];
    int rc = dev_alloc(p, dst, tmp.size() * sizeof(T));
    if (rc) return rc;
    NUFFT_HIP(hipMemcpy(*dst, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
    return NUFFT_OK;
}

static int upload_i32(nufft_plan* p, int32_t** dst, const std::vector<int32_t>& src) {
    int rc = dev_alloc(p, reinterpret_cast<void**>(dst), src.size() * sizeof(int32_t));
    if (rc) return rc;
    NUFFT_HIP(hipMemcpy(*dst, src.data(), src.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    return NUFFT_OK;
}

// ------------------------------------------------------------------------------------------
// stage timing
// ------------------------------------------------------------------------------------------
struct StageTimer {
    nufft_plan* p;
    int stage;
    hipStream_t stream;
    StageTimer(nufft_plan* plan, int st, hipStream_t s) : p(plan), stage(st), stream(s) {
        if (p->timing) (void)hipEventRecord(static_cast<hipEvent_t>(p->ev_begin[stage]), stream);
    }
    ~StageTimer() {
        if (p->timing) {
            (void)hipEventRecord(static_cast<hipEvent_t>(p->ev_end[stage]), stream);
            p->ev_valid[stage] = true;
        }
    }
};

// ------------------------------------------------------------------------------------------
// plan construction
// ------------------------------------------------------------------------------------------
// development switches: nufft_params.options of the plan at hand (options.h) — never the environment in a release build
static int env_int(const char* name, int fallback) { return option_int(name, fallback); }

// Default of the spreading ring's halo variant for this plan (0: clipped columns, 2: halo variant); see DESIGN.md section 4.9.
// Measured (256^3 -> 512^3, Np = 1e7; spread + FFT stages, ms; scripts/r4_halo_matrix.sh): Float64 m = 2 ... 7: 2.08 / 2.57 / 3.03 / 5.72 /
// 11.4 / 20.7 with clipped columns, 1.88 / 2.22 / 2.62 / 4.68 / 8.35 / 14.8 with the halo variant; Float32 alike (m = 8: 17.3 -> 13.1).
// Only where the plan's own dimension-1 pass adds the side buffer (real plans on the pruned FFT path); smarch_plan falls back to the
// clipped columns where the variant cannot run (axes the column does not divide, LDS).
// Complex data (interleaved components double the window): it pays where a 32 x 16 column or larger still fits — ComplexF64 m = 2 (3.29 -> 2.84 ms),
// ComplexF32 m = 2, 3 (2.55 -> 1.97, 4.03 -> 3.51); with 16 x 16 columns the reach is 1.2 x the grid (ComplexF64 m = 3: 4.70 -> 5.08 ms).
static int smarch_halo_default(const nufft_plan* p) {      // (build_device: ... and only on the plan's own pruned FFT path)
    if (p->D != 3) return 0;
    if (!p->is_complex || p->smarch_parts == 2) return 2;      // (complex data part by part through the real kernel: as real data)
    return p->M <= (p->dtype == NUFFT_F32 ? 3 : 2) ? 2 : 0;
}

// Largest half-support for which the automatic choice takes the marching window (measured spread + FFT stages):
// real data 6 (Float32 with the halo variant: 7); complex data part by part through the real kernel: 6 — measured round 5, 256^3 -> 512^3,
// Np = 1e7, spread + FFT stages in ms, window against patches: ComplexF64 m = 4 5.46 / 6.31 (interleaved window), m = 5 9.87 / 14.30, m = 6 14.47 / 24.88,
// m = 7 24.10 / 24.29; ComplexF32 m = 4 4.65 / 5.61, m = 5 8.78 / 9.94, m = 6 13.08 / 15.08, m = 7 20.67 / 14.94; the interleaved complex
// instantiations (NUFFT_SMARCH_SPLIT=0): ComplexF64 4, ComplexF32 3
static int ring_max_half_support(const nufft_plan* p) {
    if (!p->is_complex) return (p->dtype == NUFFT_F32 && p->smarch.halo == 2) ? 7 : 6;
    if (p->smarch.parts == 2 && p->smarch.halo == 2) return 6;
    return p->dtype == NUFFT_F64 ? 4 : 3;
}

static void predict_sort_column(nufft_plan* p);

static int build_host(nufft_plan* p, const nufft_params* in) {
    p->opts.parse(in->options);
    set_current_options(&p->opts);
    p->dtype = in->dtype;
    p->is_complex = in->is_complex != 0;
    p->D = in->ndim;
    p->M = in->half_support > 0 ? in->half_support : 4;
    p->sigma_req = in->sigma > 0 ? in->sigma : 2.0;
    p->evalmode = in->evalmode;
    p->C = in->ntransforms > 0 ? in->ntransforms : 1;
    p->fftshift = in->fftshift != 0;
    p->device = in->device;

    if (p->dtype != NUFFT_F32 && p->dtype != NUFFT_F64) return fail(NUFFT_ERR_INVALID_ARG, "dtype must be NUFFT_F32 or NUFFT_F64");
    if (p->D < 1 || p->D > 3) return fail(NUFFT_ERR_UNSUPPORTED, "ndim must be 1, 2 or 3");
    if (in->kernel < NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL || in->kernel > NUFFT_KERNEL_BSPLINE)
        return fail(NUFFT_ERR_UNSUPPORTED, "kernel must be one of NUFFT_KERNEL_* (BackwardsKaiserBessel, KaiserBessel, Gaussian, BSpline)");
    p->kernel = in->kernel;
    if (in->kernel_param < 0.0 || (in->kernel_param != 0.0 && p->kernel == NUFFT_KERNEL_BSPLINE))
        return fail(NUFFT_ERR_INVALID_ARG, "kernel_param must be positive (and BSplineKernel has no parameter)");
    if (p->evalmode != NUFFT_EVAL_DIRECT && p->evalmode != NUFFT_EVAL_FAST_APPROXIMATION)
        return fail(NUFFT_ERR_INVALID_ARG, "evalmode must be Direct (0) or FastApproximation (1)");
    // gpu_method only reschedules the same sums in the reference (src/spreading/gpu.jl:168-214): both symbols are accepted
    if (in->gpu_method != NUFFT_METHOD_SHARED_MEMORY && in->gpu_method != NUFFT_METHOD_GLOBAL_MEMORY)
        return fail(NUFFT_ERR_INVALID_ARG, "expected gpu_method ∈ (:global_memory, :shared_memory)");      // src/blocking/gpu.jl:26
    for (int d = 0; d < 3; ++d) {
        if (in->kernel_param_dim[d] < 0.0 || (in->kernel_param_dim[d] != 0.0 && p->kernel == NUFFT_KERNEL_BSPLINE))
            return fail(NUFFT_ERR_INVALID_ARG, "kernel_param_dim must be positive (and BSplineKernel has no parameter)");
        if (in->N_over[d] < 0) return fail(NUFFT_ERR_INVALID_ARG, "N_over must be >= 0");
    }
    if (in->point_transform != NUFFT_POINT_TRANSFORM_IDENTITY && in->point_transform != NUFFT_POINT_TRANSFORM_NFFT)
        return fail(NUFFT_ERR_UNSUPPORTED, "point_transform must be identity or the AbstractNFFTs convention (closures cannot cross the ABI)");
    p->point_transform = in->point_transform;
    if (p->M < kMinM || p->M > kMaxM) return fail(NUFFT_ERR_UNSUPPORTED, "half-support M must be in 2..10");
    if (!(p->sigma_req >= 1.0)) return fail(NUFFT_ERR_INVALID_ARG, "sigma must be >= 1");

    // sigma is converted to real(Z) before the size rule (src/plan.jl:573-576)
    const double sigma_wanted = p->dtype == NUFFT_F32 ? (double)(float)p->sigma_req : p->sigma_req;
    p->sigma = 0.0;
    for (int d = 0; d < p->D; ++d) {
        p->N[d] = in->N[d];
        if (p->N[d] < 1) return fail(NUFFT_ERR_INVALID_ARG, "grid dimensions must be >= 1");
        const bool r2c = !p->is_complex && d == 0;
        p->Nover[d] = oversampled_size(p->N[d], sigma_wanted, r2c);
        if (in->N_over[d] > 0) {      // the caller holds the reference's plan already: gridsize(p.kernels[d]) verbatim
            if (in->N_over[d] < p->N[d] || (r2c && (in->N_over[d] & 1)))
                return fail(NUFFT_ERR_INVALID_ARG, "N_over must be >= N (and even along dimension 1 of a real plan)");
            p->Nover[d] = in->N_over[d];
        }
        if (p->Nover[d] < 2 * p->M) {   // check_nufft_size, src/plan.jl:545-556
            return fail(NUFFT_ERR_SIZE_TOO_SMALL, "data size is too small: sigma*N = " + std::to_string(p->Nover[d]) +
                                                      " < " + std::to_string(2 * p->M) + " = 2M");
        }
        if (p->Nover[d] > (int64_t)1 << 30) return fail(NUFFT_ERR_UNSUPPORTED, "oversampled dimension exceeds 2^30");
        p->sigma = std::max(p->sigma, (double)p->Nover[d] / (double)p->N[d]);
        p->Nspec[d] = r2c ? p->Nover[d] / 2 + 1 : p->Nover[d];
    }
    p->npoly = (p->kernel == NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL || p->kernel == NUFFT_KERNEL_KAISER_BESSEL) ? p->M + 4 : 0;
    for (int d = 0; d < p->D; ++d) {
        const bool r2c = !p->is_complex && d == 0;
        double sigma_d = (double)p->Nover[d] / (double)p->N[d];               // src/plan.jl:503
        if (p->dtype == NUFFT_F32) sigma_d = (double)(float)sigma_d;
        // optimal_kernel(kernel, T, h, Ñ, σ): shape parameter in the plan's precision, or the caller's
        // (kaiser_bessel_backwards.jl:123-136, kaiser_bessel.jl:151-165, gaussian.jl:107-116, bspline.jl:86-88)
        double beta = 0.0;
        if (in->kernel_param_dim[d] > 0.0) beta = in->kernel_param_dim[d];
        else if (in->kernel_param > 0.0) beta = in->kernel_param;
        else if (p->kernel == NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL) beta = bkb_beta(p->M, sigma_d);
        else if (p->kernel == NUFFT_KERNEL_KAISER_BESSEL) beta = kb_beta(p->M, sigma_d);
        else if (p->kernel == NUFFT_KERNEL_GAUSSIAN) beta = gaussian_ell(p->M, sigma_d);
        if (p->dtype == NUFFT_F32) beta = (double)(float)beta;
        p->beta[d] = beta;
        // Power-of-two normalisation of the window: the BKB window peaks at sinh(β)/π ≈ e^β/2π (4e15 at
        // M = 8; the KB window at I0(β)), so products of D window values overflow Float32 (and 1/ϕ̂^D
        // underflows) in the reference's formulation.  Scaling window and ϕ̂ by the same 2^k is exact in
        // binary floating point, leaves every result bit-identical where the reference is finite, and
        // keeps all intermediates O(1).  The Gaussian and B-spline windows peak at <= 1: k = 0.
        p->scale_exp[d] = 0;
        p->coefs[d].clear();
        double fourier_param = beta;
        if (p->kernel == NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL) {
            p->scale_exp[d] = -(int)std::lround(std::log2(std::sinh(beta) / M_PI));
            const double bop = p->dtype == NUFFT_F32 ? (double)((float)beta / (float)M_PI) : beta / M_PI;
            p->eval_p0[d] = beta;
            p->beta_over_pi_scaled[d] = std::ldexp(bop, p->scale_exp[d]);
            bkb_poly_coefficients(p->M, beta, p->coefs[d]);
        } else if (p->kernel == NUFFT_KERNEL_KAISER_BESSEL) {
            p->scale_exp[d] = -(int)std::lround(std::log2(bessel_i0(beta)));
            p->eval_p0[d] = beta;
            p->beta_over_pi_scaled[d] = std::ldexp(1.0, p->scale_exp[d]);
            kb_poly_coefficients(p->M, beta, p->coefs[d]);
        } else if (p->kernel == NUFFT_KERNEL_GAUSSIAN) {
            const double dx = 2.0 * M_PI / (double)p->Nover[d];
            double sg = beta * dx;
            double tau = 2.0 * sg * sg;
            if (p->dtype == NUFFT_F32) {           // σ = α Δx; τ = 2σ² in T (gaussian.jl:76-80)
                const float dxf = 2.0f * (float)M_PI / (float)p->Nover[d];
                const float sgf = (float)beta * dxf;
                tau = (double)(2.0f * sgf * sgf);
                p->eval_p0[d] = (double)dxf;
            } else {
                p->eval_p0[d] = dx;
            }
            p->tau[d] = tau;
            p->beta_over_pi_scaled[d] = tau;
            fourier_param = tau;
        } else {
            p->eval_p0[d] = 0.0;
            p->beta_over_pi_scaled[d] = 1.0;
        }
        std::vector<double> ks;
        wavenumbers(p->N[d], r2c, ks);
        p->Nout[d] = (int64_t)ks.size();
        std::vector<double> kk = ks;
        if (p->fftshift) {   // AbstractFFTs.fftshift(kx), src/plan.jl:509-511
            const size_t n = ks.size(), sh = n / 2;
            for (size_t i = 0; i < n; ++i) kk[(i + sh) % n] = ks[i];
        }
        fourier_coefficients_kernel(p->kernel, kk, p->M, p->Nover[d], fourier_param, p->phihat[d]);
        non_oversampled_indices(ks, p->Nspec[d], p->fftshift, p->index_map[d]);
    }

    // bins and tile geometry
    p->spread_threads = in->spread_threads > 0 ? in->spread_threads : env_int("NUFFT_SPREAD_THREADS", 1024);
    p->interp_threads = in->interp_threads > 0 ? in->interp_threads : env_int("NUFFT_INTERP_THREADS", 1024);
    if (p->spread_threads % 64 || p->interp_threads % 64 || p->spread_threads > 1024 || p->interp_threads > 1024 ||
        p->spread_threads < 64 || p->interp_threads < 64)
        return fail(NUFFT_ERR_INVALID_ARG, "workgroup sizes must be multiples of 64 in 64..1024");
    int budget = in->lds_budget_bytes > 0 ? in->lds_budget_bytes : env_int("NUFFT_LDS_BUDGET", kLdsLimit);
    if (budget > kLdsLimit - 256) budget = kLdsLimit - 256;   // small margin for compiler-generated LDS
    const int ncomp = p->is_complex ? 2 : 1;
    const int rb = (int)real_bytes(p);
    int forced_sp[3] = {in->tile_dims[0], in->tile_dims[1], in->tile_dims[2]};
    int forced_ip[3] = {in->interp_tile_dims[0], in->interp_tile_dims[1], in->interp_tile_dims[2]};
    auto env_tile = [](const char* name, int* out) {
        const char* e = option_str(name);
        if (e && *e && out[0] <= 0) {
            int a = 0, b = 0, c = 0;
            const int n = std::sscanf(e, "%d,%d,%d", &a, &b, &c);
            if (n >= 1) { out[0] = a; out[1] = n >= 2 ? b : a; out[2] = n >= 3 ? c : (n >= 2 ? b : a); }
        }
    };
    env_tile("NUFFT_SPREAD_TILE", forced_sp);
    env_tile("NUFFT_INTERP_TILE", forced_ip);
    int bin_log2 = in->bin_log2 > 0 ? in->bin_log2 : env_int("NUFFT_BIN_LOG2", 2);
    if (bin_log2 < 1 || bin_log2 > 4) return fail(NUFFT_ERR_INVALID_ARG, "bin_log2 must be in 1..4");
    // Default configuration on a large grid: take the compile-time interpolation tile, whose kernel
    // variant has constant LDS strides (fixed_interp_tile, device_common.h).
    int fixed_ip[4] = {0, 0, 0, 0};
    if (forced_ip[0] <= 0 && bin_log2 == 2 && p->interp_threads == 1024 && budget == kLdsLimit - 256 &&
        env_int("NUFFT_INTERP_FIXED", 1)) {
        interp_fixed_dims(p->dtype, p->is_complex, p->D, p->M, fixed_ip);
        bool ok = fixed_ip[0] > 0;
        for (int d = 0; d < p->D && ok; ++d) ok = fixed_ip[d] < p->Nover[d];
        if (ok) for (int d = 0; d < 3; ++d) forced_ip[d] = fixed_ip[d];
        else fixed_ip[0] = 0;
    }
    // the same for the spreading tile (fixed_spread_tile): every edge must leave room for the clipped halo
    int fixed_sp[5] = {0, 0, 0, 0, 0};
    if (forced_sp[0] <= 0 && bin_log2 == 2 && p->spread_threads == 1024 && budget == kLdsLimit - 256 &&
        env_int("NUFFT_SPREAD_FIXED", 1)) {
        spread_fixed_dims(p->dtype, p->is_complex, p->D, p->M, fixed_sp);
        bool ok = fixed_sp[0] > 0;
        for (int d = 0; d < p->D && ok; ++d) ok = fixed_sp[d] + 2 * p->M - 1 <= p->Nover[d] && fixed_sp[d] < p->Nover[d];
        if (ok) for (int d = 0; d < 3; ++d) forced_sp[d] = fixed_sp[d];
        else fixed_sp[0] = 0;
    }
    // Tile edges are multiples of the bin edge; when even one bin plus halo overflows the LDS (large M,
    // complex Float64) retry with smaller bins down to single cells.
    bool found = false;
    for (; bin_log2 >= 0 && !found; --bin_log2) {
        found = choose_tiles(p->D, p->M, ncomp, rb, p->Nover, budget, p->spread_threads / 64, p->interp_threads / 64,
                             forced_sp, forced_ip, bin_log2, p->tile);
        if (!found && (fixed_sp[0] > 0 || fixed_ip[0] > 0)) {
            // the compile-time tiles are not the caller's choice: fall back to the run-time search
            if (fixed_sp[0] > 0) forced_sp[0] = forced_sp[1] = forced_sp[2] = 0;
            if (fixed_ip[0] > 0) forced_ip[0] = forced_ip[1] = forced_ip[2] = 0;
            fixed_sp[0] = fixed_ip[0] = 0;
            found = choose_tiles(p->D, p->M, ncomp, rb, p->Nover, budget, p->spread_threads / 64, p->interp_threads / 64,
                                 forced_sp, forced_ip, bin_log2, p->tile);
        }
    }
    if (!found) {
        return fail(NUFFT_ERR_LDS_TOO_SMALL,
                    "LDS is too small for the chosen problem (element type, half-support M, dimensions): "
                    "reduce M or the tile size");
    }
    p->interp_fixed = fixed_ip[0] > 0;
    for (int d = 0; d < p->D; ++d) p->interp_fixed = p->interp_fixed && p->tile.ip.n[d] == fixed_ip[d];
    p->interp_fixed = p->interp_fixed && p->tile.ip.row_stride == fixed_ip[3];
    p->spread_fixed = fixed_sp[0] > 0 && p->tile.sp.row_stride == fixed_sp[3] && p->tile.sp.plane_stride == fixed_sp[4];
    for (int d = 0; d < p->D; ++d) p->spread_fixed = p->spread_fixed && p->tile.sp.n[d] == fixed_sp[d] && p->tile.sp.nt[d] > 1;
    // cube accumulation: the tile's cubes must be the grid's cubes (oversampled sizes that are multiples of 4)
    // (opt-in: measured slower than the face mapping at C2, 3.94 against 3.09 ms — the LDS pipe is saturated either way,
    // DESIGN.md section 4.4)
    p->spread_cubes = p->spread_fixed && env_int("NUFFT_SPREAD_CUBES", 0) != 0 &&
                      spread_cubes_available(p->dtype, p->is_complex, p->D, p->M);
    for (int d = 0; d < p->D; ++d) p->spread_cubes = p->spread_cubes && p->Nover[d] % 4 == 0;
    p->lds_spread = lds_layout((int)p->tile.sp.elems, 8, rb, p->D, p->M, ncomp, p->spread_threads / 64, p->tile.sp.max_items,
                               spread_strip_pad(p->D, ncomp)).total;
    p->lds_interp = lds_layout((int)p->tile.ip.elems, rb, rb, p->D, p->M, ncomp, p->interp_threads / 64, p->tile.ip.max_items).total;
    if (p->lds_spread > kLdsLimit || p->lds_interp > kLdsLimit)
        return fail(NUFFT_ERR_LDS_TOO_SMALL, "LDS is too small for the chosen problem: work-item table does not fit");
    if (p->tile.nbins >= ((int64_t)1 << 31) - 2) return fail(NUFFT_ERR_UNSUPPORTED, "too many bins");
    // spreading engine: MFMA patches where they apply (3-D, 4-cell bins, default window evaluation), LDS tiles otherwise
    {
        int req = in->spread_method != NUFFT_SPREAD_AUTO ? in->spread_method : env_int("NUFFT_SPREAD_METHOD", NUFFT_SPREAD_AUTO);
        if (req < NUFFT_SPREAD_AUTO || req > NUFFT_SPREAD_MARCHING_RING) return fail(NUFFT_ERR_INVALID_ARG, "unknown spread_method");
        p->spread_method_req = req;
        // NUFFT_PATCH_F32ACC=0: ComplexF32 plans keep the Float64-accumulating patch kernel (A/B runs)
        // NUFFT_PATCH_PLANAR=0: the components of a real plan with ntransforms = 2 / 3 are spread one after the other (A/B runs)
        const int planar_nc = (!p->is_complex && (p->C == 2 || p->C == 3) && env_int("NUFFT_PATCH_PLANAR", 1) != 0) ? p->C : 0;
        const PatchPlan pp = patch_plan(p->dtype, p->is_complex, p->D, p->M, make_geom(p), needs_other_eval(p->kernel, p->evalmode),
                                        env_int("NUFFT_PATCH_F32ACC", 1) != 0, planar_nc);
        p->patch.eligible = pp.eligible;
        p->patch.npx = pp.npx; p->patch.npy = pp.npy; p->patch.nseg = pp.nseg; p->patch.segl = pp.segl;
        p->patch.ntasks = pp.ntasks; p->patch.lds_bytes = pp.lds_bytes; p->patch.pby = pp.pby; p->patch.occ = pp.occ; p->patch.f32acc = pp.f32acc; p->patch.planar = pp.planar;
        if (req == NUFFT_SPREAD_MFMA_PATCHES && !pp.eligible)
            return fail(NUFFT_ERR_UNSUPPORTED, "spread_method = MFMA patches needs a 3-D grid of 4-cell bins with every oversampled "
                                               "axis a multiple of 4 and at least 2 (patch + stencil) bins long, and the default window evaluation");
        // automatic choice, from the measurements in DESIGN.md section 4.4: the patches win where the stencil carries
        // more matrix work per point visit (complex data, M >= 5: 1.15x ... 2x), the LDS tiles for real data at M <= 4
        // — and for real plans with ntransforms = 2 / 3, whose components the patches spread together (shared windows and operands:
        // C4 9.1 -> 7.5 ms, two components 6.1 -> 5.2 ms)
        const bool prefer_patches = p->is_complex || p->M >= 5 || pp.planar != 0 || env_int("NUFFT_PREFER_PATCHES", 0) != 0;     // (the switch: test runs)
        p->spread_method = (pp.eligible && (req == NUFFT_SPREAD_MFMA_PATCHES || (req == NUFFT_SPREAD_AUTO && prefer_patches)))
                               ? NUFFT_SPREAD_MFMA_PATCHES : NUFFT_SPREAD_LDS_TILES;
        // third engine, the z-marching LDS window (smarch_kernels.h): the per-point arithmetic of the LDS tiles with 1.5 instead of
        // 2.1 visits per point and no per-plane control.  Automatic choice from the measured spread stage of the three engines
        // (256^3 -> 512^3, Np = 1e7): real data up to M = 6 (M = 4: 2.44 ms against 3.07 tiles / 4.05 patches;
        // M = 6: 10.8 / 15.5 / 13.2; M = 7: 20.1 against 13.4 patches), ntransforms components one after the other (C = 3: 7.3 ms
        // against 7.5 with the planar patches); ComplexF64 up to M = 4 (5.06 against 5.36 patches), ComplexF32 up to M = 3 (M = 4:
        // 4.71 against 4.10 patches).  256 CUs assumed here, build_device() redoes the decomposition for the device.
        // halo variant of the ring (every point spread once, the stencil reach through a side buffer that the first FFT pass adds):
        // real data on grids whose axes the column divides; NUFFT_SMARCH_HALO=0 / 2 forces the choice (A/B runs, tests)
        // complex data part by part through the REAL window kernel (smarch_kernels.h: 8 x 8 faces, 32 x 32 columns, halo variant — ComplexF64
        // m = 4: 5.06 -> 3.9 ms); NUFFT_SMARCH_SPLIT=0: the interleaved complex instantiations (A/B runs, tests)
        p->smarch_parts = (p->is_complex && env_int("NUFFT_SMARCH_SPLIT", 1) != 0) ? 2 : 1;
        const int want_halo = env_int("NUFFT_SMARCH_HALO", smarch_halo_default(p)) == 2 ? 2 : 0;
        p->smarch = smarch_plan(p->dtype, p->is_complex, p->D, p->M, make_geom(p), needs_other_eval(p->kernel, p->evalmode), 256, p->C, want_halo, p->smarch_parts);
        if (!p->smarch.eligible && want_halo)
            p->smarch = smarch_plan(p->dtype, p->is_complex, p->D, p->M, make_geom(p), needs_other_eval(p->kernel, p->evalmode), 256, p->C, 0, p->smarch_parts);
        if (!p->smarch.eligible && p->smarch_parts == 2) {      // no real-kernel decomposition: the complex instantiations as before
            p->smarch_parts = 1;
            const int wh = env_int("NUFFT_SMARCH_HALO", smarch_halo_default(p)) == 2 ? 2 : 0;
            p->smarch = smarch_plan(p->dtype, p->is_complex, p->D, p->M, make_geom(p), needs_other_eval(p->kernel, p->evalmode), 256, p->C, wh, 1);
            if (!p->smarch.eligible && wh)
                p->smarch = smarch_plan(p->dtype, p->is_complex, p->D, p->M, make_geom(p), needs_other_eval(p->kernel, p->evalmode), 256, p->C, 0, 1);
        }
        if (req == NUFFT_SPREAD_MARCHING_RING && !p->smarch.eligible)
            return fail(NUFFT_ERR_UNSUPPORTED, "spread_method = marching ring needs a 3-D grid of 4-cell bins with every oversampled axis a multiple "
                                               "of 4 and longer than a column plus a stencil, and the default window evaluation");
        // (with the halo variant the window also beats the patches for Float32 at M = 7: 10.8 against 12.8 ms spread + FFT)
        const int ring_max_m = ring_max_half_support(p);
        const bool prefer_ring = env_int("NUFFT_PREFER_RING", 1) != 0 && p->M <= ring_max_m && env_int("NUFFT_PREFER_PATCHES", 0) == 0;
        if (p->smarch.eligible && (req == NUFFT_SPREAD_MARCHING_RING || (req == NUFFT_SPREAD_AUTO && prefer_ring)))
            p->spread_method = NUFFT_SPREAD_MARCHING_RING;
    }
    if (p->device < 0) predict_sort_column(p);
    return NUFFT_OK;
}

// Column-layer sort: ONE column for both rings — the spreading window's (chosen for this grid by its launch model, smarch_plan); the
// interpolation ring takes it wherever its kernels can hold it (a column <= their compile-time one: march_setup.inc).  Round 6; before,
// only plans whose two rings happened to pick the same column qualified (Float64 / ComplexF64 m = 4).  `sm`: the window's final plan
// (halo variant), p->interp_parts decided.  Pure host arithmetic: also what the prediction for host-only plans runs (predict_sort_column).
static bool shared_ring_column(const nufft_plan* p, const SMarchPlan& sm, ColumnTasks* shared) {
    if (!sm.eligible || sm.halo != 2 || p->D != 3) return false;
    const int nkeys = sm.ct.ncolx * sm.ct.ncoly * p->tile.nb[2];
    const int mcplx = p->interp_parts == 2 ? 0 : (int)p->is_complex;
    const bool poly = p->evalmode != NUFFT_EVAL_DIRECT;
    if (p->Nover[0] % sm.n1 != 0 || p->Nover[1] % sm.n2 != 0 || nkeys > kCoarseMaxKeys) return false;
    // Measured (256^3 -> 512^3, Np = 1e7, set_points + spread / set_points + interpolation against the plan without it, ms; scripts/r6_e.sh,
    // profiles/round6_*): Float64 m = 2 / 3 / 4 -0.18 / -0.18 / -0.12 and -0.14 / -0.20 / -0.21, ComplexF64 m = 4 -0.12 / -0.29, Float32 m = 3 / 4
    // -0.06 / -0.03 and -0.07 / -0.04, ComplexF32 m = 3 / 4 -0.15 / -0.04 and -0.02 / -0.02 (polynomial window: +0.02) — and Float32 m = 2
    // -0.11 but +0.12 (Direct) / +0.22 (polynomial) on the type-2 side: its ring gathers through the LDS strips (no register window at 4-lane
    // rows) from a 64 x 64 column of its own, which the 32 x 32 window column cannot replace.  That one keeps its own column
    // (NUFFT_COARSE_SORT=2: shared all the same — the test of its staged instantiation).
    if (p->dtype == NUFFT_F32 && !p->is_complex && p->M == 2 && env_int("NUFFT_COARSE_SORT", 1) != 2) return false;
    if (!interp_march_staged_available(p->dtype, mcplx, p->M, poly, sm.n1, sm.n2)) return false;
    const ColumnTasks ct = march_column_tasks(p->dtype, mcplx, p->M, poly, make_geom(p), sm.n1, sm.n2);
    if (ct.ntasks <= 0 || ct.ncolx != sm.ct.ncolx || ct.ncoly != sm.ct.ncoly || (size_t)ct.ncolx * ct.ncoly >= 65536 || p->tile.nb[2] > 2048) return false;
    *shared = ct;
    return true;
}

// interpolation ring: which form (complex data part by part through the real kernels?) and whether it exists for this plan — host arithmetic
static bool choose_interp_ring(nufft_plan* p) {
    p->interp_march_mode = env_int("NUFFT_INTERP_MARCH", 1);
    // ComplexF64 part by part through the REAL ring kernels (march_setup.inc, MarchGeom::parts = 2): the complex instantiation reads 128-bit pairs
    // from LDS at a quarter of the rate (scripts/microbench7.hip) and owns a narrower column — two passes of the real kernel are faster from
    // m = 4 on (interpolation stage 256^3 -> 512^3, Np = 1e7: 2.76 against 2 x 1.2 ms; m = 8: 20.4 against 2 x 5.8), and the plan can then share
    // its columns with the spreading window (column-layer sort).  ComplexF32 keeps its paired-lane kernel (1.25 ms against 2 x 1.08).
    // NUFFT_INTERP_SPLIT=0: the complex instantiations (A/B runs, tests)
    p->interp_parts = (p->is_complex && p->dtype == NUFFT_F64 && env_int("NUFFT_INTERP_SPLIT", 1) != 0) ? 2 : 1;
    const bool other = needs_other_eval(p->kernel, p->evalmode), poly = p->evalmode != NUFFT_EVAL_DIRECT;
    bool ok = p->interp_march_mode != 0 && interp_march_available(p->dtype, p->interp_parts == 2 ? 0 : (int)p->is_complex, p->D, p->M, poly, make_geom(p), other);
    if (!ok && p->interp_parts == 2) {
        p->interp_parts = 1;
        ok = p->interp_march_mode != 0 && interp_march_available(p->dtype, p->is_complex, p->D, p->M, poly, make_geom(p), other);
    }
    return ok;
}

// What a device plan of these parameters would report as nufft_info.sort_column on a 256-CU device whose side buffer can be allocated:
// the decisions of build_device that need no device, in its order.  Host-only plans report it (tests compute a GPU test's eligibility here).
static void predict_sort_column(nufft_plan* p) {
    p->sort_column_pred[0] = p->sort_column_pred[1] = 0;
    if (p->D != 3 || p->spread_method != NUFFT_SPREAD_MARCHING_RING || env_int("NUFFT_COARSE_SORT", 1) == 0) return;
    bool pruned = env_int("NUFFT_PRUNED_FFT", 1) != 0;
    for (int d = p->is_complex ? 0 : 1; d < p->D && pruned; ++d) pruned = fft_lines_supported(p->dtype, p->Nover[d]);
    const bool compact = pruned && (p->is_complex || (real_lines_supported(p->dtype, p->Nover[0]) && env_int("NUFFT_COMPACT_DIM1", 1) != 0));
    int want_halo = p->smarch.halo;
    if (want_halo == 2 && !(pruned && compact) && env_int("NUFFT_SMARCH_HALO", 0) != 2) want_halo = 0;
    if (want_halo != 2) return;
    SMarchPlan sm = smarch_plan(p->dtype, p->is_complex, p->D, p->M, make_geom(p), needs_other_eval(p->kernel, p->evalmode), 256, p->C, want_halo, p->smarch_parts);
    if (!sm.eligible || sm.halo != 2) return;
    const int save_parts = p->interp_parts, save_mode = p->interp_march_mode;
    ColumnTasks shared{};
    if (choose_interp_ring(p) && shared_ring_column(p, sm, &shared)) { p->sort_column_pred[0] = shared.bxw; p->sort_column_pred[1] = shared.byw; }
    p->interp_parts = save_parts; p->interp_march_mode = save_mode;
}

static int build_device(nufft_plan* p) {
    int ndev = 0;
    NUFFT_HIP(hipGetDeviceCount(&ndev));
    if (p->device >= ndev) return fail(NUFFT_ERR_INVALID_ARG, "device ordinal out of range");
    DeviceGuard guard(p->device);
    std::call_once(g_rocfft_once, [] { (void)rocfft_setup(); });
    {
        // every engine set-up below (ring launch models, sort slices) sizes itself for this device — not for the default of 256 CUs
        hipDeviceProp_t prop;
        NUFFT_HIP(hipGetDeviceProperties(&prop, p->device));
        p->num_cus = prop.multiProcessorCount;
    }

    int rc;
    const int D = p->D, L = 2 * p->M;
    // polynomial coefficients [D][npoly][2M]
    {
        std::vector<double> all;
        for (int d = 0; d < D; ++d)
            for (double c : p->coefs[d]) all.push_back(std::ldexp(c, p->scale_exp[d]));
        if (all.empty()) all.push_back(0.0);      // kernels without a polynomial form (Gaussian, B-spline)
        (void)L;
        rc = p->dtype == NUFFT_F32 ? upload<float>(p, &p->d_coefs, all) : upload<double>(p, &p->d_coefs, all);
        if (rc) return rc;
    }
    for (int d = 0; d < D; ++d) {
        {
            // round to T first, then scale: the scaled device value is exactly 2^k times the reference's
            std::vector<double> ph(p->phihat[d].size());
            for (size_t i = 0; i < ph.size(); ++i) {
                const double v = p->dtype == NUFFT_F32 ? (double)(float)p->phihat[d][i] : p->phihat[d][i];
                ph[i] = std::ldexp(v, p->scale_exp[d]);
            }
            rc = p->dtype == NUFFT_F32 ? upload<float>(p, &p->d_phihat[d], ph) : upload<double>(p, &p->d_phihat[d], ph);
            if (rc) return rc;
        }
        std::vector<int32_t> im(p->index_map[d].size());
        std::vector<int32_t> inv((size_t)p->Nspec[d], -1);
        for (size_t i = 0; i < im.size(); ++i) {
            im[i] = (int32_t)p->index_map[d][i];
            inv[(size_t)im[i]] = (int32_t)i;
        }
        if ((rc = upload_i32(p, &p->d_index_map[d], im))) return rc;
        if ((rc = upload_i32(p, &p->d_inv_map[d], inv))) return rc;
    }

    // oversampled arrays (init_plan_data, src/plan.jl:37-60), components contiguous
    p->grid_elems = 1;
    p->spec_elems = 1;
    for (int d = 0; d < D; ++d) {
        p->grid_elems *= p->Nover[d];
        p->spec_elems *= p->Nspec[d];
    }
    if ((rc = dev_alloc(p, &p->d_us, (size_t)p->grid_elems * value_bytes(p) * p->C, "us"))) return rc;
    // pruned FFT path: D >= 2, every higher dimension (and dimension 1 of complex plans) of an instantiated length — decided before the
    // spectra and the rocFFT plans are set up: a plan on this path never runs the D-dimensional rocFFT transform, so it neither creates
    // those plans nor allocates their work buffer (round 6: 1.07 GB of C2's 4.5 GB, 8.6 GB of C3's 27.8 GB were that buffer), and a real
    // plan whose dimension-1 pass is the library's own keeps the COMPACT spectrum only (N1/2 + 1 modes per line, rows padded to 128 bytes:
    // 0.57 instead of 1.08 GB per component at C2)
    p->pruned_fft = D >= 2 && env_int("NUFFT_PRUNED_FFT", 1) != 0;
    for (int d = p->is_complex ? 0 : 1; d < D && p->pruned_fft; ++d) p->pruned_fft = fft_lines_supported(p->dtype, p->Nover[d]);
    p->compact_dim1 = p->pruned_fft && (p->is_complex || (real_lines_supported(p->dtype, p->Nover[0]) && env_int("NUFFT_COMPACT_DIM1", 1) != 0));
    // row stride of the compact dimension-1 spectrum and of tmp2: the strided passes read groups of adjacent columns,
    // 128 bytes per row — with rows of N1/2 + 1 = 129 elements every group straddled two cache lines (PMC: 1.6x the bytes
    // read); real plans pad their intermediate rows to 128 bytes
    p->spec_row = p->compact_dim1 ? p->Nout[0] : p->Nspec[0];
    if (p->compact_dim1 && !p->is_complex && env_int("NUFFT_FFT_PAD_ROWS", 1) != 0) {
        const int64_t q = 128 / (int64_t)(2 * real_bytes(p));
        p->spec_row = (p->Nout[0] + q - 1) / q * q;
        if (p->spec_row > p->Nspec[0]) p->spec_row = p->Nout[0];      // (cannot happen for sigma >= 1.25)
    }
    p->pspec_elems = p->spec_elems;
    if (p->compact_dim1) {
        p->pspec_elems = p->is_complex ? p->Nout[0] : p->spec_row;
        for (int d = 1; d < D; ++d) p->pspec_elems *= p->Nover[d];
    }
    if (!p->is_complex || p->pruned_fft) {
        if ((rc = dev_alloc(p, &p->d_uhat, (size_t)p->pspec_elems * 2 * real_bytes(p) * p->C, "uhat"))) return rc;
    }

    // bin-sort scratch that does not depend on Np
    const size_t nt1 = (size_t)p->tile.nbins + 1;
    if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->d_counts), nt1 * sizeof(uint32_t), "bin_counts"))) return rc;
    if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->d_offsets), nt1 * sizeof(uint32_t), "bin_offsets"))) return rc;
    p->scan_tmp_bytes = binsort_scan_tmp_bytes((int)p->tile.nbins);
    if ((rc = dev_alloc(p, &p->d_scan_tmp, p->scan_tmp_bytes))) return rc;
    // load balance: slot tables for tiles + a budget of extra slices (a quarter of the tiles, at least 1024)
    p->balance_enabled = env_int("NUFFT_BALANCE", 1) != 0;
    {
        nufft_plan::Balance& b = p->bal;
        const int64_t nsp = p->tile.sp.ntiles, nip = p->tile.ip.ntiles;
        // budget of extra slices per tiling (NUFFT_BALANCE_EXTRA: tests force 0 so that small grids, where the budget
        // always reaches some tile, still run the engines that serve unsliced point sets)
        const int64_t extra_env = env_int("NUFFT_BALANCE_EXTRA", -1);
        b.extra[0] = p->balance_enabled ? (uint32_t)(extra_env >= 0 ? extra_env : std::max<int64_t>(1024, nsp / 4)) : 0u;
        b.extra[1] = p->balance_enabled ? (uint32_t)(extra_env >= 0 ? extra_env : std::max<int64_t>(1024, nip / 4)) : 0u;
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&b.d_work), balance_work_words((int)(nsp + nip)) * sizeof(uint32_t)))) return rc;
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&b.d_nslices), (size_t)(nsp + nip + 1) * sizeof(uint32_t)))) return rc;
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&b.d_desc_off), (size_t)(nsp + nip + 1) * sizeof(uint32_t)))) return rc;
        if ((rc = dev_alloc(p, &b.d_desc, (size_t)(nsp + nip + b.extra[0] + b.extra[1]) * 8))) return rc;
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&b.d_slots), 2 * sizeof(uint32_t)))) return rc;
        b.tmp_bytes = balance_scan_tmp_bytes((int)(nsp + nip));
        if ((rc = dev_alloc(p, &b.d_tmp, b.tmp_bytes))) return rc;
    }

    // rocFFT plans (plan_rfft / plan_brfft / plan_fft! / plan_bfft!, src/plan.jl:45-46,57-58): the general path only
    size_t lengths[3] = {1, 1, 1};
    for (int d = 0; d < D; ++d) lengths[d] = (size_t)p->Nover[d];
    const rocfft_precision prec = p->dtype == NUFFT_F32 ? rocfft_precision_single : rocfft_precision_double;
    NUFFT_ROCFFT(rocfft_execution_info_create(&p->fft_info));
    if (!p->pruned_fft) {
        if (p->is_complex) {
            NUFFT_ROCFFT(rocfft_plan_create(&p->fft_fw, rocfft_placement_inplace, rocfft_transform_type_complex_forward, prec,
                                            (size_t)D, lengths, (size_t)p->C, nullptr));
            NUFFT_ROCFFT(rocfft_plan_create(&p->fft_bw, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, prec,
                                            (size_t)D, lengths, (size_t)p->C, nullptr));
        } else {
            NUFFT_ROCFFT(rocfft_plan_create(&p->fft_fw, rocfft_placement_notinplace, rocfft_transform_type_real_forward, prec,
                                            (size_t)D, lengths, (size_t)p->C, nullptr));
            NUFFT_ROCFFT(rocfft_plan_create(&p->fft_bw, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, prec,
                                            (size_t)D, lengths, (size_t)p->C, nullptr));
        }
        size_t wf = 0, wb = 0;
        NUFFT_ROCFFT(rocfft_plan_get_work_buffer_size(p->fft_fw, &wf));
        NUFFT_ROCFFT(rocfft_plan_get_work_buffer_size(p->fft_bw, &wb));
        p->fft_work_bytes = std::max(wf, wb);
        if (p->fft_work_bytes > 0) {
            if ((rc = dev_alloc(p, &p->d_fft_work, p->fft_work_bytes, "rocfft_work"))) return rc;
            NUFFT_ROCFFT(rocfft_execution_info_set_work_buffer(p->fft_info, p->d_fft_work, p->fft_work_bytes));
        }
    }

    if (p->pruned_fft && !p->is_complex && !p->compact_dim1) {
        size_t len1[1] = {(size_t)p->Nover[0]};
        size_t batch = (size_t)p->C;
        for (int d = 1; d < D; ++d) batch *= (size_t)p->Nover[d];
        NUFFT_ROCFFT(rocfft_plan_create(&p->fft1_fw, rocfft_placement_notinplace, rocfft_transform_type_real_forward, prec, 1, len1, batch, nullptr));
        NUFFT_ROCFFT(rocfft_plan_create(&p->fft1_bw, rocfft_placement_notinplace, rocfft_transform_type_real_inverse, prec, 1, len1, batch, nullptr));
        size_t w1 = 0, w2 = 0;
        NUFFT_ROCFFT(rocfft_plan_get_work_buffer_size(p->fft1_fw, &w1));
        NUFFT_ROCFFT(rocfft_plan_get_work_buffer_size(p->fft1_bw, &w2));
        if (std::max(w1, w2) > p->fft_work_bytes) {
            dev_free(p, p->d_fft_work);
            p->fft_work_bytes = std::max(w1, w2);
            if ((rc = dev_alloc(p, &p->d_fft_work, p->fft_work_bytes, "rocfft_work"))) return rc;
            NUFFT_ROCFFT(rocfft_execution_info_set_work_buffer(p->fft_info, p->d_fft_work, p->fft_work_bytes));
        }
    }
    if (p->pruned_fft) {
        for (int d = 0; d < D; ++d) {
            std::vector<double> inv(p->phihat[d].size());
            for (size_t i = 0; i < inv.size(); ++i) {
                const double v = p->dtype == NUFFT_F32 ? (double)(float)p->phihat[d][i] : p->phihat[d][i];
                inv[i] = 1.0 / std::ldexp(v, p->scale_exp[d]);
            }
            rc = p->dtype == NUFFT_F32 ? upload<float>(p, &p->d_invphi[d], inv) : upload<double>(p, &p->d_invphi[d], inv);
            if (rc) return rc;
        }
        {
            std::vector<double> one(1, 1.0);
            rc = p->dtype == NUFFT_F32 ? upload<float>(p, &p->d_one, one) : upload<double>(p, &p->d_one, one);
            if (rc) return rc;
        }
        for (int d = p->compact_dim1 ? 0 : 1; d < D; ++d) {
            const int64_t n = p->Nover[d];
            std::vector<double> twf(2 * (size_t)n), twb(2 * (size_t)n);
            for (int64_t m = 0; m < n; ++m) {
                const double ang = 2.0 * M_PI * (double)m / (double)n;
                twf[2 * m] = std::cos(ang); twf[2 * m + 1] = -std::sin(ang);
                twb[2 * m] = std::cos(ang); twb[2 * m + 1] = std::sin(ang);
            }
            rc = p->dtype == NUFFT_F32 ? upload<float>(p, &p->d_tw_fw[d], twf) : upload<double>(p, &p->d_tw_fw[d], twf);
            if (rc) return rc;
            rc = p->dtype == NUFFT_F32 ? upload<float>(p, &p->d_tw_bw[d], twb) : upload<double>(p, &p->d_tw_bw[d], twb);
            if (rc) return rc;
        }
        if (D == 3) {
            const size_t elems = (size_t)(p->compact_dim1 ? p->spec_row : p->Nout[0]) * p->Nout[1] * p->Nover[2];
            if ((rc = dev_alloc(p, &p->d_tmp2, elems * 2 * real_bytes(p), "tmp2"))) return rc;
        }
    }

    // kernels: allow the large dynamic LDS allocations
    // both variants: the general one also serves per-point weights on the default kernels
    for (int other = 0; other < 2; ++other) {
        if (!other && needs_other_eval(p->kernel, p->evalmode)) continue;
        NUFFT_HIP(prepare_spread(p->dtype, p->is_complex, D, p->M, (int)p->lds_spread, other != 0));
        NUFFT_HIP(prepare_interp(p->dtype, p->is_complex, D, p->M, (int)p->lds_interp, other != 0));
    }

    // second interpolation engine (z-marching ring, march_kernels.h): 3-D plans with the default window evaluation.  Which of
    // the two kernels gathers a point set is decided on the device at set_points (heaviest ring task and total work against the
    // ring's fitted advantage over the tile kernel: balance.hip, d_march_choice[2]); the mode is latched here —
    // NUFFT_INTERP_MARCH = 0: never the ring (A/B runs), 2: always (tests of its instantiations on small grids)
    p->debug_tasks = env_int("NUFFT_DEBUG_TASKS", 0) != 0;
    p->halo_fuse = env_int("NUFFT_SMARCH_HALO_FUSE", 1) != 0;
    p->interp_march = choose_interp_ring(p);
    auto march_cplx = [&]() { return p->interp_parts == 2 ? 0 : (int)p->is_complex; };
    if (p->interp_march) {
        NUFFT_HIP(prepare_interp_march(p->dtype, march_cplx(), p->M, p->evalmode != NUFFT_EVAL_DIRECT));
        p->march_ct = march_column_tasks(p->dtype, march_cplx(), p->M, p->evalmode != NUFFT_EVAL_DIRECT, make_geom(p));
        const size_t ncols = (size_t)p->march_ct.ncolx * p->march_ct.ncoly;
        if (ncols >= 65536 || p->tile.nb[2] > 2048) p->interp_march = false;      // (beyond the task kernels' table formats)
        else {
            if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->d_march_choice), 16 * sizeof(uint32_t)))) return rc;
            NUFFT_HIP(hipMemset(p->d_march_choice, 0, 16 * sizeof(uint32_t)));
            if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->d_march_cols), (2 * ncols + 2) * sizeof(uint32_t)))) return rc;
            if ((rc = dev_alloc(p, &p->d_march_tasks, (size_t)column_task_table_entries(p->march_ct, p->tile.nb[2]) * 8))) return rc;
        }
    }

    if (p->spread_method == NUFFT_SPREAD_MARCHING_RING) {
        const bool other = needs_other_eval(p->kernel, p->evalmode);
        int want_halo = p->smarch.halo;
        if (want_halo == 2 && !(p->pruned_fft && p->compact_dim1) && env_int("NUFFT_SMARCH_HALO", 0) != 2) want_halo = 0;   // no fused consumer: not worth it
        // the side buffer (0.52 x the grid per component at 32 x 32 columns, m = 4) is workspace the clipped columns do not need: when it
        // does not fit, the plan keeps the ring without it instead of failing
        for (;;) {
            p->smarch = smarch_plan(p->dtype, p->is_complex, D, p->M, make_geom(p), other, p->num_cus, p->C, want_halo, p->smarch_parts);
            if (!p->smarch.eligible && want_halo)
                p->smarch = smarch_plan(p->dtype, p->is_complex, D, p->M, make_geom(p), other, p->num_cus, p->C, 0, p->smarch_parts);
            if (!p->smarch.eligible || p->smarch.halo != 2) break;
            const size_t bytes = (size_t)p->smarch.halo_reals * real_bytes(p) * p->C * p->smarch.parts;
            // (NUFFT_TEST_HALO_ALLOC_FAIL=1: the test of this fallback)
            if (!env_int("NUFFT_TEST_HALO_ALLOC_FAIL", 0) && dev_alloc(p, &p->d_smarch_halo, bytes, "ring_side_buffer") == NUFFT_OK) break;
            p->d_smarch_halo = nullptr;
            want_halo = 0;
        }
        // build_host chose the engine for 256 CUs and the halo variant it hoped for: redo the automatic choice for what this device got
        // (Float32 m = 7 belongs to the ring only with the halo variant: 20.1 ms clipped against 13.4 ms with the patches)
        const int ring_max_m = ring_max_half_support(p);
        const bool keep = p->smarch.eligible && (p->spread_method_req == NUFFT_SPREAD_MARCHING_RING || p->M <= ring_max_m);
        if (!keep) {
            if (p->spread_method_req == NUFFT_SPREAD_MARCHING_RING)
                return fail(NUFFT_ERR_UNSUPPORTED, "marching-ring spreading: no decomposition for this device");
            dev_free(p, p->d_smarch_halo);
            p->smarch.eligible = false;
            const bool prefer_patches = p->is_complex || p->M >= 5 || p->patch.planar != 0;
            p->spread_method = (p->patch.eligible && prefer_patches) ? NUFFT_SPREAD_MFMA_PATCHES : NUFFT_SPREAD_LDS_TILES;
        }
    }
    if (p->spread_method == NUFFT_SPREAD_MARCHING_RING) {
        NUFFT_HIP(prepare_spread_march(p->dtype, p->smarch.parts == 2 ? 0 : p->is_complex, p->M, p->smarch.halo));
        const size_t ncols = (size_t)p->smarch.ct.ncolx * p->smarch.ct.ncoly;
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->d_smarch_choice), 16 * sizeof(uint32_t)))) return rc;
        NUFFT_HIP(hipMemset(p->d_smarch_choice, 0, 16 * sizeof(uint32_t)));
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->d_smarch_cols), (2 * ncols + 2) * sizeof(uint32_t)))) return rc;
        if ((rc = dev_alloc(p, &p->d_smarch_tasks, (size_t)column_task_table_entries(p->smarch.ct, p->tile.nb[2]) * 8))) return rc;
    }

    if (p->spread_method == NUFFT_SPREAD_MFMA_PATCHES) {
        NUFFT_HIP(prepare_spread_patch(p->dtype, p->is_complex, p->M, false, p->patch.planar));
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->d_patch_choice), 16 * sizeof(uint32_t)))) return rc;
        NUFFT_HIP(hipMemset(p->d_patch_choice, 0, 16 * sizeof(uint32_t)));
        // task table of the patch engine, rebuilt by every set_points (balance.hip): [columns] points, [columns + 1] first task,
        // [ntasks] {column, layers}
        const size_t ncols = (size_t)p->patch.npx * p->patch.npy;
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->d_patch_cols), (2 * ncols + 2) * sizeof(uint32_t)))) return rc;
        if ((rc = dev_alloc(p, &p->d_patch_tasks, ((size_t)p->patch.ntasks + 2 * ncols) * 8))) return rc;      // patch_task_table_entries
        p->wave_slots = p->num_cus * 4 * p->patch.occ;      // 4 SIMDs per CU
    }

    // Column-layer sort (binsort.hip, CoarseSort): where the spreading window (halo variant: a column visits its own points only) and
    // the interpolation ring own the same columns, set_points only groups the points by (column, layer of bins) — few enough keys for
    // LDS histograms, no global atomics — and the interpolation ring's staged variant orders a layer's points by bin on their way
    // into LDS.  Per point set: only while both rings serve it (device flags); NUFFT_COARSE_SORT=0 keeps the fine sort (A/B runs).
    p->coarse = CoarseSort{};
    if (p->spread_method == NUFFT_SPREAD_MARCHING_RING && p->smarch.halo == 2 && p->interp_march && D == 3 && env_int("NUFFT_COARSE_SORT", 1) != 0) {
        ColumnTasks shared{};
        const bool same = shared_ring_column(p, p->smarch, &shared);
        const int nkeys = p->smarch.ct.ncolx * p->smarch.ct.ncoly * p->tile.nb[2];
        const int mcplx = p->interp_parts == 2 ? 0 : (int)p->is_complex;
        if (same) {
            // the ring's task tables for the shared column (they were sized for the ring's own column above)
            p->march_ct = shared;
            const ColumnTasks& mc = p->march_ct;
            const size_t ncols = (size_t)mc.ncolx * mc.ncoly;
            dev_free(p, p->d_march_cols);
            dev_free(p, p->d_march_tasks);
            if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->d_march_cols), (2 * ncols + 2) * sizeof(uint32_t)))) return rc;
            if ((rc = dev_alloc(p, &p->d_march_tasks, (size_t)column_task_table_entries(mc, p->tile.nb[2]) * 8))) return rc;
            p->coarse.enabled = 1;
            p->coarse.cbx = mc.bxw; p->coarse.cby = mc.byw; p->coarse.ncx = mc.ncolx; p->coarse.ncy = mc.ncoly;
            p->coarse.nkeys = nkeys;
            p->coarse.groups = p->num_cus;
            p->coarse.flag_a = p->d_smarch_choice + 2;
            p->coarse.flag_b = p->d_march_choice + 2;
            if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->coarse.table), (size_t)p->coarse.groups * nkeys * sizeof(uint32_t), "sort_slice_table"))) return rc;
            NUFFT_HIP(prepare_binsort_coarse(p->dtype, nkeys));
            NUFFT_HIP(prepare_interp_march_staged(p->dtype, mcplx, p->M, p->evalmode != NUFFT_EVAL_DIRECT));
        }
    }

    // Dense-set engine of the spreading window (dmarch_kernels.h): the points of a bin accumulated in registers by the FP64 matrix pipe, one flush
    // per bin.  Taken per point set from the mean bin load (set_points: dense_now); it needs the fine-bin order, so on plans of the
    // column-layer sort a dense point set takes the slab sort below.  NUFFT_DENSE=0: never; NUFFT_DENSE_MIN: the threshold (points per bin).
    p->dense_available = p->spread_method == NUFFT_SPREAD_MARCHING_RING && env_int("NUFFT_DENSE", 1) != 0 && !needs_other_eval(p->kernel, p->evalmode) &&
                         spread_dense_available(p->dtype, p->is_complex, p->M, p->evalmode != NUFFT_EVAL_DIRECT, p->smarch);
    if (p->dense_available) NUFFT_HIP(prepare_spread_dense(p->dtype, p->M, p->evalmode != NUFFT_EVAL_DIRECT));
    // Break-even against the stream of atomics, mean points per bin — measured (profiles/round6_dense_engine.md; 256^3 Float64, spread stage, ms, dense /
    // atomic): m = 4 Direct() 3.69 / 2.97 at 19 per bin, 9.38 / 9.21 at 60; polynomial window 2.76 / 2.72 at 19, 6.66 / 8.40 at 60, and on the
    // reference's folded N(0, 1) sets 4.52 / 5.57 at a mean of 19; m = 5 Direct() 6.38 / 4.13 at 4.8, 12.99 / 14.18 at 19, polynomial 5.05 / 3.91 and
    // 8.79 / 13.27; m = 6 Direct() 7.70 / 6.48 and 15.80 / 22.61, polynomial 7.95 / 7.49 and 15.14 / 23.63; m = 2, 3 lose at 19 per bin (1.15 - 1.4 x).
    // The FP64 matrix instructions run on the SIMD's FP64 vector ALUs (vector 37 % + matrix 25 % + LDS 32 % of the kernel's cycles ADD UP,
    // counters in the same file): the engine only trades LDS-atomic time for vector time, which pays where a point costs 20 - 36 atomics
    // (m = 5, 6) or the window evaluation is cheap (polynomial) and the bins are full.
    static const int dense_min_direct[7] = {0, 0, 1 << 30, 1 << 30, 96, 16, 12};
    static const int dense_min_poly[7] = {0, 0, 1 << 30, 1 << 30, 24, 10, 6};
    p->dense_min = env_int("NUFFT_DENSE_MIN", p->M <= 6 ? (p->evalmode == NUFFT_EVAL_DIRECT ? dense_min_direct : dense_min_poly)[p->M] : 1 << 30);

    // every other 3-D plan: two-level slab sort (column_tasks.h) — the sorted array and offsets of the fine sort, without global atomics
    p->slab = CoarseSort{};
    if (D == 3 && env_int("NUFFT_SLAB_SORT", 1) != 0 && p->tile.nb[0] <= kSlabMaxBins && p->tile.nb[2] <= kCoarseMaxKeys) {
        p->slab.enabled = 1;
        p->slab.mode = 2;
        p->slab_min_points = env_int("NUFFT_SLAB_MIN_POINTS", 16384);
        p->slab_fill = std::min(95, std::max(5, env_int("NUFFT_SLAB_FILL", 85)));
        // [slices][keys] table of level 1: at most one slice per CU; the keys are (rows of bins / slab height) x layers — never more than
        // nb[1] x nb[2], whatever height set_points picks (a 16^3 plan used to pay the 37.7 MB of the largest grid: ADVICE round 5)
        p->slab_max_keys = (int)std::min<int64_t>(kCoarseMaxKeys, (int64_t)p->tile.nb[1] * p->tile.nb[2]);
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->slab.table), (size_t)p->num_cus * p->slab_max_keys * sizeof(uint32_t), "sort_slice_table"))) return rc;
        if ((rc = dev_alloc(p, reinterpret_cast<void**>(&p->slab.flagmem), 32))) return rc;      // [0] fullest slab (running), [4] flag
        NUFFT_HIP(hipMemset(p->slab.flagmem, 0, 32));
        p->slab.flag_a = p->slab.flag_b = p->slab.flagmem + 4;
        NUFFT_HIP(prepare_binsort_slab(p->dtype, kLdsLimit - 256));
    }

    if (p->coarse.enabled && p->slab.enabled && env_int("NUFFT_SORT_ADAPTIVE", 1) != 0) {
        // (host-mapped: the scatter pass writes the rings' decisions here, set_points reads them without synchronising — nufft_internal.h)
        NUFFT_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->sort_feedback), 16, hipHostMallocMapped));
        p->sort_feedback[0] = p->sort_feedback[1] = 1u;
        p->sort_feedback[2] = 0u;
    }
    if (p->spread_method == NUFFT_SPREAD_MFMA_PATCHES || p->spread_method == NUFFT_SPREAD_MARCHING_RING || p->interp_march) NUFFT_HIP(prepare_column_tasks());

    for (int s = 0; s < NUFFT_NUM_STAGES; ++s) {
        hipEvent_t a, b;
        NUFFT_HIP(hipEventCreate(&a));
        NUFFT_HIP(hipEventCreate(&b));
        p->ev_begin[s] = a;
        p->ev_end[s] = b;
    }
    return NUFFT_OK;
}

static void release(nufft_plan* p) {
    if (!p) return;
    if (current_options() == &p->opts) set_current_options(nullptr);
    if (p->device >= 0) {
        DeviceGuard guard(p->device);
        (void)hipDeviceSynchronize();
        if (p->sort_feedback) (void)hipHostFree(p->sort_feedback);
        for (auto& al : p->allocs) (void)hipFree(al.first);      // every device buffer of the plan (dev_alloc)
        p->allocs.clear();
        if (p->fft1_fw) (void)rocfft_plan_destroy(p->fft1_fw);
        if (p->fft1_bw) (void)rocfft_plan_destroy(p->fft1_bw);
        if (p->fft_fw) (void)rocfft_plan_destroy(p->fft_fw);
        if (p->fft_bw) (void)rocfft_plan_destroy(p->fft_bw);
        if (p->fft_info) (void)rocfft_execution_info_destroy(p->fft_info);
        for (int s = 0; s < NUFFT_NUM_STAGES; ++s) {
            if (p->ev_begin[s]) (void)hipEventDestroy(static_cast<hipEvent_t>(p->ev_begin[s]));
            if (p->ev_end[s]) (void)hipEventDestroy(static_cast<hipEvent_t>(p->ev_end[s]));
        }
    }
    delete p;
}

static int require_device(const nufft_plan* p) {
    if (!p) return fail(NUFFT_ERR_INVALID_ARG, "null plan");
    set_current_options(&p->opts);      // the development switches any code below may consult are this plan's (options.h)
    if (p->device < 0) return fail(NUFFT_ERR_NO_DEVICE, "host-only plan (device = -1) has no device path");
    return NUFFT_OK;
}

static int require_points(const nufft_plan* p) {
    int rc = require_device(p);
    if (rc) return rc;
    if (p->Np < 0) return fail(NUFFT_ERR_NO_POINTS, "set_points must be called before exec");
    return NUFFT_OK;
}

static TileKernelArgs tile_args(const nufft_plan* p, bool interp) {
    TileKernelArgs a{};
    a.dtype = p->dtype;
    a.is_complex = p->is_complex;
    a.D = p->D;
    a.M = p->M;
    a.evalmode = p->evalmode;
    a.kernel = p->kernel;
    a.C = p->C;
    a.g = make_geom(p);
    a.sorted = p->d_sorted;
    a.offsets = p->d_offsets;
    a.coefs = p->d_coefs;
    for (int d = 0; d < 3; ++d) {
        a.beta[d] = p->eval_p0[d];
        a.beta_over_pi[d] = p->beta_over_pi_scaled[d];
    }
    a.grid = p->d_us;
    a.grid_stride = p->grid_elems;
    a.prefactor = 1.0;
    if (interp) {
        // prefactor = prod(Δx_d), src/interpolation/gpu.jl:55-56
        for (int d = 0; d < p->D; ++d) a.prefactor *= 2.0 * M_PI / (double)p->Nover[d];
    }
    a.weights = p->cb_point_weights;
    a.threads = interp ? p->interp_threads : p->spread_threads;
    a.fixed_tile = interp ? p->interp_fixed : p->spread_fixed;
    a.march = interp && p->interp_march;      // (the ring applies per-point weights itself)
    a.interp_parts = p->interp_parts;
    a.coarse = p->coarse_now;      // (a dense point set of a column-layer plan, or one behind two clustered ones, was sorted by fine bins: the plain ring gathers it)
    a.coarse_a = p->coarse.flag_a;
    a.coarse_b = p->coarse.flag_b;
    a.march_ct = p->march_ct;
    a.march_flag = p->d_march_choice ? p->d_march_choice + 2 : nullptr;
    a.march_tasks = static_cast<const uint2*>(p->d_march_tasks);
    a.cubes = !interp && p->spread_cubes && !needs_other_eval(p->kernel, p->evalmode) && !p->cb_point_weights;
    a.halo = p->d_smarch_halo;
    a.lds_bytes = (int)(interp ? p->lds_interp : p->lds_spread);
    const nufft_plan::Balance& b = p->bal;
    const int nt = (int)(interp ? p->tile.ip.ntiles : p->tile.sp.ntiles);
    a.ntiles = nt + (int)b.extra[interp ? 1 : 0];     // launch grid: tiles + budget of extra slices
    a.desc = static_cast<const char*>(b.d_desc) + (interp ? ((size_t)p->tile.sp.ntiles + b.extra[0]) * 8 : 0);
    a.desc_total = b.d_slots + (interp ? 1 : 0);
    a.xcd_chunk = env_int("NUFFT_XCD_CHUNK", 8);
    return a;
}

// ---- halo variant of the spreading ring: the state "side buffer written, reach not yet added to us" (nufft_internal.h: halo_hint) ----
static bool halo_plan(const nufft_plan* p) { return p->spread_method == NUFFT_SPREAD_MARCHING_RING && p->smarch.halo == 2 && p->d_smarch_halo && p->d_smarch_choice; }
// any call on the plan that is being captured into a hipGraph makes the host hint unreliable from then on (a replay changes the device
// state behind the host's back): every consumer / voiding launch is enqueued unconditionally afterwards, gated on the device word
static void note_capture(nufft_plan* p, hipStream_t stream) {
    if (p->halo_sticky || !halo_plan(p)) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return; }
    if (st != hipStreamCaptureStatusNone) p->halo_sticky = true;
}
static bool halo_maybe_pending(const nufft_plan* p) { return halo_plan(p) && (p->halo_hint || p->halo_sticky); }
// the grids are about to be overwritten: an unconsumed side buffer is void
static int void_halo(nufft_plan* p, hipStream_t stream) {
    if (halo_maybe_pending(p)) NUFFT_HIP(launch_zero_fill(p->d_smarch_choice + kHaloStateWord, 4, stream));
    p->halo_hint = false;
    return NUFFT_OK;
}

// The halo variant's side buffer has not been added to the grid yet (nufft_spread_deferred): do it now, unless the caller is
// the FFT pass that adds it on the fly.
static int complete_halo(nufft_plan* p, hipStream_t stream) {
    if (!halo_maybe_pending(p)) return NUFFT_OK;
    TileKernelArgs a = tile_args(p, false);
    NUFFT_HIP(launch_smarch_halo_add(a, p->smarch, p->d_smarch_choice + kHaloStateWord, stream));      // (returns at once where the word is 0)
    NUFFT_HIP(launch_zero_fill(p->d_smarch_choice + kHaloStateWord, 4, stream));
    p->halo_hint = false;
    return NUFFT_OK;
}


static DeconvArgs deconv_args(const nufft_plan* p) {
    DeconvArgs a{};
    a.dtype = p->dtype;
    a.D = p->D;
    a.C = p->C;
    for (int d = 0; d < 3; ++d) {
        a.nout[d] = (int)p->Nout[d];
        a.nspec[d] = (int)p->Nspec[d];
        a.phihat[d] = p->d_phihat[d];
        a.index_map[d] = p->d_index_map[d];
        a.inv_map[d] = p->d_inv_map[d];
    }
    a.spec = p->is_complex ? p->d_us : p->d_uhat;
    a.spec_stride = p->spec_elems;
    a.normfactor = 1.0;
    a.mode_factors = p->cb_mode_factors;
    return a;
}

static int fft_exec(nufft_plan* p, bool forward, hipStream_t stream) {
    NUFFT_ROCFFT(rocfft_execution_info_set_stream(p->fft_info, stream));
    void* in[1];
    void* out[1];
    if (p->is_complex) {
        in[0] = p->d_us;
        NUFFT_ROCFFT(rocfft_execute(forward ? p->fft_fw : p->fft_bw, in, nullptr, p->fft_info));
    } else if (forward) {
        in[0] = p->d_us;
        out[0] = p->d_uhat;
        NUFFT_ROCFFT(rocfft_execute(p->fft_fw, in, out, p->fft_info));
    } else {
        in[0] = p->d_uhat;
        out[0] = p->d_us;
        NUFFT_ROCFFT(rocfft_execute(p->fft_bw, in, out, p->fft_info));
    }
    return NUFFT_OK;
}

static int ilog2(int64_t n) {
    int l = 0;
    while (((int64_t)1 << l) < n) ++l;
    return l;
}

// ---- pruned FFT path (see fft_lines.hip) ---------------------------------------------------------
// type 1, stage "FFT": rocFFT r2c along dim 1 and, for D = 3, the pruned pass along dim 2 into tmp2.
static int pruned_forward_fft(nufft_plan* p, hipStream_t stream, bool fuse) {
    // behind the halo variant of the spreading ring (nufft_spread_deferred): the pass adds the side buffer while it loads its lines
    RealLineHalo hh{};
    if (fuse) {
        hh.flag = p->d_smarch_choice + kHaloStateWord;      // (1 implies that the ring served the point set)
        hh.ny = (int)p->Nover[1];
        hh.layout = make_halo_layout(p->smarch.n1, p->smarch.n2, p->M, (p->is_complex && p->smarch.parts != 2) ? 2 : 1, p->smarch.ct.ncolx, p->smarch.ct.ncoly);
    }
    if (p->is_complex) {
        int64_t per = 1;
        for (int d = 1; d < p->D; ++d) per *= p->Nover[d];
        const size_t cb = 2 * real_bytes(p);
        for (int c = 0; c < p->C; ++c) {
            const void* in = static_cast<char*>(p->d_us) + (size_t)c * p->grid_elems * cb;
            void* out = static_cast<char*>(p->d_uhat) + (size_t)c * p->pspec_elems * cb;
            hh.buffer = static_cast<char*>(p->d_smarch_halo) + (size_t)c * p->smarch.parts * p->smarch.halo_reals * real_bytes(p);
            hh.buffer2 = p->smarch.parts == 2 ? static_cast<const char*>(hh.buffer) + (size_t)p->smarch.halo_reals * real_bytes(p) : nullptr;      // (planar: real parts, then imaginary parts)
            NUFFT_HIP(launch_cplx_lines(p->dtype, p->Nover[0], true, in, out, per, (int)p->Nout[0], p->d_index_map[0], p->d_tw_fw[0], stream,
                                        fuse ? &hh : nullptr));
        }
        return NUFFT_OK;
    }
    if (p->compact_dim1) {
        int64_t nlines = p->C;
        for (int d = 1; d < p->D; ++d) nlines *= p->Nover[d];
        // components are contiguous both in us (Ñ1 reals per line) and in the compact spectrum (N_out1 per line);
        // the per-component offset of the compact spectrum is nlines_per_component * N_out1 <= spec_elems
        const size_t rb = real_bytes(p);
        if (p->C == 1) {
            hh.buffer = p->d_smarch_halo;
            NUFFT_HIP(launch_real_lines(p->dtype, p->Nover[0], true, p->d_us, p->d_uhat, nlines, (int)p->Nout[0], (int)p->spec_row, p->d_tw_fw[0], stream,
                                        fuse ? &hh : nullptr));
        } else {
            const int64_t per = nlines / p->C;
            for (int c = 0; c < p->C; ++c) {
                const void* in = static_cast<char*>(p->d_us) + (size_t)c * p->grid_elems * rb;
                void* out = static_cast<char*>(p->d_uhat) + (size_t)c * p->pspec_elems * 2 * rb;
                hh.buffer = static_cast<char*>(p->d_smarch_halo) + (size_t)c * p->smarch.halo_reals * rb;
                NUFFT_HIP(launch_real_lines(p->dtype, p->Nover[0], true, in, out, per, (int)p->Nout[0], (int)p->spec_row, p->d_tw_fw[0], stream,
                                            fuse ? &hh : nullptr));
            }
        }
        return NUFFT_OK;
    }
    NUFFT_ROCFFT(rocfft_execution_info_set_stream(p->fft_info, stream));
    void* in[1] = {p->d_us};
    void* out[1] = {p->d_uhat};
    NUFFT_ROCFFT(rocfft_execute(p->fft1_fw, in, out, p->fft_info));
    return NUFFT_OK;
}

static int pruned_forward_pass(nufft_plan* p, int c, int dim, void* user_out, hipStream_t stream) {
    const size_t cb = 2 * real_bytes(p);
    const int64_t K1 = p->Nout[0];
    const int64_t S1 = p->spec_row;                              // row stride of the dimension-1 spectrum
    // ... and of tmp2: unpadded on the way forward — the last pass would transform the pad columns too (5 % more lines at
    // 129 -> 136) and is not bound by its reads: measured 0.172 against 0.146 ms at C2; the padded rows of the dimension-1
    // spectrum do pay (pass 2 reads them: -0.015 ms), and so does a padded tmp2 on the way back (type 2: -0.06 ms)
    const int64_t T1 = K1;
    FftLinePass q{};
    q.map = p->d_index_map[dim];
    q.nk = (int)p->Nout[dim];
    q.twiddle = p->d_tw_fw[dim];
    const bool last = dim == p->D - 1;
    if (dim == 1) {
        q.in = static_cast<char*>(p->d_uhat) + (size_t)c * p->pspec_elems * cb;
        q.a_total = q.a_out = K1;
        q.in_stride_j = S1;
        q.in_stride_c = S1 * p->Nover[1];
        q.nc = p->D == 3 ? (int)p->Nover[2] : 1;
        q.out_stride_j = last ? K1 : T1;
        q.out_stride_c = q.out_stride_j * p->Nout[1];
        q.out = last ? user_out : p->d_tmp2;
    } else {
        q.in = p->d_tmp2;
        q.a_total = q.a_out = T1 * p->Nout[1];
        q.in_stride_j = T1 * p->Nout[1];
        q.in_stride_c = 0;
        q.nc = 1;
        q.out_stride_j = K1 * p->Nout[1];
        q.out_stride_c = 0;
        q.out = user_out;
        if (T1 != K1) { q.row_a = (int)T1; q.row_valid = (int)K1; q.row_in = (int)T1; q.row_out = (int)K1; }
    }
    q.mult = last ? p->cb_mode_factors : nullptr;      // uniform callback menu: same layout as the caller's array
    if (last) {
        // deconvolution + normalisation fused into the last pass: normfactor / (ϕ̂1 ϕ̂_last) (ϕ̂2 was applied by pass 2)
        q.fa = p->d_invphi[0]; q.ka = (int)K1;
        q.fk = p->d_invphi[dim];
        q.scale = 1.0;
        for (int d = 0; d < p->D; ++d) q.scale *= 2.0 * M_PI / (double)p->Nover[d];
    } else {
        q.fa = p->d_one; q.ka = 1;
        q.fk = p->d_invphi[dim];
        q.scale = 1.0;
    }
    NUFFT_HIP(launch_fft_lines(p->dtype, p->Nover[dim], true, q, stream));
    return NUFFT_OK;
}

static int pruned_backward_pass(nufft_plan* p, int c, int dim, const void* user_in, hipStream_t stream) {
    const size_t cb = 2 * real_bytes(p);
    const int64_t K1 = p->Nout[0];
    const int64_t S1 = p->spec_row;                              // row stride of the dimension-1 spectrum
    const int64_t T1 = p->compact_dim1 ? p->spec_row : K1;       // ... and of tmp2
    FftLinePass q{};
    q.map = p->d_index_map[dim];
    q.nk = (int)p->Nout[dim];
    q.twiddle = p->d_tw_bw[dim];
    q.scale = 1.0;
    const bool first = dim == p->D - 1;       // the pass that reads the caller's array
    if (dim == 2) {
        q.in = user_in;
        q.a_total = q.a_out = T1 * p->Nout[1];
        q.in_stride_j = K1 * p->Nout[1];
        q.in_stride_c = 0;
        q.nc = 1;
        q.out = p->d_tmp2;
        q.out_stride_j = T1 * p->Nout[1];
        q.out_stride_c = 0;
        if (T1 != K1) { q.row_a = (int)T1; q.row_valid = (int)K1; q.row_in = (int)K1; q.row_out = (int)T1; }
    } else {
        q.in = first ? user_in : p->d_tmp2;
        q.a_total = K1;
        q.a_out = p->compact_dim1 ? K1 : S1;   // general path: columns k1 >= N_out1 of the oversampled spectrum are written as zeros
        q.in_stride_j = first ? K1 : T1;
        q.in_stride_c = q.in_stride_j * p->Nout[1];
        q.nc = p->D == 3 ? (int)p->Nover[2] : 1;
        q.out = static_cast<char*>(p->d_uhat) + (size_t)c * p->pspec_elems * cb;
        q.out_stride_j = S1;
        q.out_stride_c = S1 * p->Nover[1];
    }
    q.mult = first ? p->cb_mode_factors : nullptr;
    if (first) { q.fa = p->d_invphi[0]; q.ka = (int)K1; } else { q.fa = p->d_one; q.ka = 1; }
    q.fk = p->d_invphi[dim];
    NUFFT_HIP(launch_fft_lines(p->dtype, p->Nover[dim], false, q, stream));
    return NUFFT_OK;
}

}  // namespace nufft

using namespace nufft;

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int nufft_version(void) { return NUFFT_MI355X_VERSION; }

const char* nufft_last_error_message(void) { return g_last_error.c_str(); }

const char* nufft_strerror(int code) {
    switch (code) {
        case NUFFT_OK: return "success";
        case NUFFT_ERR_INVALID_ARG: return "invalid argument (ArgumentError)";
        case NUFFT_ERR_SIZE_TOO_SMALL: return "data size is too small: sigma*N < 2M (ArgumentError)";
        case NUFFT_ERR_DIM_MISMATCH: return "dimension mismatch (DimensionMismatch)";
        case NUFFT_ERR_LDS_TOO_SMALL: return "LDS too small for the chosen problem (ArgumentError)";
        case NUFFT_ERR_UNSUPPORTED: return "unsupported configuration";
        case NUFFT_ERR_NO_POINTS: return "set_points has not been called";
        case NUFFT_ERR_ALLOC: return "device allocation failed";
        case NUFFT_ERR_HIP: return "HIP runtime error";
        case NUFFT_ERR_ROCFFT: return "rocFFT error";
        case NUFFT_ERR_NO_DEVICE: return "host-only plan has no device path";
        default: return "unknown error code";
    }
}

int nufft_plan_create_ex(nufft_plan** out, const nufft_params* params) {
    if (!out || !params) return fail(NUFFT_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    // The caller says how much of the struct it knows (struct_size = sizeof(nufft_params) of ITS header; 0 = the layout of ABI <= 102, which
    // ended before kernel_param_dim): only that prefix is read, the rest counts as zero — a binding built against an older header can
    // never make the library read past its struct.
    nufft_params prm;
    std::memset(&prm, 0, sizeof(prm));
    size_t known = params->struct_size > 0 ? (size_t)params->struct_size : offsetof(nufft_params, kernel_param_dim);
    if (known < offsetof(nufft_params, kernel_param_dim)) return fail(NUFFT_ERR_INVALID_ARG, "nufft_params.struct_size is smaller than any published layout");
    if (known > sizeof(prm)) known = sizeof(prm);
    std::memcpy(&prm, params, known);
    params = &prm;
    nufft_plan* p = new (std::nothrow) nufft_plan();
    if (!p) return fail(NUFFT_ERR_ALLOC, "out of host memory");
    int rc = build_host(p, params);
    if (rc == NUFFT_OK && p->device >= 0) rc = build_device(p);
    if (rc != NUFFT_OK) {
        const std::string keep = g_last_error;
        release(p);
        g_last_error = keep;
        return rc;
    }
    *out = p;
    return NUFFT_OK;
}

int nufft_plan_create(nufft_plan** out, int dtype, int is_complex, int ndim, const int64_t* N, int half_support,
                      double sigma, int kernel, int evalmode, int ntransforms, int fftshift, int point_transform,
                      int device) {
    if (!N) return fail(NUFFT_ERR_INVALID_ARG, "null dims");
    if (ndim < 1 || ndim > 3) return fail(NUFFT_ERR_UNSUPPORTED, "ndim must be 1, 2 or 3");
    nufft_params prm;
    std::memset(&prm, 0, sizeof(prm));
    prm.dtype = dtype;
    prm.is_complex = is_complex;
    prm.ndim = ndim;
    for (int d = 0; d < ndim; ++d) prm.N[d] = N[d];
    prm.half_support = half_support;
    prm.sigma = sigma;
    prm.kernel = kernel;
    prm.evalmode = evalmode;
    prm.ntransforms = ntransforms;
    prm.fftshift = fftshift;
    prm.point_transform = point_transform;
    prm.gpu_method = NUFFT_METHOD_SHARED_MEMORY;
    prm.device = device;
    prm.struct_size = (int32_t)sizeof(prm);
    return nufft_plan_create_ex(out, &prm);
}

const char* nufft_plan_options(const nufft_plan* p) {
    if (!p) return "";
    const_cast<nufft_plan*>(p)->opts_text = p->opts.str();
    return p->opts_text.c_str();
}

int nufft_workspace_breakdown(const nufft_plan* p, char* out, int64_t capacity) {
    if (!p || !out || capacity < 1) return fail(NUFFT_ERR_INVALID_ARG, "null argument");
    std::map<std::string, int64_t> sums;
    for (const auto& a : p->allocs) sums[a.second.first] += a.second.second;
    std::string text;
    for (const auto& e : sums) text += e.first + "=" + std::to_string(e.second) + ";";
    if ((int64_t)text.size() + 1 > capacity) return fail(NUFFT_ERR_DIM_MISMATCH, "buffer too small for the breakdown");
    std::memcpy(out, text.c_str(), text.size() + 1);
    return NUFFT_OK;
}

int nufft_plan_destroy(nufft_plan* plan) {
    release(plan);
    return NUFFT_OK;
}

int nufft_plan_info(const nufft_plan* p, nufft_info* o) {
    if (!p || !o) return fail(NUFFT_ERR_INVALID_ARG, "null argument");
    std::memset(o, 0, sizeof(*o));
    o->dtype = p->dtype;
    o->is_complex = p->is_complex;
    o->ndim = p->D;
    o->half_support = p->M;
    o->ntransforms = p->C;
    o->evalmode = p->evalmode;
    o->fftshift = p->fftshift;
    o->device = p->device;
    for (int d = 0; d < 3; ++d) {
        o->N[d] = d < p->D ? p->N[d] : 1;
        o->N_over[d] = d < p->D ? p->Nover[d] : 1;
        o->N_out[d] = d < p->D ? p->Nout[d] : 1;
        o->beta[d] = p->beta[d];
        o->bin_dims[d] = d < p->D ? 1 << p->tile.blog[d] : 1;
        o->nbins[d] = p->tile.nb[d];
        o->spread_tile[d] = p->tile.sp.n[d];
        o->spread_ntiles[d] = p->tile.sp.nt[d];
        o->interp_tile[d] = p->tile.ip.n[d];
        o->interp_ntiles[d] = p->tile.ip.nt[d];
        o->window_scale_log2[d] = d < p->D ? p->scale_exp[d] : 0;
    }
    o->sigma = p->sigma;
    o->spread_threads = p->spread_threads;
    o->interp_threads = p->interp_threads;
    o->lds_bytes_spread = p->lds_spread;
    o->lds_bytes_interp = p->lds_interp;
    o->workspace_bytes = p->workspace_bytes;
    o->num_points = p->Np;
    o->npoly = p->npoly;
    o->kernel = p->kernel;
    o->spread_max_items = p->tile.sp.max_items;
    o->interp_max_items = p->tile.ip.max_items;
    o->spread_method = p->spread_method;
    const bool patches = p->spread_method == NUFFT_SPREAD_MFMA_PATCHES;
    o->patch_dims[0] = patches ? 4 : 0;
    o->patch_dims[1] = patches ? p->patch.pby : 0;
    o->patch_f32acc = patches ? p->patch.f32acc : 0;
    o->patch_planar = patches ? p->patch.planar : 0;
    const bool ring = p->spread_method == NUFFT_SPREAD_MARCHING_RING;
    o->ring_column[0] = ring ? p->smarch.n1 : 0;
    o->ring_column[1] = ring ? p->smarch.n2 : 0;
    o->ring_segments = ring ? p->smarch.ct.nseg : 0;
    o->ring_halo = (ring && p->smarch.halo == 2) ? 1 : 0;
    // (host-only plans: what a device plan of these parameters would report — predict_sort_column)
    o->sort_column[0] = p->device < 0 ? p->sort_column_pred[0] : (p->coarse.enabled ? p->coarse.cbx : 0);
    o->sort_column[1] = p->device < 0 ? p->sort_column_pred[1] : (p->coarse.enabled ? p->coarse.cby : 0);
    return NUFFT_OK;
}

int nufft_plan_get_phi_hat(const nufft_plan* p, int dim, double* out, int64_t capacity) {
    if (!p || !out || dim < 0 || dim >= p->D) return fail(NUFFT_ERR_INVALID_ARG, "bad argument");
    if (capacity < (int64_t)p->phihat[dim].size()) return fail(NUFFT_ERR_DIM_MISMATCH, "buffer too small");
    std::memcpy(out, p->phihat[dim].data(), p->phihat[dim].size() * sizeof(double));
    return NUFFT_OK;
}

int nufft_plan_get_poly_coefs(const nufft_plan* p, int dim, double* out, int64_t capacity) {
    if (!p || !out || dim < 0 || dim >= p->D) return fail(NUFFT_ERR_INVALID_ARG, "bad argument");
    if (capacity < (int64_t)p->coefs[dim].size()) return fail(NUFFT_ERR_DIM_MISMATCH, "buffer too small");
    std::memcpy(out, p->coefs[dim].data(), p->coefs[dim].size() * sizeof(double));
    return NUFFT_OK;
}

int nufft_plan_get_index_map(const nufft_plan* p, int dim, int64_t* out, int64_t capacity) {
    if (!p || !out || dim < 0 || dim >= p->D) return fail(NUFFT_ERR_INVALID_ARG, "bad argument");
    if (capacity < (int64_t)p->index_map[dim].size()) return fail(NUFFT_ERR_DIM_MISMATCH, "buffer too small");
    std::memcpy(out, p->index_map[dim].data(), p->index_map[dim].size() * sizeof(int64_t));
    return NUFFT_OK;
}

// bytes per point of the {bin, rank} array of the fine sort; plans of the slab sort keep the records of its level 1 in the same allocation
// (a point set takes one sort or the other)
// (plans of the column-layer sort use neither unless a point set is clustered — the fine sort, 8 bytes — or dense — the slab sort, a record)
static int64_t binrank_bytes(const nufft_plan* p, bool slab_sort) {
    return slab_sort ? std::max<int64_t>(8, (int64_t)point_record_bytes(p->dtype, p->D)) : 8;
}

int nufft_set_points(nufft_plan* p, int64_t np, const void* const* coords, void* stream_) {
    int rc = require_device(p);
    if (rc) return rc;
    if (np < 0) return fail(NUFFT_ERR_INVALID_ARG, "negative number of points");
    if (np >= ((int64_t)1 << 31) - 1) return fail(NUFFT_ERR_UNSUPPORTED, "number of points exceeds 2^31 - 2");
    if (!coords) return fail(NUFFT_ERR_INVALID_ARG, "null coordinate table");
    for (int d = 0; d < p->D; ++d)
        if (np > 0 && !coords[d]) return fail(NUFFT_ERR_INVALID_ARG, "null coordinate vector");
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    p->Np = -1;      // no valid point set until everything below has been enqueued (a failed call leaves the plan without points)
    if (np > p->Np_capacity) {
        // resize_no_copy!, src/blocking/blocking.jl:55-61 (old contents are discarded).  hipFree / hipMalloc are not
        // capturable: pre-size the plan with the largest point set before capturing set_points in a hipGraph.
        dev_free(p, p->d_sorted);
        dev_free(p, p->d_vsorted);
        p->Np_capacity = 0;
        if (p->spread_method == NUFFT_SPREAD_MFMA_PATCHES && (rc = dev_alloc(p, &p->d_vsorted, (size_t)np * value_bytes(p) * p->C, "vsorted"))) return rc;
        if ((rc = dev_alloc(p, &p->d_sorted, (size_t)np * point_record_bytes(p->dtype, p->D), "sorted"))) return rc;
        p->Np_capacity = np;
    }
    // dense point set (mean load of the 4^3-cell bins): the matrix-pipe engine serves it where the ring does, from the fine-bin order
    p->dense_now = p->dense_available && np >= (int64_t)p->dense_min * p->tile.nbins;
    {
        // scratch of the sort this point set takes: {bin, rank} of the fine sort (8 bytes per point; also what a clustered set of a
        // column-layer plan falls back to, decided on the device), or the level-1 records of the slab sort
        // adaptive choice (nufft_internal.h: sort_feedback): what the rings decided for the point sets before this one
        if (p->sort_feedback) {
            hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
            const bool capturing = hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
            if (!capturing) (void)hipGetLastError();
            const uint32_t seq = __atomic_load_n(&p->sort_feedback[2], __ATOMIC_ACQUIRE);
            if (!capturing && seq != p->sort_seq_seen && seq == p->sort_seq) {      // the record of the latest point set has arrived: count it once
                p->sort_seq_seen = seq;
                const bool both = p->sort_feedback[0] != 0u && p->sort_feedback[1] != 0u;
                if (both) { p->sort_miss_streak = 0; p->sort_prefer_slab = false; }
                else if (++p->sort_miss_streak >= 2) p->sort_prefer_slab = true;
            }
        }
        p->coarse_now = p->coarse.enabled && !p->dense_now && !(p->sort_prefer_slab && p->slab.enabled);
        const bool slab_scratch = p->slab.enabled && !p->coarse_now;
        const int64_t need = np * binrank_bytes(p, slab_scratch);
        if (need > p->binrank_capacity) {
            dev_free(p, p->d_binrank);
            p->binrank_capacity = 0;
            if ((rc = dev_alloc(p, &p->d_binrank, (size_t)need, "sort_scratch"))) return rc;
            p->binrank_capacity = need;
        }
    }
    note_capture(p, stream);
    p->halo_hint = false;              // a deferred spread of the previous point set that was never consumed is void (the ring's task kernel
                                       // below clears the device word with the rest of its per-point-set record)
    StageTimer tm(p, NUFFT_STAGE_SET_POINTS, stream);
    SortArgs s{};
    s.dtype = p->dtype;
    s.D = p->D;
    s.point_transform = p->point_transform;
    s.np = np;
    for (int d = 0; d < 3; ++d) s.coords[d] = d < p->D ? coords[d] : nullptr;
    s.g = make_geom(p);
    s.counts = p->d_counts;
    s.counts_clean = p->counts_clean;
    p->counts_clean = false;
    s.offsets = p->d_offsets;
    s.binrank = p->d_binrank;
    s.sorted = p->d_sorted;
    s.scan_tmp = p->d_scan_tmp;
    s.scan_tmp_bytes = p->scan_tmp_bytes;
    s.cs = p->coarse;
    const bool coarse = p->coarse_now;
    if (p->coarse.enabled && !coarse) s.cs = CoarseSort{};
    if (coarse && p->sort_feedback) {      // (the scatter pass of the column-layer sort reports the rings' decisions: it runs behind their task kernels)
        s.cs.fb_a = p->coarse.flag_a; s.cs.fb_b = p->coarse.flag_b;
        s.cs.feedback = p->sort_feedback; s.cs.seq = ++p->sort_seq;
    }
    bool slab = false;
    if (!coarse && p->slab.enabled && np >= std::max<int64_t>(p->slab_min_points, 1)) {
        // slab height for this point set: the tallest slab (longest runs in level 1: C3 `set_points` 5.2 ms with 32 768 slabs, 4.4 ms with 16 384)
        // whose average load is at most 85 % of what a level-2 workgroup can hold (fuller slabs take its two-pass form), while there are
        // slabs enough to fill the chip
        const int capmax = slab_sort_capacity(p->dtype, kLdsLimit - 256);
        int best = -1;
        for (int sby = 1; sby <= p->tile.nb[1] && (int64_t)sby * p->tile.nb[0] <= kSlabMaxBins; sby *= 2) {
            const int64_t nkeys = (int64_t)p->tile.nb[2] * ((p->tile.nb[1] + sby - 1) / sby);
            if (nkeys > p->slab_max_keys) continue;
            if (np / nkeys > (int64_t)capmax * p->slab_fill / 100) break;
            if (best > 0 && nkeys < 2048) break;
            best = sby;
        }
        if (best > 0) {
            slab = true;
            s.cs = p->slab;
            s.cs.cbx = p->tile.nb[0]; s.cs.ncx = 1;
            s.cs.cby = best; s.cs.ncy = (p->tile.nb[1] + best - 1) / best;
            s.cs.nkeys = p->tile.nb[2] * s.cs.ncy;
            s.cs.groups = (int)std::min<int64_t>(p->num_cus, std::max<int64_t>(1, (np + 16383) / 16384));
            const int64_t mean = np / s.cs.nkeys;
            s.cs.cap = (int)std::min<int64_t>(capmax, (std::max<int64_t>(2 * mean, mean + 1536) + 63) / 64 * 64);
            s.cs.lds2 = slab_sort_lds_bytes(p->dtype, s.cs.cap);
            s.cs.temp = p->d_binrank;
        }
    }
    // plans of the column-layer sort: histogram by column layers first — its offsets are all the task kernels below need to decide
    // whether both rings serve this point set; the scatter pass (of whichever sort that decision selects) and the tile tables follow
    if (coarse) NUFFT_HIP(launch_binsort_coarse_count(s, stream));
    else if (slab) {
        // (nothing else decides here: the fullest slab does, on the device — both halves back to back)
        NUFFT_HIP(launch_binsort_coarse_count(s, stream));
        NUFFT_HIP(launch_binsort_coarse_finish(s, stream));
    } else {
        if (p->slab.enabled) NUFFT_HIP(launch_zero_fill(p->slab.flagmem + 4, 16, stream));      // (what nufft_sort_columns_used reports)
        NUFFT_HIP(launch_binsort(s, stream));
    }
    // slices per tile from the work each tile now holds (balance.hip)
    auto balance = [&]() -> int {
        const nufft_plan::Balance& b = p->bal;
        BalanceArgs q{};
        q.g = s.g;
        q.D = p->D;
        q.M = p->M;
        q.enabled = p->balance_enabled && np > 0;
        q.extra_sp = b.extra[0];
        q.extra_ip = b.extra[1];
        q.smax = 4096;
        q.offsets = p->d_offsets;
        q.work = b.d_work;
        q.nslices = b.d_nslices;
        q.desc_off = b.d_desc_off;
        q.desc = static_cast<uint2*>(b.d_desc);
        q.slots_in_use = b.d_slots;
        q.scan_tmp = b.d_tmp;
        q.scan_tmp_bytes = b.tmp_bytes;
        q.skip_a = coarse ? p->coarse.flag_a : nullptr;
        q.skip_b = coarse ? p->coarse.flag_b : nullptr;
        q.sp_served = coarse ? p->coarse.flag_a : nullptr;      // (otherwise the task kernels below clear the spreading slots)
        NUFFT_HIP(launch_balance(q, stream));
        return NUFFT_OK;
    };
    if (!coarse && (rc = balance())) return rc;
    if (p->spread_method == NUFFT_SPREAD_MFMA_PATCHES) {
        // which engine serves this point set: decided on the device from the heaviest patch task (balance.hip)
        PatchPlan pp{};
        pp.eligible = true;
        pp.npx = p->patch.npx; pp.npy = p->patch.npy; pp.nseg = p->patch.nseg; pp.segl = p->patch.segl;
        pp.ntasks = p->patch.ntasks; pp.lds_bytes = p->patch.lds_bytes; pp.pby = p->patch.pby; pp.f32acc = p->patch.f32acc; pp.planar = p->patch.planar;
        // the patches' measured advantage over the LDS tiles on uniform points grows with the matrix work per point visit
        // (ComplexF64 m = 4 1.17, m = 7 3.7; Float64 m = 6 1.5-1.7, m = 9 3.7; ComplexF32 on the FP32
        // matrix pipe m = 8 3.4-4): 1.15 x g^(m - 4) with g = 1.2 (real) / 1.45 (complex), x 1.3 with Float32 accumulators,
        // x 0.85 because the estimate counts a task's own column only and no scheduling tail.  An explicit request always
        // takes the patches.
        double advantage = 0.85 * 1.15 * std::pow(p->is_complex ? 1.45 : 1.2, (double)std::max(p->M - 4, 0)) * (p->patch.f32acc ? 1.3 : 1.0);
        if (p->spread_method_req == NUFFT_SPREAD_MFMA_PATCHES) advantage = 0.0;
        const int slots = p->wave_slots;
        const int clo = -((p->M + 2) / 4), chi = (3 + p->M) / 4;          // cubes a stencil reaches beside its bin (PatchCfg::CLO, CHI)
        const size_t ncols = (size_t)pp.npx * pp.npy;
        NUFFT_HIP(launch_patch_tasks(s.g, pp, clo, chi, p->d_offsets, np, slots, advantage, p->d_patch_choice, p->bal.d_slots, p->d_patch_cols,
                                     p->d_patch_cols + ncols, static_cast<uint2*>(p->d_patch_tasks), stream));
    }
    if (p->spread_method == NUFFT_SPREAD_MARCHING_RING) {
        // tasks of the spreading ring for this point set, and whether it serves it (balance.hip): while the heaviest task stays
        // within its measured advantage over the LDS tiles on uniform points (x 0.85 for what the estimate leaves out); an
        // explicit request always takes the ring
        // (halo variant: 1.87 instead of 2.44 ms at C2, + 0.15 ms in the FFT pass: 3.67 / 2.62 = 1.4 against the tiles, stage + FFT; the
        // reference's benchmark distribution — folded N(0, 1), sigma = 1.5, Np = 1.68e7 — sits at the threshold: tiles 6.78 + 0.32 ms
        // below 1.19, window 6.18 + 0.38 ms from 1.3 on, hence 0.93 instead of 0.85 here)
        const double adv_env = option_double("NUFFT_SMARCH_ADVANTAGE", 0.0);
        double advantage = p->smarch.halo == 2 ? 0.93 * 1.4 : 0.85 * 1.3;
        if (adv_env > 0.0) advantage = adv_env;       // (experiments)
        if (p->spread_method_req == NUFFT_SPREAD_MARCHING_RING) advantage = 0.0;
        const size_t ncols = (size_t)p->smarch.ct.ncolx * p->smarch.ct.ncoly;
        // (the C components are independent workgroups of one launch: each component has num_cus / C compute units' worth of the chip)
        NUFFT_HIP(launch_smarch_tasks(s.g, p->smarch, p->d_offsets, np, std::max(1, p->num_cus / (p->C * p->smarch.parts)), advantage, p->d_smarch_choice, p->bal.d_slots, p->d_smarch_cols,
                                      p->d_smarch_cols + ncols, static_cast<uint2*>(p->d_smarch_tasks), stream));
    }
    if (p->interp_march) {
        // tasks of the interpolation ring for this point set, and whether it serves it (balance.hip).  What the ring saves
        // over interp_tile_kernel is grid traffic (halo 1.5x instead of 2.6x at m = 4, 3.75x instead of 12.5x at m = 8), which
        // it also takes off the critical path; its gather itself is ~9 % slower per point (ring index arithmetic, two barriers
        // per bin layer).  So its advantage falls with the point density rho = points per oversampled cell,
        //   advantage = (rho + c_t) / (1.09 rho + c_r),  c_t = 0.038 (m/4)^1.85,  c_r = 0.0086 (m/4)^1.2
        // fitted to: C2 (m = 4, rho = 0.075) 1.26, Float64 m = 6 1.55, C3 (m = 8, rho = 0.093) 1.9, and Float64 m = 4 at
        // rho = 0.3 (1.68e7 points on 384^3) 1.0 — above that density the tile kernel is the faster one even for uniform points.
        // (NUFFT_INTERP_MARCH=2: always the ring — tests of its instantiations on small grids)
        const double rho = (double)np / (double)std::max<int64_t>(p->grid_elems, 1), mr = (double)p->M / 4.0;
        const double fitted = (rho + 0.038 * std::pow(mr, 1.85)) / (1.09 * rho + 0.0086 * std::pow(mr, 1.2));
        const double advantage = p->interp_march_mode == 2 ? 0.0 : fitted;      // (<= 0: always the ring)
        // Point sets far from uniform (tasks of equal point count): the same trade at the density the points themselves see,
        // rho_eff = sum n^2 / (sum n x cells) over the column layers (balance.hip).  Uniform sets break even at
        // rho* = (c_t - c_r) / 0.09 (0.33 at m = 4); cut tasks amortise the window loads better — measured at m = 4, Np = 1e7 ... 1.7e7:
        // rho_eff = 0.42 (folded N(0, 1) on 512^3, Gaussian cluster of 1 rad) ring 1.87 / 1.88 ms against 2.10 / 2.09 tiles;
        // 1.66 (folded N(0, 1), 1.68e7 points on 384^3) 3.42 against 2.58; 3.3 (cluster of 0.5 rad) 2.01 against 1.43 — so the ring
        // keeps such sets up to 2.5 rho*.
        const double rho_star = (0.038 * std::pow(mr, 1.85) - 0.0086 * std::pow(mr, 1.2)) / 0.09;
        const size_t ncols = (size_t)p->march_ct.ncolx * p->march_ct.ncoly;
        NUFFT_HIP(launch_march_tasks(s.g, p->march_ct, p->d_offsets, np, p->num_cus, advantage, 2.5 * rho_star, p->d_march_choice, p->d_march_cols,
                                     p->d_march_cols + ncols, static_cast<uint2*>(p->d_march_tasks), stream));
    }
    if (coarse) {
        // both rings have decided: column-layer scatter, or the fine sort for a point set one of them hands to the tile kernels
        NUFFT_HIP(launch_binsort_coarse_finish(s, stream));
        if ((rc = balance())) return rc;
    } else if (p->sort_feedback && p->coarse.enabled && !p->dense_now) {
        // a plan of the column-layer sort on the slab sort (the sets before this one went to the tile kernels): what the rings say about THIS set
        NUFFT_HIP(launch_sort_feedback(p->coarse.flag_a, p->coarse.flag_b, p->sort_feedback, ++p->sort_seq, stream));
    }
    if (p->debug_tasks) {
        // development check: every column's tasks tile [0, nb[2]) without gaps or overlaps
        auto check = [&](const char* name, const ColumnTasks& ct, const void* tab, const uint32_t* choice) {
            const size_t ncols = (size_t)ct.ncolx * ct.ncoly, ntab = (size_t)column_task_table_entries(ct, p->tile.nb[2]);
            std::vector<uint32_t> h(2 * ntab), ch(8);
            (void)hipStreamSynchronize(stream);
            (void)hipMemcpy(h.data(), tab, ntab * 8, hipMemcpyDeviceToHost);
            (void)hipMemcpy(ch.data(), choice, 32, hipMemcpyDeviceToHost);
            std::vector<std::vector<std::pair<int, int>>> segs(ncols);
            size_t used = 0, maxlen = 0;
            for (size_t t = 0; t < ntab; ++t) {
                const uint32_t c = h[2 * t], y = h[2 * t + 1];
                if (!y) continue;
                ++used;
                if (c >= ncols) { fprintf(stderr, "[tasks %s] entry %zu: column %u out of range\n", name, t, c); continue; }
                segs[c].push_back({(int)(y & 0xffffu), (int)(y >> 16)});
                maxlen = std::max(maxlen, (size_t)((y >> 16) - (y & 0xffffu)));
            }
            size_t bad = 0;
            for (size_t c = 0; c < ncols; ++c) {
                std::sort(segs[c].begin(), segs[c].end());
                int z = 0;
                for (auto& sg : segs[c]) { if (sg.first != z) ++bad; z = sg.second; }
                if (z != p->tile.nb[2]) ++bad;
            }
            fprintf(stderr, "[tasks %s] columns %zu, table %zu, in use %zu, longest %zu layers, flag %u, uniform %u, bad columns %zu\n", name, ncols, ntab,
                    used, maxlen, ch[2], ch[3], bad);
        };
        if (p->spread_method == NUFFT_SPREAD_MFMA_PATCHES) {
            ColumnTasks ct{p->patch.npx, p->patch.npy, 4, p->patch.pby, p->patch.nseg, p->patch.segl, p->patch.ntasks, 1, 0, 0, 0};
            check("patches", ct, p->d_patch_tasks, p->d_patch_choice);
        }
        if (p->interp_march) check("ring", p->march_ct, p->d_march_tasks, p->d_march_choice);
        if (p->spread_method == NUFFT_SPREAD_MARCHING_RING) check("spreading ring", p->smarch.ct, p->d_smarch_tasks, p->d_smarch_choice);
    }
    p->Np = np;
    p->counts_clean = np > 0;      // the scatter pass has cleared the histogram (no point, no scatter pass: cleared next time)
    return NUFFT_OK;
}

int nufft_spread_engine_used(nufft_plan* p, int* engine_out, void* stream_) {
    int rc = require_points(p);
    if (rc) return rc;
    if (!engine_out) return fail(NUFFT_ERR_INVALID_ARG, "null output");
    *engine_out = NUFFT_SPREAD_LDS_TILES;
    if (p->spread_method != NUFFT_SPREAD_MFMA_PATCHES && p->spread_method != NUFFT_SPREAD_MARCHING_RING) return NUFFT_OK;
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    uint32_t flag = 0;
    const uint32_t* src = p->spread_method == NUFFT_SPREAD_MFMA_PATCHES ? p->d_patch_choice : p->d_smarch_choice;
    NUFFT_HIP(hipMemcpyAsync(&flag, src + 2, sizeof(flag), hipMemcpyDeviceToHost, stream));
    NUFFT_HIP(hipStreamSynchronize(stream));
    *engine_out = flag ? p->spread_method : NUFFT_SPREAD_LDS_TILES;
    if (flag && p->spread_method == NUFFT_SPREAD_MARCHING_RING && p->dense_now) *engine_out = NUFFT_SPREAD_MARCHING_RING_DENSE;
    return NUFFT_OK;
}

int nufft_sort_columns_used(nufft_plan* p, int* used_out, void* stream_) {
    int rc = require_points(p);
    if (rc) return rc;
    if (!used_out) return fail(NUFFT_ERR_INVALID_ARG, "null output");
    *used_out = 0;
    if (!p->coarse.enabled && !p->slab.enabled) return NUFFT_OK;
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    uint32_t fa = 0, fb = 0;
    if (p->slab.enabled && !p->coarse_now) {
        NUFFT_HIP(hipMemcpyAsync(&fa, p->slab.flag_a, sizeof(fa), hipMemcpyDeviceToHost, stream));
        NUFFT_HIP(hipStreamSynchronize(stream));
        *used_out = fa != 0 ? 2 : 0;
        return NUFFT_OK;
    }
    NUFFT_HIP(hipMemcpyAsync(&fa, p->coarse.flag_a, sizeof(fa), hipMemcpyDeviceToHost, stream));
    NUFFT_HIP(hipMemcpyAsync(&fb, p->coarse.flag_b, sizeof(fb), hipMemcpyDeviceToHost, stream));
    NUFFT_HIP(hipStreamSynchronize(stream));
    *used_out = (fa != 0 && fb != 0) ? 1 : 0;
    return NUFFT_OK;
}

int nufft_interp_engine_used(nufft_plan* p, int* engine_out, void* stream_) {
    int rc = require_points(p);
    if (rc) return rc;
    if (!engine_out) return fail(NUFFT_ERR_INVALID_ARG, "null output");
    *engine_out = NUFFT_INTERP_LDS_TILES;
    if (!p->interp_march) return NUFFT_OK;
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    uint32_t flag = 0;
    NUFFT_HIP(hipMemcpyAsync(&flag, p->d_march_choice + 2, sizeof(flag), hipMemcpyDeviceToHost, stream));
    NUFFT_HIP(hipStreamSynchronize(stream));
    if (flag) *engine_out = NUFFT_INTERP_MARCHING_RING;
    return NUFFT_OK;
}

int64_t nufft_sizeof_params(void) { return (int64_t)sizeof(nufft_params); }
int64_t nufft_sizeof_info(void) { return (int64_t)sizeof(nufft_info); }

int nufft_fill_zeros(nufft_plan* p, void* stream_) {
    int rc = require_device(p);
    if (rc) return rc;
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer tm(p, NUFFT_STAGE_T1_ZERO, stream);
    note_capture(p, stream);
    if ((rc = void_halo(p, stream))) return rc;
    NUFFT_HIP(launch_zero_fill(p->d_us, (size_t)p->grid_elems * value_bytes(p) * p->C, stream));
    return NUFFT_OK;
}

// defer_halo: exec_type1 with the halo variant of the ring on a plan whose first FFT pass adds the side buffer itself
static int spread_impl(nufft_plan* p, const void* const* values_in, void* stream_, bool defer_halo) {
    int rc = require_points(p);
    if (rc) return rc;
    if (!values_in) return fail(NUFFT_ERR_INVALID_ARG, "null value table");
    for (int c = 0; c < p->C; ++c)
        if (p->Np > 0 && !values_in[c]) return fail(NUFFT_ERR_INVALID_ARG, "null value vector");
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer tm(p, NUFFT_STAGE_T1_SPREAD, stream);
    note_capture(p, stream);
    TileKernelArgs a = tile_args(p, false);
    a.values_in = values_in;
    // LDS-tile engine.  On plans of the MFMA-patch engine set_points has decided on the device which of the two serves
    // this point set (balance.hip): the other one finds no work (no slots / flag = 0) and returns at once.
    if (p->balance_enabled)    // tiles shared by several workgroups accumulate with atomics: zero them first
        NUFFT_HIP(launch_zero_split_tiles(p->dtype, a.g, p->D, p->is_complex, p->C, p->bal.d_nslices, p->d_us,
                                          p->grid_elems * (p->is_complex ? 2 : 1), stream));
    NUFFT_HIP(launch_spread(a, stream));
    if (p->spread_method == NUFFT_SPREAD_MARCHING_RING) {
        // (halo variant: the kernel sets the device word "side buffer pending")
        NUFFT_HIP(launch_spread_march(a, p->smarch, p->d_smarch_choice + 2, static_cast<const uint2*>(p->d_smarch_tasks), p->d_smarch_choice + kHaloStateWord, p->dense_now, stream));
        // halo variant: the grid is complete once the side buffer has been added — here, or by the first FFT pass
        if (halo_plan(p)) {
            p->halo_hint = true;
            if (!defer_halo && (rc = complete_halo(p, stream))) return rc;
        }
    }
    if (p->spread_method == NUFFT_SPREAD_MFMA_PATCHES) {
        // values gathered into sorted order (per-point weights of the callback menu folded in), then the patches
        const uint32_t* enabled = p->d_patch_choice + 2;
        const int64_t vstride = p->Np * (p->is_complex ? 2 : 1);
        if (p->patch.planar)
            NUFFT_HIP(launch_gather_planar(p->dtype, p->D, p->d_sorted, p->Np, values_in, p->C, p->cb_point_weights, p->d_vsorted, enabled, stream));
        else
            for (int c = 0; c < p->C; ++c)
                NUFFT_HIP(launch_gather_values(p->dtype, p->is_complex, p->D, p->d_sorted, p->Np, values_in[c], p->cb_point_weights,
                                               static_cast<char*>(p->d_vsorted) + (size_t)c * vstride * real_bytes(p), enabled, stream));
        PatchPlan pp{};
        pp.eligible = true;
        pp.npx = p->patch.npx; pp.npy = p->patch.npy; pp.nseg = p->patch.nseg; pp.segl = p->patch.segl;
        pp.ntasks = p->patch.ntasks; pp.lds_bytes = p->patch.lds_bytes; pp.pby = p->patch.pby; pp.f32acc = p->patch.f32acc; pp.planar = p->patch.planar;
        NUFFT_HIP(launch_spread_patch(a, pp, p->d_vsorted, vstride, enabled, static_cast<const uint2*>(p->d_patch_tasks), stream));
    }
    return NUFFT_OK;
}

int nufft_spread(nufft_plan* p, const void* const* values_in, void* stream_) { return spread_impl(p, values_in, stream_, false); }
int nufft_spread_deferred(nufft_plan* p, const void* const* values_in, void* stream_) { return spread_impl(p, values_in, stream_, true); }

int nufft_fft_forward(nufft_plan* p, void* stream_) {
    int rc = require_device(p);
    if (rc) return rc;
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer tm(p, NUFFT_STAGE_T1_FFT, stream);
    // behind nufft_spread_deferred: the dimension-1 pass of a real plan's own FFT adds the side buffer while it loads its lines;
    // every other FFT path gets the completed grid
    // (the fused pass adds the buffer to the lines it has loaded, not to us: halo_pending stays set — us still lacks the reach, the
    // side buffer stays valid until the next spread / set_points — and nufft_complete_grid / _copy_grid / _interpolate add it on demand)
    note_capture(p, stream);
    const bool fuse_halo = halo_maybe_pending(p) && p->pruned_fft && p->compact_dim1 && p->D == 3 && p->halo_fuse;      // (compact_dim1: true for complex plans)
    if (!fuse_halo && (rc = complete_halo(p, stream))) return rc;
    if (p->pruned_fft) {
        if ((rc = pruned_forward_fft(p, stream, fuse_halo))) return rc;
        if (p->D == 3 && p->C == 1) return pruned_forward_pass(p, 0, 1, nullptr, stream);
        return NUFFT_OK;      // C > 1: tmp2 is reused per component, both passes run in the deconvolution stage
    }
    return fft_exec(p, true, stream);
}

int nufft_deconvolve_truncate(nufft_plan* p, void* const* uhat_out, void* stream_) {
    int rc = require_device(p);
    if (rc) return rc;
    if (!uhat_out) return fail(NUFFT_ERR_INVALID_ARG, "null output table");
    for (int c = 0; c < p->C; ++c)
        if (!uhat_out[c]) return fail(NUFFT_ERR_INVALID_ARG, "null output array");
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer tm(p, NUFFT_STAGE_T1_DECONV, stream);
    if (p->pruned_fft) {
        for (int c = 0; c < p->C; ++c) {
            if (p->D == 3 && p->C > 1 && (rc = pruned_forward_pass(p, c, 1, nullptr, stream))) return rc;
            if ((rc = pruned_forward_pass(p, c, p->D - 1, uhat_out[c], stream))) return rc;
        }
        return NUFFT_OK;
    }
    DeconvArgs a = deconv_args(p);
    a.normfactor = 1.0;
    for (int d = 0; d < p->D; ++d) a.normfactor *= 2.0 * M_PI / (double)p->Nover[d];   // src/NonuniformFFTs.jl:181
    NUFFT_HIP(launch_deconv_truncate(a, uhat_out, stream));
    return NUFFT_OK;
}

int nufft_deconvolve_pad(nufft_plan* p, const void* const* uhat_in, void* stream_) {
    int rc = require_device(p);
    if (rc) return rc;
    if (!uhat_in) return fail(NUFFT_ERR_INVALID_ARG, "null input table");
    for (int c = 0; c < p->C; ++c)
        if (!uhat_in[c]) return fail(NUFFT_ERR_INVALID_ARG, "null input array");
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer tm(p, NUFFT_STAGE_T2_DECONV, stream);
    note_capture(p, stream);
    if ((rc = void_halo(p, stream))) return rc;      // type 2 starts: the grids will be overwritten (complex plans of the general path: by this stage)
    if (p->pruned_fft) {
        // zero-padding + deconvolution are fused into the pruned inverse passes; tmp2 is reused per component
        for (int c = 0; c < p->C; ++c) {
            if (p->D == 3 && (rc = pruned_backward_pass(p, c, 2, uhat_in[c], stream))) return rc;
            if ((rc = pruned_backward_pass(p, c, 1, uhat_in[c], stream))) return rc;
        }
        return NUFFT_OK;
    }
    DeconvArgs a = deconv_args(p);
    NUFFT_HIP(launch_deconv_pad(a, uhat_in, stream));
    return NUFFT_OK;
}

int nufft_fft_backward(nufft_plan* p, void* stream_) {
    int rc = require_device(p);
    if (rc) return rc;
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer tm(p, NUFFT_STAGE_T2_FFT, stream);
    note_capture(p, stream);
    if ((rc = void_halo(p, stream))) return rc;      // the grids are overwritten: an unconsumed deferred spread is void
    if (p->pruned_fft && p->is_complex) {
        int64_t per = 1;
        for (int d = 1; d < p->D; ++d) per *= p->Nover[d];
        const size_t cb = 2 * real_bytes(p);
        for (int c = 0; c < p->C; ++c) {
            const void* in = static_cast<char*>(p->d_uhat) + (size_t)c * p->pspec_elems * cb;
            void* out = static_cast<char*>(p->d_us) + (size_t)c * p->grid_elems * cb;
            NUFFT_HIP(launch_cplx_lines(p->dtype, p->Nover[0], false, in, out, per, (int)p->Nout[0], p->d_index_map[0], p->d_tw_bw[0], stream));
        }
        return NUFFT_OK;
    }
    if (p->pruned_fft && p->compact_dim1) {
        int64_t per = 1;
        for (int d = 1; d < p->D; ++d) per *= p->Nover[d];
        const size_t rb = real_bytes(p);
        for (int c = 0; c < p->C; ++c) {
            const void* in = static_cast<char*>(p->d_uhat) + (size_t)c * p->pspec_elems * 2 * rb;
            void* out = static_cast<char*>(p->d_us) + (size_t)c * p->grid_elems * rb;
            NUFFT_HIP(launch_real_lines(p->dtype, p->Nover[0], false, in, out, per, (int)p->Nout[0], (int)p->spec_row, p->d_tw_bw[0], stream));
        }
        return NUFFT_OK;
    }
    if (p->pruned_fft) {
        NUFFT_ROCFFT(rocfft_execution_info_set_stream(p->fft_info, stream));
        void* in[1] = {p->d_uhat};
        void* out[1] = {p->d_us};
        NUFFT_ROCFFT(rocfft_execute(p->fft1_bw, in, out, p->fft_info));
        return NUFFT_OK;
    }
    return fft_exec(p, false, stream);
}

int nufft_interpolate(nufft_plan* p, void* const* values_out, void* stream_) {
    int rc = require_points(p);
    if (rc) return rc;
    if (!values_out) return fail(NUFFT_ERR_INVALID_ARG, "null output table");
    for (int c = 0; c < p->C; ++c)
        if (p->Np > 0 && !values_out[c]) return fail(NUFFT_ERR_INVALID_ARG, "null output vector");
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    StageTimer tm(p, NUFFT_STAGE_T2_INTERP, stream);
    note_capture(p, stream);
    if ((rc = complete_halo(p, stream))) return rc;
    if (p->Np == 0) return NUFFT_OK;
    TileKernelArgs a = tile_args(p, true);
    a.values_out = values_out;
    NUFFT_HIP(launch_interp(a, stream));
    return NUFFT_OK;
}

int nufft_exec_type1(nufft_plan* p, void* const* uhat_out, const void* const* values_in, void* stream) {
    int rc = require_points(p);
    if (rc) return rc;
    if (!uhat_out || !values_in) return fail(NUFFT_ERR_INVALID_ARG, "null argument");
    // (0) "Fill with zeros" (src/NonuniformFFTs.jl:161-167) is not needed: the output-driven spreading
    // kernel stores every grid cell exactly once.
    // (halo variant of the spreading ring: the FFT stage completes the grid — its dimension-1 pass adds the side buffer on the fly)
    if ((rc = nufft_spread_deferred(p, values_in, stream))) return rc;  // (1) :169-172
    if ((rc = nufft_fft_forward(p, stream))) return rc;                // (2) :174-177
    return nufft_deconvolve_truncate(p, uhat_out, stream);             // (3) :179-185
}

// Fused callback menu: the two documented uses of NUFFTCallbacks (src/plan.jl:105-143, test/callbacks.jl:17-25)
// that can cross a C ABI.  The pointers are only read while the call enqueues its kernels.
namespace {
struct CallbackScope {
    nufft_plan* p;
    CallbackScope(nufft_plan* plan, const nufft_callbacks* cb) : p(plan) {
        p->cb_point_weights = cb ? cb->point_weights : nullptr;
        p->cb_mode_factors = cb ? cb->mode_factors : nullptr;
    }
    ~CallbackScope() { p->cb_point_weights = p->cb_sticky_weights; p->cb_mode_factors = p->cb_sticky_factors; }
};
}  // namespace

// The callback menu for the STAGE-level entry points (nufft_spread[_deferred], nufft_deconvolve_truncate, nufft_deconvolve_pad,
// nufft_interpolate): in force until the next call (NULL: none).  A binding that enqueues the stages one by one — as the reference's
// exec_type1! / exec_type2! do, each under its own timer label (src/NonuniformFFTs.jl:157-186, 246-283) — brackets them with this.
int nufft_set_callbacks(nufft_plan* p, const nufft_callbacks* cb) {
    if (!p) return fail(NUFFT_ERR_INVALID_ARG, "null plan");
    p->cb_sticky_weights = p->cb_point_weights = cb ? cb->point_weights : nullptr;
    p->cb_sticky_factors = p->cb_mode_factors = cb ? cb->mode_factors : nullptr;
    return NUFFT_OK;
}

int nufft_exec_type1_cb(nufft_plan* p, void* const* uhat_out, const void* const* values_in, const nufft_callbacks* cb,
                        void* stream) {
    if (!p) return fail(NUFFT_ERR_INVALID_ARG, "null plan");
    CallbackScope scope(p, cb);
    return nufft_exec_type1(p, uhat_out, values_in, stream);
}

int nufft_exec_type2_cb(nufft_plan* p, void* const* values_out, const void* const* uhat_in, const nufft_callbacks* cb,
                        void* stream) {
    if (!p) return fail(NUFFT_ERR_INVALID_ARG, "null plan");
    CallbackScope scope(p, cb);
    return nufft_exec_type2(p, values_out, uhat_in, stream);
}

int nufft_exec_type2(nufft_plan* p, void* const* values_out, const void* const* uhat_in, void* stream) {
    int rc = require_points(p);
    if (rc) return rc;
    if (!values_out || !uhat_in) return fail(NUFFT_ERR_INVALID_ARG, "null argument");
    if ((rc = nufft_deconvolve_pad(p, uhat_in, stream))) return rc;    // (0)+(1) :260-272
    if ((rc = nufft_fft_backward(p, stream))) return rc;               // (2) :274-277
    return nufft_interpolate(p, values_out, stream);                   // (3) :279-282
}

int nufft_grid_ptr(const nufft_plan* p, int which, int component, void** out_ptr, int64_t* out_bytes) {
    int rc = require_device(p);
    if (rc) return rc;
    if (!out_ptr || component < 0 || component >= p->C) return fail(NUFFT_ERR_INVALID_ARG, "bad argument");
    if (which == 0) {
        if (halo_maybe_pending(p))
            return fail(NUFFT_ERR_INVALID_ARG, "us lacks the side buffer of a deferred spread (nufft_spread_deferred / nufft_exec_type1 on a "
                                               "ring_halo plan): call nufft_complete_grid first, or use nufft_copy_grid");
        const size_t bytes = (size_t)p->grid_elems * value_bytes(p);
        *out_ptr = static_cast<char*>(p->d_us) + bytes * component;
        if (out_bytes) *out_bytes = (int64_t)bytes;
    } else if (which == 1 && !p->is_complex) {
        const size_t bytes = (size_t)p->pspec_elems * 2 * real_bytes(p);      // (the compact spectrum on plans of the library's own FFT passes)
        *out_ptr = static_cast<char*>(p->d_uhat) + bytes * component;
        if (out_bytes) *out_bytes = (int64_t)bytes;
    } else {
        return fail(NUFFT_ERR_INVALID_ARG, "which must be 0 (us) or 1 (ûs, real plans only)");
    }
    return NUFFT_OK;
}

int nufft_complete_grid(nufft_plan* p, void* stream_) {
    int rc = require_device(p);
    if (rc) return rc;
    DeviceGuard guard(p->device);
    note_capture(p, static_cast<hipStream_t>(stream_));
    return complete_halo(p, static_cast<hipStream_t>(stream_));
}

int nufft_copy_grid(nufft_plan* p, int which, int component, void* dst, int64_t capacity_bytes, void* stream_) {
    int rc = require_device(p);
    if (rc) return rc;
    // every argument is checked before anything is enqueued or any plan state changes
    if (component < 0 || component >= p->C) return fail(NUFFT_ERR_INVALID_ARG, "bad argument");
    if (which != 0 && !(which == 1 && !p->is_complex)) return fail(NUFFT_ERR_INVALID_ARG, "which must be 0 (us) or 1 (ûs, real plans only)");
    if (!dst) return fail(NUFFT_ERR_INVALID_ARG, "null destination");
    const size_t bytes = which == 0 ? (size_t)p->grid_elems * value_bytes(p) : (size_t)p->pspec_elems * 2 * real_bytes(p);
    if (capacity_bytes < (int64_t)bytes) return fail(NUFFT_ERR_DIM_MISMATCH, "destination buffer too small");
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (which == 0) {
        note_capture(p, stream);
        if ((rc = complete_halo(p, stream))) return rc;
    }
    const char* src = static_cast<const char*>(which == 0 ? p->d_us : p->d_uhat) + bytes * (size_t)component;
    NUFFT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
    return NUFFT_OK;
}

int nufft_get_sort_result(nufft_plan* p, int32_t* perm_host, int64_t perm_capacity, uint32_t* tile_offsets_host,
                          int64_t offsets_capacity, void* stream_) {
    int rc = require_points(p);
    if (rc) return rc;
    DeviceGuard guard(p->device);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (perm_host) {
        if (perm_capacity < p->Np) return fail(NUFFT_ERR_DIM_MISMATCH, "perm buffer too small");
        if (p->Np > 0) {
            int32_t* tmp = nullptr;
            NUFFT_HIP(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)p->Np * sizeof(int32_t)));
            hipError_t e = launch_extract_perm(p->dtype, p->D, p->d_sorted, p->Np, tmp, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(perm_host, tmp, (size_t)p->Np * sizeof(int32_t), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            (void)hipFree(tmp);
            NUFFT_HIP(e);
        }
    }
    if (tile_offsets_host) {
        const int64_t n = p->tile.nbins + 1;
        if (offsets_capacity < n) return fail(NUFFT_ERR_DIM_MISMATCH, "offset buffer too small");
        NUFFT_HIP(hipMemcpyAsync(tile_offsets_host, p->d_offsets, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        NUFFT_HIP(hipStreamSynchronize(stream));
    }
    return NUFFT_OK;
}

int nufft_set_timing(nufft_plan* p, int enable) {
    int rc = require_device(p);
    if (rc) return rc;
    p->timing = enable != 0;
    if (!p->timing)
        for (int s = 0; s < NUFFT_NUM_STAGES; ++s) p->ev_valid[s] = false;
    return NUFFT_OK;
}

int nufft_get_stage_times(nufft_plan* p, float* ms_out) {
    int rc = require_device(p);
    if (rc) return rc;
    if (!ms_out) return fail(NUFFT_ERR_INVALID_ARG, "null output");
    DeviceGuard guard(p->device);
    for (int s = 0; s < NUFFT_NUM_STAGES; ++s) {
        ms_out[s] = -1.0f;
        if (!p->ev_valid[s]) continue;
        hipEvent_t a = static_cast<hipEvent_t>(p->ev_begin[s]);
        hipEvent_t b = static_cast<hipEvent_t>(p->ev_end[s]);
        NUFFT_HIP(hipEventSynchronize(b));
        float ms = 0.f;
        NUFFT_HIP(hipEventElapsedTime(&ms, a, b));
        ms_out[s] = ms;
    }
    return NUFFT_OK;
}

}  // extern "C"
