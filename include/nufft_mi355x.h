/*
 * nufft_mi355x.h — C ABI of the MI355X-native NUFFT engine (libnufft_mi355x.so).
 *
 * Drop-in boundary for the GPU hot path of jipolanco/NonuniformFFTs.jl
 * (PlanNUFFT -> set_points! -> exec_type1! / exec_type2!).  The reference has no FFI: its
 * backend "plugin API" is multiple dispatch on a KernelAbstractions backend plus the hooks of
 * ext/NonuniformFFTsAMDGPUExt.jl.  Each entry point below names the reference generic
 * function (file:line relative to the reference checkout) that a Julia `ccall` shim would
 * route to it; INTEGRATION.md shows that shim.
 *
 * Conventions (identical to the reference):
 *   - all data pointers are DEVICE pointers owned by the caller (ROCArray / torch tensor);
 *   - arrays are column-major ("dimension 1 fastest"), coordinates are structure-of-arrays;
 *   - no entry point synchronises the device: all work is enqueued on the `stream` argument
 *     (a hipStream_t passed as void*; NULL = the default stream);
 *   - errors are return codes, never exceptions; nothing is launched when a check fails;
 *   - the plan owns its scratch (oversampled grids, bin-sort buffers, rocFFT plans);
 *   - one plan must not be used concurrently from several threads (same as the reference).
 *
 * No torch / Julia / C++ types appear in any signature.
 */
#ifndef NUFFT_MI355X_H
#define NUFFT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NUFFT_MI355X_VERSION 104 /* 0.1.4: nufft_params.struct_size (was reserved[0]) and .options (appended); the library no longer reads NUFFT_*
                                    environment variables; nufft_plan_options, nufft_workspace_breakdown added;
                                    0.1.3: nufft_params grew (kernel_param_dim, N_over; nufft_info: sort_column: what a binding that already holds the reference's
                                    per-dimension kernel data forwards verbatim), NUFFT_METHOD_GLOBAL_MEMORY accepted, nufft_copy_grid
                                    takes a non-const plan since 102;
                                    0.1.2: nufft_spread_deferred added, nufft_info.reserved_info became ring_halo (same layout) since 101;
                                    0.1.1: nufft_info grew (patch_dims .. ring_segments) since 100 — a caller built against an older
                                    header must compare nufft_sizeof_info() / nufft_version() with its own before nufft_plan_info() */

/* ---- return codes ------------------------------------------------------------------- */
enum {
    NUFFT_OK = 0,
    NUFFT_ERR_INVALID_ARG   = 1, /* Julia ArgumentError (bad enum/eltype/null pointer)            */
    NUFFT_ERR_SIZE_TOO_SMALL = 2, /* ArgumentError "data size is too small" src/plan.jl:545-556    */
    NUFFT_ERR_DIM_MISMATCH  = 3, /* DimensionMismatch src/NonuniformFFTs.jl:92-114                */
    NUFFT_ERR_LDS_TOO_SMALL = 4, /* ArgumentError of block_dims_gpu_shmem src/gpu_common.jl:55-65 */
    NUFFT_ERR_UNSUPPORTED   = 5, /* feature outside the built menu (kernel, M, gpu_method, ...)   */
    NUFFT_ERR_NO_POINTS     = 6, /* exec_* before set_points                                      */
    NUFFT_ERR_ALLOC         = 7, /* hipMalloc failed                                              */
    NUFFT_ERR_HIP           = 8, /* any other HIP runtime error                                   */
    NUFFT_ERR_ROCFFT        = 9, /* rocFFT plan creation / execution failed                       */
    NUFFT_ERR_NO_DEVICE     = 10 /* device entry point called on a host-only (device = -1) plan   */
};

/* ---- enums --------------------------------------------------------------------------- */
enum { NUFFT_F32 = 0, NUFFT_F64 = 1 };                         /* real(Z) of the plan            */
enum {                                                          /* the four kernels of src/Kernels */
    NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL = 0,  /* default_kernel(::ROCBackend), kaiser_bessel_backwards.jl */
    NUFFT_KERNEL_KAISER_BESSEL           = 1,  /* KaiserBesselKernel, kaiser_bessel.jl                     */
    NUFFT_KERNEL_GAUSSIAN                = 2,  /* GaussianKernel, gaussian.jl                              */
    NUFFT_KERNEL_BSPLINE                 = 3   /* BSplineKernel, bspline.jl                                */
};
enum { NUFFT_EVAL_DIRECT = 0, NUFFT_EVAL_FAST_APPROXIMATION = 1 }; /* Kernels.EvaluationMode     */
enum {                                                          /* gpu_method (src/blocking/gpu.jl:26): scheduling-only in the   */
    NUFFT_METHOD_SHARED_MEMORY = 0,                             /* reference (same sums, src/spreading/gpu.jl:168-214); every    */
    NUFFT_METHOD_GLOBAL_MEMORY = 1                              /* plan here runs the LDS engines, both values are accepted      */
};
enum {
    NUFFT_POINT_TRANSFORM_IDENTITY = 0, /* point_transform = identity (src/plan.jl:476)                    */
    NUFFT_POINT_TRANSFORM_NFFT     = 1  /* _transform_point_convention, src/abstractNFFTs.jl:147-155:
                                           x in [-1/2, 1/2), opposite sign of the exponent (plan_nfft)     */
};

/* Stage identifiers (nufft_get_stage_times), in the order of the reference's TimerOutputs
 * labels: src/blocking/gpu.jl:93-139, src/NonuniformFFTs.jl:157-186,246-283. */
enum {
    NUFFT_STAGE_SET_POINTS = 0, /* "Set points": bin-sort of the points                           */
    NUFFT_STAGE_T1_ZERO    = 1, /* "(0) Fill with zeros"                                          */
    NUFFT_STAGE_T1_SPREAD  = 2, /* "(1) Spreading"                                                */
    NUFFT_STAGE_T1_FFT     = 3, /* "(2) Forward FFT"                                              */
    NUFFT_STAGE_T1_DECONV  = 4, /* "(3) Deconvolution"                                            */
    NUFFT_STAGE_T2_DECONV  = 5, /* "(0)+(1) zero-pad + deconvolution" (fused)                     */
    NUFFT_STAGE_T2_FFT     = 6, /* "(2) Backward FFT"                                             */
    NUFFT_STAGE_T2_INTERP  = 7, /* "(3) Interpolation"                                            */
    NUFFT_NUM_STAGES       = 8
};

typedef struct nufft_plan nufft_plan; /* opaque */

/* Plan parameters: the keyword arguments of PlanNUFFT (src/plan.jl:467-482,568-599) that can
 * cross a C ABI, plus the MI355X tile knobs that replace `block_size` / `gpu_batch_size`.
 * Zero-initialise, then set what you need (0 means "reference default" for every field). */
typedef struct nufft_params {
    int32_t dtype;           /* NUFFT_F32 | NUFFT_F64  = real(Z)                                  */
    int32_t is_complex;      /* Z <: Complex (non-uniform values are complex)                     */
    int32_t ndim;            /* 1..3                                                               */
    int64_t N[3];            /* uniform grid size Ns (dimension 1 first)                           */
    int32_t half_support;    /* m = HalfSupport(M); 0 -> 4 (src/plan.jl:583)                       */
    double  sigma;           /* oversampling factor; 0 -> 2.0 (src/plan.jl:573)                    */
    int32_t kernel;          /* NUFFT_KERNEL_*                                                     */
    int32_t evalmode;        /* NUFFT_EVAL_*; the ROC default is Direct (ext/..AMDGPUExt.jl:56)    */
    int32_t ntransforms;     /* ntransforms = Val(C); 0 -> 1                                       */
    int32_t fftshift;        /* fftshift = true/false (src/plan.jl:472)                            */
    int32_t point_transform; /* NUFFT_POINT_TRANSFORM_*                                            */
    int32_t gpu_method;      /* NUFFT_METHOD_*                                                     */
    int32_t device;          /* HIP device ordinal; -1 = host-only plan (parameter math only)      */
    /* --- MI355X tuning knobs (0 = automatic) --- */
    int32_t tile_dims[3];    /* spreading tile (cells; replaces block_dims_gpu_shmem's cube)       */
    int32_t lds_budget_bytes;/* LDS bytes the tile search may use (<= 163840 on gfx950)            */
    int32_t spread_threads;  /* workgroup size of the spreading kernel (multiple of 64)            */
    int32_t interp_threads;  /* workgroup size of the interpolation kernel                         */
    int32_t interp_tile_dims[3]; /* interpolation tile interior (cells)                            */
    int32_t bin_log2;        /* log2 of the bin edge of the point sort (default 2: 4^D cells)      */
    int32_t spread_method;   /* NUFFT_SPREAD_* (0 = automatic: MFMA patches where they apply)       */
    double  kernel_param;    /* KaiserBesselKernel(β) / BackwardsKaiserBesselKernel(β) / GaussianKernel(ℓ):
                                explicit shape parameter; 0 -> the optimal one for (M, σ)              */
    int32_t struct_size;     /* sizeof(nufft_params) of the CALLER's header (ABI >= 104; the slot was reserved[0]): the library reads only
                                that many bytes and treats the rest as zero.  0 = the layout of ABI <= 102, which ends here: every field
                                below is then ignored — a caller that fills them must set struct_size                              */
    int32_t reserved;
    double  kernel_param_dim[3]; /* per-dimension shape parameter, as the reference's kernel data holds it (the field β of
                                    BackwardsKaiserBesselKernelData / KaiserBesselKernelData, kaiser_bessel_backwards.jl:84,
                                    kaiser_bessel.jl:112; σ / Δx of GaussianKernelData, gaussian.jl:67,76-78): an entry > 0 overrides
                                    kernel_param and the optimal value for that dimension, so a binding forwards p.kernels[d] verbatim */
    int64_t N_over[3];       /* oversampled grid size Ñ_d (gridsize(p.kernels[d]), src/Kernels/Kernels.jl:87): an entry > 0 replaces the
                                size rule of src/plan.jl:485-498 for that dimension (it must be even in dimension 1 of a real plan
                                and >= N_d); sigma is then only reported */
    const char* options;     /* development switches of this plan, "NUFFT_NAME=value;NUFFT_OTHER=value" (DESIGN.md section 4.3 lists them;
                                A/B experiments and tests of rarely taken paths — the defaults are the measured optimum), or NULL.
                                The library reads no environment variable: what changes a plan is in this struct.  Copied at
                                plan creation; nufft_plan_options() returns the canonical form the plan holds                  */
} nufft_params;

/* What show(::PlanNUFFT) prints (src/plan.jl:362-392) plus sizes a caller needs. */
/* Spreading engines (nufft_info.spread_method; nufft_params.spread_method selects, 0 = automatic):
 *   LDS tiles    — output-driven LDS tile with native ds_add_f64 (every D, M, kernel, grid size)
 *   MFMA patches — register-resident patches accumulated by the matrix pipe (3-D grids of 4-cell bins):
 *                  v_mfma_f64_4x4x4_4b with Float64 accumulators, or — ComplexF32 plans whose dimension 3 is a
 *                  multiple of 8 cells — v_mfma_f32_16x16x4 with Float32 accumulators (nufft_info.patch_f32acc)
 *   marching ring — a workgroup owns a column of the grid and marches along dimension 3 with a ring of 2M - 1 + 4 planes in
 *                  LDS (ds_add_f64 as for the tiles, 1.4 - 1.5 point visits per point instead of 2.1, finished planes leave with
 *                  coalesced stores while the ring moves on): 3-D grids of 4-cell bins; the automatic choice for real data, M <= 4
 *   marching ring, dense — the same window for dense point sets (mean load of the 4^3-cell bins above a threshold per M, decided per point
 *                  set by nufft_set_points): the points of a bin are accumulated in registers by v_mfma_f64_16x16x4 and flushed to the
 *                  window once per bin.  Reported by nufft_spread_engine_used only (never a plan parameter). */
enum { NUFFT_SPREAD_AUTO = 0, NUFFT_SPREAD_LDS_TILES = 1, NUFFT_SPREAD_MFMA_PATCHES = 2, NUFFT_SPREAD_MARCHING_RING = 3,
       NUFFT_SPREAD_MARCHING_RING_DENSE = 4 };

typedef struct nufft_info {
    int32_t dtype, is_complex, ndim, half_support, ntransforms, evalmode, fftshift, device;
    int64_t N[3];            /* Ns                                                                 */
    int64_t N_over[3];       /* oversampled grid dims  (src/plan.jl:485-498)                       */
    int64_t N_out[3];        /* size(p): dims of the uniform arrays (src/plan.jl:426)              */
    double  sigma;           /* actual sigma = max(N_over / N) (src/plan.jl:500)                   */
    double  beta[3];         /* kernel shape parameter per dimension (β; ℓ/Δx for the Gaussian; 0: B-spline) */
    int32_t bin_dims[3];     /* cells per sort bin                                                 */
    int32_t nbins[3];        /* bins per dimension                                                 */
    int32_t spread_tile[3];  /* spreading tile: interior cells held in LDS (no halo)               */
    int32_t spread_ntiles[3];
    int32_t interp_tile[3];  /* interpolation tile interior; LDS holds interior + 2M - 1           */
    int32_t interp_ntiles[3];
    int32_t spread_threads, interp_threads;
    int64_t lds_bytes_spread, lds_bytes_interp;
    int64_t workspace_bytes; /* device bytes owned by the plan right now                           */
    int64_t num_points;      /* Np of the last set_points                                          */
    int32_t npoly;           /* M + 4 polynomial coefficients per sub-interval                     */
    int32_t window_scale_log2[3]; /* device windows and phi_hat are scaled by 2^k_d (exact) */
    int32_t kernel;          /* NUFFT_KERNEL_*                                                     */
    int32_t spread_max_items, interp_max_items; /* capacity of the per-tile work-item tables (runs of sorted points) */
    int32_t spread_method;   /* NUFFT_SPREAD_LDS_TILES, _MFMA_PATCHES or _MARCHING_RING (what nufft_spread launches)  */
    int32_t patch_dims[2];   /* MFMA patches: cube columns (of 4 x 4 cells) a wave owns along dimensions 1, 2; 0 otherwise */
    int32_t patch_f32acc;    /* MFMA patches: 1 = ComplexF32 on v_mfma_f32_16x16x4 with Float32 accumulators (the reference's
                                accumulation type, src/spreading/gpu.jl:271-283), 0 = v_mfma_f64_4x4x4 with Float64 ones  */
    int32_t patch_planar;    /* MFMA patches: real plans with ntransforms = 2 / 3 spread that many components together (shared window
                                evaluation and operands — the reference's TODO at src/spreading/gpu.jl:293); 0 = one at a time    */
    int32_t ring_column[2];  /* marching ring: cells of a workgroup's column along dimensions 1, 2; 0 otherwise         */
    int32_t ring_segments;   /* marching ring: segments a column is cut into along dimension 3 for uniform point sets  */
    int32_t ring_halo;       /* marching ring: 1 = halo variant (every point spread once by its own column; the stencil reach
                                travels through a side buffer that the first FFT pass adds), 0 = clipped columns          */
    int32_t sort_column[2];  /* plans whose spreading window (halo variant) and interpolation ring own the same columns: bins of a column
                                along dimensions 1, 2 — set_points then groups the points by (column, layer of bins) only, unless a
                                ring hands the point set to the tile kernels (nufft_sort_columns_used); 0: always the fine bins   */
} nufft_info;

/* ---- plan lifetime -------------------------------------------------------------------- */

/* PlanNUFFT(Z, Ns; m, σ, kernel, ntransforms, backend = ROCBackend(), kernel_evalmode, fftshift,
 * gpu_method = :shared_memory) -> _PlanNUFFT, src/plan.jl:467-541; BlockDataGPU src/blocking/gpu.jl:41-69;
 * init_plan_data (grids + FFT plans) src/plan.jl:37-60. */
int nufft_plan_create_ex(nufft_plan** out, const nufft_params* params);

/* Flat-argument form of the same constructor (what SURVEY.md §8(b) lists). */
int nufft_plan_create(nufft_plan** out, int dtype, int is_complex, int ndim, const int64_t* N,
                      int half_support, double sigma, int kernel, int evalmode, int ntransforms,
                      int fftshift, int point_transform, int device);

/* Julia finalizer of the plan. */
int nufft_plan_destroy(nufft_plan* plan);

/* size(p), ndims(p), ntransforms(p), show(p): src/plan.jl:360-435. */
int nufft_plan_info(const nufft_plan* plan, nufft_info* out);

/* Plan-time host arrays, for inspection and tests:
 *   phi_hat : Kernels.fourier_coefficients(g), src/Kernels/Kernels.jl:84 (length N_out[dim])
 *   poly    : piecewise-polynomial coefficients cs[k][j], k < npoly, j < 2M
 *             (src/Kernels/piecewise_polynomial.jl:50-74), row-major [npoly][2M]
 *   index_map : non_oversampled_indices!, src/NonuniformFFTs.jl:318-348, 0-based */
int nufft_plan_get_phi_hat(const nufft_plan* plan, int dim, double* out, int64_t capacity);
int nufft_plan_get_poly_coefs(const nufft_plan* plan, int dim, double* out, int64_t capacity);
int nufft_plan_get_index_map(const nufft_plan* plan, int dim, int64_t* out, int64_t capacity);

/* ---- the hot path --------------------------------------------------------------------- */

/* set_points!(p, (xs, ys, zs)) -> set_points_impl!(::GPU, ...), src/set_points.jl:33-52,
 * src/blocking/gpu.jl:73-142.  coords[d] = device vector of Np reals of the plan's precision.
 * The coordinates are folded to [0, 2π) and bin-sorted by LDS tile into plan-owned storage; the
 * caller's arrays are only read (and, unlike the reference, need not stay alive afterwards). */
int nufft_set_points(nufft_plan* plan, int64_t num_points, const void* const* coords, void* stream);

/* exec_type1!(ûs_k, p, vp), src/NonuniformFFTs.jl:148-195.
 * values_in[c]: device vector Z[Np]; uhat_out[c]: device array complex(T)[N_out...]. */
int nufft_exec_type1(nufft_plan* plan, void* const* uhat_out, const void* const* values_in, void* stream);

/* exec_type2!(vp, p, ûs_k), src/NonuniformFFTs.jl:237-291. */
int nufft_exec_type2(nufft_plan* plan, void* const* values_out, const void* const* uhat_in, void* stream);

/* exec_type1!(ûs, p, vp; callbacks) / exec_type2!(vp, p, ûs; callbacks), src/NonuniformFFTs.jl:148-195,237-291,
 * for the two documented uses of NUFFTCallbacks (src/plan.jl:105-143, test/callbacks.jl:17-25) — arbitrary
 * closures cannot cross a C ABI:
 *   point_weights  T[Np]  (device, real, in the caller's point order): callbacks.nonuniform = (v, n) -> v * w[n],
 *                  applied to the values read by type 1 / written by type 2 (src/spreading/gpu.jl:289,
 *                  src/interpolation/gpu.jl:254);
 *   mode_factors   T[N_out...] (device, real, same layout as one uniform array, shared by all components):
 *                  callbacks.uniform = (w, idx) -> w * f[idx], applied to the modes written by type 1 / read by
 *                  type 2 (src/NonuniformFFTs.jl:395-399,461-464).
 * Either pointer may be NULL (identity).  Both are fused into existing kernels: no extra pass over the data. */
typedef struct nufft_callbacks {
    const void* point_weights;
    const void* mode_factors;
} nufft_callbacks;
int nufft_exec_type1_cb(nufft_plan* plan, void* const* uhat_out, const void* const* values_in,
                        const nufft_callbacks* callbacks, void* stream);
int nufft_exec_type2_cb(nufft_plan* plan, void* const* values_out, const void* const* uhat_in,
                        const nufft_callbacks* callbacks, void* stream);

/* ---- stage-level entry points (the backend-dispatched generic functions) -- */

/* The callback menu (above) for the stage-level entry points below: `callbacks.nonuniform` is read by nufft_spread[_deferred] and
 * nufft_interpolate, `callbacks.uniform` by nufft_deconvolve_truncate and nufft_deconvolve_pad — the arguments the reference passes
 * to spread_from_points! / interpolate! / copy_deconvolve_to_*! (src/NonuniformFFTs.jl:170,183,270,280).  In force until the next call;
 * NULL (or two NULL pointers) = none.  The pointers are read while a stage enqueues its kernels, not later. */
int nufft_set_callbacks(nufft_plan* plan, const nufft_callbacks* callbacks);

/* fill_with_zeros_kernel!(us), src/NonuniformFFTs.jl:116-122,161-167. */
int nufft_fill_zeros(nufft_plan* plan, void* stream);
/* spread_from_points!(::GPU, ...), src/spreading/gpu.jl:134-214 (adds onto the plan's grids). */
int nufft_spread(nufft_plan* plan, const void* const* values_in, void* stream);
/* The same stage as exec_type1 enqueues it: on plans whose spreading engine is the marching ring's halo variant (nufft_info.ring_halo)
 * the stencil reach beyond each workgroup's column is left in a side buffer.  nufft_fft_forward consumes it on the fly (its
 * dimension-1 pass adds the buffer to the lines it loads — the spectrum is that of the complete grid, `us` itself stays without
 * the reach); nufft_complete_grid, nufft_copy_grid(which = 0) and nufft_interpolate add it to `us` (once), before or after that FFT.
 * The pending state ends with the next nufft_set_points, nufft_spread[_deferred], nufft_fill_zeros or nufft_fft_backward.
 * nufft_spread completes the grid itself with one more pass.  Identical to nufft_spread on every other plan.
 * The state is host-side: a deferred spread and its consumer must be enqueued on the same stream, and captured in the same hipGraph
 * (tests/test_gpu_graph.py), since a replayed spread does not set it again.
 * (No reference counterpart: src/NonuniformFFTs.jl:169-177 calls spread_from_points! and _type1_fft! back to back.) */
int nufft_spread_deferred(nufft_plan* plan, const void* const* values_in, void* stream);
/* _type1_fft!, src/NonuniformFFTs.jl:197-211. */
int nufft_fft_forward(nufft_plan* plan, void* stream);
/* copy_deconvolve_to_non_oversampled!(::GPU, ...), src/NonuniformFFTs.jl:387-414. */
int nufft_deconvolve_truncate(nufft_plan* plan, void* const* uhat_out, void* stream);
/* fill_with_zeros + copy_deconvolve_to_oversampled!(::GPU, ...), src/NonuniformFFTs.jl:260-272,453-480. */
int nufft_deconvolve_pad(nufft_plan* plan, const void* const* uhat_in, void* stream);
/* _type2_fft! / _fft_c2r!, src/NonuniformFFTs.jl:293-314. */
int nufft_fft_backward(nufft_plan* plan, void* stream);
/* interpolate!(::GPU, ...), src/interpolation/gpu.jl:40-118. */
int nufft_interpolate(nufft_plan* plan, void* const* values_out, void* stream);

/* Adds the side buffer of a deferred spread to `us` if that has not happened yet (no-op otherwise): after it, `us` holds the full
 * spread field as after the reference's spread_from_points! (src/NonuniformFFTs.jl:169-172), also behind nufft_exec_type1. */
int nufft_complete_grid(nufft_plan* plan, void* stream);

/* Device pointer of plan-owned oversampled arrays (p.data.us / p.data.ûs, src/plan.jl:3-29):
 * which = 0 -> us[c] (real T[N_over] or complex), which = 1 -> ûs[c] (real plans only).
 * which = 0 is refused (NUFFT_ERR_INVALID_ARG) while a deferred spread is pending: call nufft_complete_grid first. */
int nufft_grid_ptr(const nufft_plan* plan, int which, int component, void** out_ptr, int64_t* out_bytes);

/* Device-to-device copy of one plan-owned oversampled array (same `which` as nufft_grid_ptr) into
 * a caller buffer of at least `capacity_bytes`; enqueued on `stream`. */
int nufft_copy_grid(nufft_plan* plan, int which, int component, void* dst, int64_t capacity_bytes, void* stream);

/* Sorted-point inspection: copies the bin-sort permutation (sorted position -> original index,
 * 0-based; BlockDataGPU.pointperm, src/blocking/gpu.jl:15) and the per-tile offsets
 * (cumulative_npoints_per_block, :13) to HOST buffers.  Synchronises `stream`. */
int nufft_get_sort_result(nufft_plan* plan, int32_t* perm_host, int64_t perm_capacity,
                          uint32_t* tile_offsets_host, int64_t offsets_capacity, void* stream);

/* Which sort the last nufft_set_points used: 1 = by column layers (nufft_info.sort_column; the offsets returned by
 * nufft_get_sort_result then hold a column layer's points in its first bin and nothing in its other bins), 0 = by fine bins
 * (histogram with global atomics), 2 = by fine bins in two levels (slabs of bin rows with LDS histograms, then every slab in LDS:
 * the same array and offsets as 0 up to the order inside a bin; 3-D plans without sort_column, point sets whose fullest slab fits
 * a workgroup's LDS).  Reads device flags back (synchronises `stream`).  Inspection only. */
int nufft_sort_columns_used(nufft_plan* plan, int* used_out, void* stream);

/* ---- timing (TimerOutputs analogue, src/plan.jl:397-417) -------------------------------- */

/* enable != 0: bracket every stage with hipEvents on the caller's stream. */
int nufft_set_timing(nufft_plan* plan, int enable);
/* Milliseconds of the most recent run of every stage (NUFFT_NUM_STAGES floats; -1 = never run).
 * Synchronises on the recorded events. */
int nufft_get_stage_times(nufft_plan* plan, float* ms_out);

/* Which engine served the point set of the last nufft_set_points: NUFFT_SPREAD_LDS_TILES, _MFMA_PATCHES or _MARCHING_RING.
 * On plans whose nufft_info.spread_method is MFMA patches or the marching ring, set_points decides per point set on the
 * device (a point set that concentrates in a few patches / columns goes to the LDS tiles, whose heavy tiles are shared by
 * several workgroups);
 * this reads the decision back (4 bytes, synchronises `stream`).  Inspection only: nothing on the hot path needs it. */
int nufft_spread_engine_used(nufft_plan* plan, int* engine_out, void* stream);
/* The same for the interpolation stage of nufft_exec_type2: NUFFT_INTERP_LDS_TILES (padded boxes, heavy tiles shared by
 * several workgroups: interp_tile_kernel) or NUFFT_INTERP_MARCHING_RING (3-D plans with the default window evaluation,
 * point sets whose heaviest ring task and total work stay within the ring's measured advantage over the tile kernel:
 * interp_march_kernel).  Replaces nothing in the reference — its
 * interpolate! (src/interpolation/gpu.jl:3-89) has one shared-memory kernel; inspection only. */
enum { NUFFT_INTERP_LDS_TILES = 1, NUFFT_INTERP_MARCHING_RING = 2 };
int nufft_interp_engine_used(nufft_plan* plan, int* engine_out, void* stream);

/* ---- misc ----------------------------------------------------------------------------- */
/* sizeof(nufft_params) / sizeof(nufft_info) of the library build: a binding that mirrors the structs by hand
 * (ctypes, Julia) compares them with its own layout before the first call. */
/* Plan-owned device memory right now, buffer by buffer: "name=bytes;name=bytes;..." (NUL-terminated) into `out`; the values sum to
 * nufft_info.workspace_bytes.  Names: us, uhat, tmp2 (oversampled grids / spectra / intermediate of the pruned FFT passes), sorted
 * (bin-sorted point records), sort_scratch, sort_slice_table, bin_counts, bin_offsets, vsorted (values in sorted order: MFMA-patch
 * plans), ring_side_buffer (halo variant of the spreading window), rocfft_work, tables (everything small).  The reference's plan holds
 * us + ûs (src/plan.jl:37-60) and blockidx + pointperm + offsets (src/blocking/gpu.jl:41-69) and aliases the caller's points. */
int nufft_workspace_breakdown(const nufft_plan* plan, char* out, int64_t capacity);
/* The development switches the plan was created with (nufft_params.options), canonical form "NAME=value;..." sorted by name; "" if none.
 * The pointer stays valid until the plan is destroyed. */
const char* nufft_plan_options(const nufft_plan* plan);
int64_t nufft_sizeof_params(void);
int64_t nufft_sizeof_info(void);
const char* nufft_strerror(int code);
/* Last error message of the calling thread (more detail than nufft_strerror). */
const char* nufft_last_error_message(void);
int nufft_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NUFFT_MI355X_H */
