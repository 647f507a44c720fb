// Host-callable launchers of the HIP kernels (one translation unit per kernel family).
#pragma once

#include "options.h"
#include <hip/hip_runtime.h>
#include <cstdint>

#include "column_tasks.h"
#include "device_common.h"
#include "nufft_mi355x.h"

namespace nufft {

// ---- bin sort (binsort.hip) ------------------------------------------------------------------
struct SortArgs {
    CoarseSort cs;
    int dtype, D;
    int point_transform;   // NUFFT_POINT_TRANSFORM_*
    int64_t np;
    const void* coords[3];
    Geom g;
    uint32_t* counts;      // [nbins + 1]
    bool counts_clean;     // the histogram is all zero on entry (left so by the previous call's scatter pass)
    uint32_t* offsets;     // [nbins + 1]
    void* binrank;         // uint2[np]
    void* sorted;          // PointRec<T, D>[np]
    void* scan_tmp;
    size_t scan_tmp_bytes;
};
size_t binsort_scan_tmp_bytes(int nbins);
size_t point_record_bytes(int dtype, int D);
hipError_t launch_binsort(const SortArgs& s, hipStream_t stream);
// the two halves of a set_points on a plan with CoarseSort::enabled: column-layer histogram + scan (offsets valid for the task kernels
// that decide flag_a / flag_b), then — after those kernels — the scatter of whichever sort the flags select
hipError_t prepare_binsort_coarse(int dtype, int nkeys);
// two-level slab sort (CoarseSort::mode = 2, column_tasks.h): records a level-2 workgroup holds in lds_bytes of LDS, and back
int slab_sort_capacity(int dtype, int lds_bytes);
int slab_sort_lds_bytes(int dtype, int cap);
hipError_t prepare_binsort_slab(int dtype, int lds_bytes);
hipError_t launch_binsort_coarse_count(const SortArgs& s, hipStream_t stream);
hipError_t launch_binsort_coarse_finish(const SortArgs& s, hipStream_t stream);
hipError_t launch_sort_feedback(const uint32_t* ra, const uint32_t* rb, uint32_t* feedback, uint32_t seq, hipStream_t stream);
// zero fill by a kernel (hipGraph-safe, see binsort.hip); dst 16-byte aligned, bytes a multiple of 4
hipError_t launch_zero_fill(void* dst, size_t bytes, hipStream_t stream);
hipError_t launch_extract_perm(int dtype, int D, const void* sorted, int64_t np, int32_t* perm_dev, hipStream_t stream);

// ---- load balance (balance.hip) ----------------------------------------------------------------------
struct BalanceArgs {
    Geom g;
    int D, M;
    bool enabled;              // false: one slice per tile
    uint32_t extra_sp, extra_ip;   // budgets of extra slices (= extra workgroups of the launch grids)
    uint32_t smax;             // most slices of one tile (< 65536)
    const uint32_t* offsets;   // bin offsets of the sort
    // arrays over the tiles of both tilings, spreading tiles first
    uint32_t* work;            // [nsp + nip]
    uint32_t* nslices;         // [nsp + nip + 1]
    uint32_t* desc_off;        // [nsp + nip + 1]
    uint2* desc;               // [nsp + extra_sp + nip + extra_ip]; interpolation slots start at nsp + extra_sp
    uint32_t* slots_in_use;    // [2]
    void* scan_tmp;
    size_t scan_tmp_bytes;
    const uint32_t* skip_a;    // both nonzero: the point set is column-layer sorted and served by the two rings — no tile tables (no slots)
    const uint32_t* skip_b;
    const uint32_t* sp_served; // nonzero: a ring spreads this point set (no spreading slots); null: the task kernels that follow decide
};
size_t balance_scan_tmp_bytes(int ntiles_both);
size_t balance_work_words(int ntiles_both);       // 32-bit words of BalanceArgs::work (tile counters + partial sums)
hipError_t launch_balance(const BalanceArgs& b, hipStream_t stream);
hipError_t launch_zero_split_tiles(int dtype, const Geom& g, int D, int is_complex, int C, const uint32_t* nslices,
                                   void* grid, int64_t grid_stride_reals, hipStream_t stream);

// ---- spreading / interpolation (spread_*.hip, interp_*.hip) -------------------------------------

struct TileKernelArgs {
    int dtype, is_complex, D, M, evalmode, C;
    int kernel;                // NUFFT_KERNEL_*
    Geom g;
    const void* sorted;        // PointRec<T, D>[np]
    const uint32_t* offsets;   // [nbins + 1]
    const void* coefs;         // T[D][npoly][2M]
    double beta[3];            // first window parameter per dimension (β; Δx for the Gaussian)
    double beta_over_pi[3];    // second one: BKB (β/π) 2^k, KB 2^k, Gaussian τ (see WindowEval, plan.cpp)
    void* grid;                // C grids, contiguous, Z[Nover...]
    int64_t grid_stride;       // elements of Z between components
    const void* const* values_in;   // spread: C device vectors Z[np]
    void* const* values_out;        // interp: C device vectors Z[np]
    double prefactor;          // interp: prod(dx_d)
    const void* weights;       // optional T[np]: real weight per point (nonuniform callback menu), or null
    int threads;
    int lds_bytes;
    int ntiles;                // workgroups per component (tiles + budget of extra slices)
    const void* desc;          // uint2[ntiles]: slot -> (tile, slice << 16 | slices), see balance.hip
    const uint32_t* desc_total;    // number of slots in use
    int xcd_chunk;
    int march;                 // interpolation: also launch the z-marching kernel (march_kernels.h); decided per point set on the device
    ColumnTasks march_ct;      // ... its columns and tasks (march_column_tasks),
    const uint32_t* march_flag;   // the device flag of set_points (1: the ring serves this point set)
    const uint2* march_tasks;     // and the task table
    int fixed_tile;            // the tile (g.ip / g.sp) equals the compile-time one (kernel variant with constant strides)
    int cubes;                 // spreading with the compile-time tile: accumulate cube by cube with the FP64 matrix instruction
    void* halo;                // marching ring, halo variant: side buffer of the stencil reach (C components, SMarchPlan::halo_reals each)
    int interp_parts;          // 2: complex data interpolated part by part by the real ring kernels (march_kernels.h, MarchGeom::parts); else 1
    int coarse;                // plan of the column-layer sort (CoarseSort): point sets with both flags nonzero are sorted that way
    const uint32_t* coarse_a;
    const uint32_t* coarse_b;
};
// Compile-time interpolation tile of an instantiation: n[0..2] cells (n[0] == 0: none), n[3] = LDS row
// stride in reals; see fixed_interp_tile().
void interp_fixed_dims(int dtype, int is_complex, int D, int M, int* n);
// the same for the spreading tile (fixed_spread_tile())
void spread_fixed_dims(int dtype, int is_complex, int D, int M, int* n);
bool spread_cubes_available(int dtype, int is_complex, int D, int M);
// z-marching interpolation (march_kernels.h): available for this plan?  (3-D, 4-cell bins, default window evaluation)
bool interp_march_available(int dtype, int is_complex, int D, int M, bool poly, const Geom& g, bool other);
hipError_t prepare_interp_march(int dtype, int is_complex, int M, bool poly);
// ... its variant for column-layer sorted point sets (same columns and tasks)
bool interp_march_staged_available(int dtype, int is_complex, int M, bool poly, int n1, int n2);
hipError_t prepare_interp_march_staged(int dtype, int is_complex, int M, bool poly);
// columns (the kernel's compile-time column) and evenly cut tasks of the ring for this grid
ColumnTasks march_column_tasks(int dtype, int is_complex, int M, bool poly, const Geom& g, int n1 = 0, int n2 = 0);
hipError_t launch_spread(const TileKernelArgs& a, hipStream_t stream);
hipError_t launch_interp(const TileKernelArgs& a, hipStream_t stream);
// Sets the dynamic-LDS attribute of every instantiation that may be launched for this configuration.
hipError_t prepare_spread(int dtype, int is_complex, int D, int M, int lds_bytes, bool other);
hipError_t prepare_interp(int dtype, int is_complex, int D, int M, int lds_bytes, bool other);
bool needs_other_eval(int kernel, int evalmode);

// ---- spreading on MFMA patches (patch_kernels.h, patch_*.hip) ------------------------------------------------------
struct PatchPlan {
    bool eligible;
    int npx, npy, nseg, segl, ntasks;   // patch columns, segments of cube layers along dimension 3, wave tasks
    int lds_bytes;                      // dynamic LDS per workgroup
    int pby;                            // rows of cube columns per patch
    int occ;                            // waves per SIMD the kernel is compiled for
    int f32acc;                         // ComplexF32 on the FP32 matrix pipe with Float32 accumulators (patch32_kernels.h)
    int planar;                         // real plans with ntransforms = 2 / 3: that many components spread together (0: one at a time)
};
// allow_f32acc: ComplexF32 plans may take the FP32-matrix-pipe kernel where the grid allows it (octets along dimension 3)
// planar_nc: ntransforms of a real plan whose components may be spread together (2 or 3; 0 = one component per launch)
PatchPlan patch_plan(int dtype, int is_complex, int D, int M, const Geom& g, bool other, bool allow_f32acc = true, int planar_nc = 0);
// set_points: cuts the patch columns into tasks of about equal point count and decides on the device which engine serves
// this point set (balance.hip); choice = uint32[8], zeroed once; colsum[columns], first[columns + 1], tasktab[pp.ntasks]
hipError_t launch_march_tasks(const Geom& g, const ColumnTasks& ct, const uint32_t* offsets, int64_t np, int cus, double advantage, double rho_eff_max,
                              uint32_t* choice, uint32_t* colsum, uint32_t* first, uint2* tasktab, hipStream_t stream);
// advantage: how much faster than the LDS tiles the patches are on uniform points (<= 0: always the patches)
hipError_t launch_patch_tasks(const Geom& g, const PatchPlan& pp, int clo, int chi, const uint32_t* offsets, int64_t np,
                              int wave_slots, double advantage, uint32_t* choice, uint32_t* slots_in_use, uint32_t* colsum, uint32_t* first,
                              uint2* tasktab, hipStream_t stream);
bool patch_tasks_supported(const Geom& g);
hipError_t prepare_column_tasks();       // once per plan (device attribute of the task sort kernel)
int patch_task_table_entries(const PatchPlan& pp);       // entries of the task table = tasks the patch kernels are launched with
hipError_t prepare_spread_patch(int dtype, int is_complex, int M, bool other, int planar_nc = 0);
// all C value vectors of a real plan gathered into one interleaved buffer vout[p * C + c] (planar patch kernel)
hipError_t launch_gather_planar(int dtype, int D, const void* sorted, int64_t np, const void* const* vin, int C,
                                const void* weights, void* vout, const uint32_t* enabled, hipStream_t stream);
// vsorted: C value vectors in sorted order (launch_gather_values), vstride_reals reals apart
// enabled: device flag (null: always run); both kernels return at once when *enabled == 0
// tasktab: the task table of launch_patch_tasks
hipError_t launch_spread_patch(const TileKernelArgs& a, const PatchPlan& pp, const void* vsorted, int64_t vstride_reals,
                               const uint32_t* enabled, const uint2* tasktab, hipStream_t stream);
hipError_t launch_gather_values(int dtype, int is_complex, int D, const void* sorted, int64_t np, const void* vin,
                                const void* weights, void* vout, const uint32_t* enabled, hipStream_t stream);

// ---- spreading on the z-marching LDS ring (smarch_kernels.h, smarch_*.hip) -------------------------------------------
// 3-D plans with 4-cell bins and the default window evaluation whose axes are long enough; cus: compute units, C: components
// halo: 0 = output-driven in x and y; 2 = the halo variant (real data, grids the column divides; falls back to 0 where it cannot run)
SMarchPlan smarch_plan(int dtype, int is_complex, int D, int M, const Geom& g, bool other, int cus, int C, int halo, int parts = 1);
hipError_t prepare_spread_march(int dtype, int is_complex, int M, int halo);
// flag: device flag of set_points (1: the ring serves this point set); tasktab: its task table
hipError_t launch_spread_march(const TileKernelArgs& a, const SMarchPlan& sp, const uint32_t* flag, const uint2* tasktab, uint32_t* halo_state, bool dense, hipStream_t stream);
bool spread_dense_available(int dtype, int is_complex, int M, bool poly, const SMarchPlan& sp);
hipError_t prepare_spread_dense(int dtype, int M, bool poly);
// halo variant: grid += side buffer (a.halo); the dimension-1 FFT pass of real plans does the same while it loads its lines (launch_real_lines)
hipError_t launch_smarch_halo_add(const TileKernelArgs& a, const SMarchPlan& sp, const uint32_t* flag, hipStream_t stream);
// set_points: tasks of the ring for this point set and whether it serves it (advantage <= 0: always)
hipError_t launch_smarch_tasks(const Geom& g, const SMarchPlan& sp, const uint32_t* offsets, int64_t np, int cus, double advantage,
                               uint32_t* choice, uint32_t* slots_in_use, uint32_t* colsum, uint32_t* first, uint2* tasktab, hipStream_t stream);

// ---- deconvolution (deconv.hip) ------------------------------------------------------------------
struct DeconvArgs {
    int dtype, D, C;
    int nout[3];               // size(p)
    int nspec[3];              // dims of the oversampled spectrum
    const void* phihat[3];     // T[nout[d]]
    const int32_t* index_map[3];   // out index -> oversampled index
    const int32_t* inv_map[3];     // oversampled index -> out index or -1
    void* spec;                // C oversampled spectra (complex<T>), contiguous
    int64_t spec_stride;       // complex elements between components
    double normfactor;         // prod(2π / Ñ_d) (type 1) or 1 (type 2)
    const void* mode_factors;  // optional T[prod(nout)]: real multiplier per output mode (uniform callback), or null
};
hipError_t launch_deconv_truncate(const DeconvArgs& a, void* const* uhat_out, hipStream_t stream);
hipError_t launch_deconv_pad(const DeconvArgs& a, const void* const* uhat_in, hipStream_t stream);

// ---- pruned strided FFT passes (fft_lines.hip) ----------------------------------------------------
struct FftLinePass {
    const void* in;
    void* out;
    int64_t a_total, a_out;                 // valid contiguous indices; (backward) columns written, rest zero
    int64_t in_stride_j, in_stride_c;       // in elements (complex)
    int64_t out_stride_j, out_stride_c;
    int nc, nk;
    const int32_t* map;                     // kept index -> FFT index
    const void* fa; int ka;                 // T[ka], factor by (a mod ka)
    const void* fk;                         // T[nk], factor by kept index
    const void* twiddle;                    // complex<T>[N]
    double scale;
    const void* mult;                       // optional T[...]: real multiplier indexed like the pruned side, or null
    int row_a = 0, row_valid = 0, row_in = 0, row_out = 0;      // rows of the enumeration and their strides on both sides (0: none)
};
bool fft_lines_supported(int dtype, int64_t n);
bool real_lines_supported(int dtype, int64_t n);
// r2c (forward) / c2r of `nlines` contiguous real lines of length n with a compact spectrum of k1 modes per line
// (row: row stride of the compact spectrum in complex elements, >= k1)
// halo (forward only, or null): the side buffer of the spreading ring's halo variant for these lines (one component; lines of
// planes of ny rows), added to every line while it is loaded — when *flag != 0 (the ring served the point set)
struct RealLineHalo {
    const void* buffer;
    const void* buffer2;       // complex lines, data spread part by part by the real kernel: planar side buffers of the real (buffer) and
                               // imaginary parts (buffer2), layout.nc = 1; null: one interleaved buffer (layout.nc = 2) / real lines
    const uint32_t* flag;
    int ny;
    HaloLayout layout;
};
hipError_t launch_real_lines(int dtype, int64_t n, bool forward, const void* in, void* out, int64_t nlines, int k1, int row,
                             const void* twiddle, hipStream_t stream, const RealLineHalo* halo = nullptr);
hipError_t launch_fft_lines(int dtype, int64_t n, bool forward, const FftLinePass& p, hipStream_t stream);
// grid += side buffer of the spreading window's halo variant, line by line (C components; lines of n1cells cells, planes of ny lines)
hipError_t launch_halo_add_lines(int dtype, void* grid, const void* halo, int64_t grid_comp_reals, int64_t halo_comp_reals, int n1cells, int ny, int nz,
                                 int C, const HaloLayout& h, const uint32_t* flag, hipStream_t stream, bool planar = false);
// c2c of `nlines` contiguous complex lines of length n with a compact spectrum of k1 kept modes (map: kept -> FFT index)
hipError_t launch_cplx_lines(int dtype, int64_t n, bool forward, const void* in, void* out, int64_t nlines, int k1,
                             const int32_t* map, const void* twiddle, hipStream_t stream, const RealLineHalo* halo = nullptr);

}  // namespace nufft
