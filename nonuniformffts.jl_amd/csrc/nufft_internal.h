// Internal declarations shared by the host side (plan.cpp, plan_math.cpp) and the HIP kernel
// translation units of libnufft_mi355x.so.
#pragma once

#include "options.h"
#include <cstdint>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "column_tasks.h"
#include "nufft_mi355x.h"

struct rocfft_plan_t;
struct rocfft_execution_info_t;

namespace nufft {

constexpr int kMaxDim = 3;
constexpr int kMinM = 2;
constexpr int kMaxM = 10;
constexpr int kLdsLimit = 163840;      // gfx950: 160 KiB per workgroup

void set_error(const std::string& msg);

// -------------------------------------------------------------------------------------------
// Host parameter math (plan_math.cpp) — restates src/plan.jl:485-505, Kernels/*.jl
// -------------------------------------------------------------------------------------------
int64_t nextprod235(int64_t n);
int64_t oversampled_size(int64_t N, double sigma, bool real_first_dim);
double bkb_beta(int M, double sigma_d);
double kb_beta(int M, double sigma_d);
double gaussian_ell(int M, double sigma_d);
void kb_poly_coefficients(int M, double beta, std::vector<double>& cs);
void fourier_coefficients_kernel(int kernel, const std::vector<double>& ks, int M, int64_t Nover, double param,
                                 std::vector<double>& phihat);
double bkb_function(double y, double beta);
double bessel_i0(double x);
// cs[k * 2M + j]: coefficient of x^k on sub-interval j (j = 0 is the rightmost one).
void bkb_poly_coefficients(int M, double beta, std::vector<double>& cs);
// wavenumbers of the non-oversampled grid: rfftfreq for (real, dim 0), fftfreq otherwise.
void wavenumbers(int64_t N, bool r2c, std::vector<double>& ks);
void fourier_coefficients(const std::vector<double>& ks, int M, int64_t Nover, double beta,
                          std::vector<double>& phihat);
void non_oversampled_indices(const std::vector<double>& ks, int64_t n_axis, bool fftshift,
                             std::vector<int64_t>& indmap);

// Geometry of the fine bins and of the two LDS tilings (host mirror of device_common.h's Geom).
struct TileShapeHost {
    int n[kMaxDim] = {1, 1, 1};        // interior cells
    int nt[kMaxDim] = {1, 1, 1};       // tiles per dimension
    int row_stride = 0;                // LDS row stride in reals
    int plane_stride = 0;              // LDS plane stride in reals (>= row_stride * rows[0]: spreading tiles pad it for the banks)
    int rows[2] = {1, 1};              // LDS rows per plane, planes
    int64_t elems = 0;                 // LDS reals
    int64_t ntiles = 1;
    int max_items = 1;                 // upper bound of the work items (runs of sorted points) per tile
};

struct TileGeom {
    int blog[kMaxDim] = {0, 0, 0};     // log2(bin size)
    int nb[kMaxDim] = {1, 1, 1};       // bins per dimension
    int64_t nbins = 1;
    TileShapeHost sp;                  // spreading tile: interior only in LDS (Float64)
    TileShapeHost ip;                  // interpolation tile: interior + 2M - 1 halo in LDS
};

// Chooses bins and both tile shapes under the LDS budget; `forced_*` (cells, 0 = automatic) override
// the search.  Returns false if not even the smallest tile fits (-> NUFFT_ERR_LDS_TOO_SMALL).
bool choose_tiles(int D, int M, int ncomp, int real_bytes, const int64_t* Nover, int lds_budget_bytes,
                  int spread_waves, int interp_waves, const int* forced_sp, const int* forced_ip, int bin_log2,
                  TileGeom& g);


}  // namespace nufft

// -------------------------------------------------------------------------------------------
// The plan
// -------------------------------------------------------------------------------------------
struct nufft_plan {
    std::map<void*, std::pair<std::string, int64_t>> allocs;      // device buffers of the plan: pointer -> (bytes)
    std::string opts_text;             // (storage of nufft_plan_options)
    nufft::Options opts;               // development switches of this plan (nufft_params.options; options.h)
    // parameters
    int dtype = NUFFT_F64;
    bool is_complex = false;
    int D = 1;
    int64_t N[3] = {1, 1, 1};
    int M = 4;
    double sigma_req = 2.0;
    int evalmode = NUFFT_EVAL_DIRECT;
    int C = 1;
    bool fftshift = false;
    int device = -1;
    int spread_threads = 256;
    int interp_threads = 256;

    // derived host data
    int64_t Nover[3] = {1, 1, 1};
    int64_t Nout[3] = {1, 1, 1};       // size(p)
    int64_t Nspec[3] = {1, 1, 1};      // dims of the oversampled spectrum (r2c halves dim 0)
    double sigma = 2.0;
    double beta[3] = {0, 0, 0};
    const void* cb_point_weights = nullptr;   // set for the duration of nufft_exec_type{1,2}_cb
    const void* cb_mode_factors = nullptr;
    const void* cb_sticky_weights = nullptr;  // nufft_set_callbacks: in force for the stage-level entry points until reset
    const void* cb_sticky_factors = nullptr;
    int kernel = 0;                    // NUFFT_KERNEL_*
    int point_transform = 0;           // NUFFT_POINT_TRANSFORM_*
    double tau[3] = {0, 0, 0};         // Gaussian: 2 (ℓ Δx)²
    double eval_p0[3] = {0, 0, 0};     // window parameters handed to the kernels (see WindowEval)
    int scale_exp[3] = {0, 0, 0};      // device windows and phi_hat carry a factor 2^scale_exp[d]
    double beta_over_pi_scaled[3] = {0, 0, 0};   // (β/π rounded to T) * 2^scale_exp[d]
    int npoly = 8;
    std::vector<double> coefs[3];      // [npoly][2M]
    std::vector<double> phihat[3];
    std::vector<int64_t> index_map[3];
    nufft::TileGeom tile;
    bool interp_march = false;         // the z-marching interpolation kernel may serve point sets of this plan (march_kernels.h)
    int interp_march_mode = 1;         // NUFFT_INTERP_MARCH latched at plan creation: 0 off, 1 per point set (fitted model), 2 always
    bool debug_tasks = false;          // NUFFT_DEBUG_TASKS latched at plan creation: host check of the task tables after set_points
    nufft::ColumnTasks march_ct{};     // ... its columns and evenly cut tasks
    int num_cus = 256;                 // compute units of the device (one ring workgroup per CU)
    uint32_t* d_march_choice = nullptr;   // [16]: scratch of the task kernels (balance.hip); [2] = 1: the ring serves this point set
    uint32_t* d_march_cols = nullptr;     // [columns] points per column, then [columns + 1] first task of each column
    void* d_march_tasks = nullptr;        // uint2[ntasks + columns]: {column, end layer << 16 | first layer}
    bool interp_fixed = false;         // tile.ip is the compile-time tile of the kernel instantiation
    bool spread_fixed = false;         // tile.sp likewise
    bool spread_cubes = false;         // LDS-tile spreading accumulates cube by cube (v_mfma_f64_4x4x4 + one ds_add_f64 per cube)
    int spread_method = NUFFT_SPREAD_LDS_TILES;   // what nufft_spread launches (NUFFT_SPREAD_*)
    int spread_method_req = NUFFT_SPREAD_AUTO;    // what the caller asked for
    struct Patch {                                // decomposition of the MFMA-patch spreading (patch_kernels.h)
        bool eligible = false;
        int npx = 0, npy = 0, nseg = 0, segl = 0, ntasks = 0, lds_bytes = 0, pby = 0, occ = 2, f32acc = 0, planar = 0;
    } patch;
    uint32_t* d_patch_choice = nullptr;   // [16]: scratch of patch_split_kernel; [2] = 1: this point set is spread by the patches
    uint32_t* d_patch_cols = nullptr;     // [columns] points per patch column, then [columns + 1] first task of each column
    void* d_patch_tasks = nullptr;        // uint2[ntasks]: {column, end layer << 16 | first layer}, rebuilt by every set_points
    int interp_parts = 1;                 // 2: ComplexF64 interpolated part by part by the real ring kernels (NUFFT_INTERP_SPLIT, latched at creation)
    int smarch_parts = 1;                 // 2: complex data part by part through the real window kernel (NUFFT_SMARCH_SPLIT, latched at creation)
    nufft::SMarchPlan smarch{};           // decomposition of the z-marching spreading ring (smarch_kernels.h); eligible = false: none
    uint32_t* d_smarch_choice = nullptr;  // [16]: scratch of the task kernels; [2] = 1: this point set is spread by the ring
    uint32_t* d_smarch_cols = nullptr;    // [columns] points per column, then [columns + 1] first task of each column
    void* d_smarch_tasks = nullptr;       // uint2[table entries]: {column, end layer << 16 | first layer}, rebuilt by every set_points
    void* d_smarch_halo = nullptr;        // halo variant: side buffer of the stencil reach, C x smarch.halo_reals reals
    // Halo variant of the spreading ring: whether the side buffer still has to be added to `us` is DEVICE state (d_smarch_choice[kHaloStateWord],
    // written by the spreading kernel, cleared by whoever adds or voids it), so it follows the stream and a replayed hipGraph.  The host
    // keeps a hint that is exact while every call on the plan is eager, and is used only to skip launches that would find the word 0;
    // once any call on the plan has been captured into a hipGraph (halo_sticky) nothing is skipped any more.
    bool halo_hint = false;
    bool halo_sticky = false;
    bool halo_fuse = true;                // NUFFT_SMARCH_HALO_FUSE (latched at creation): 0 = always the separate add pass
    int wave_slots = 2048;             // resident waves of the patch kernel on this device (CUs x 4 SIMDs x its waves per SIMD)
    void* d_vsorted = nullptr;         // C value vectors in sorted order (MFMA-patch spreading)
    int64_t lds_spread = 0, lds_interp = 0;

    // device data
    void* d_coefs = nullptr;           // T[D][npoly][2M]
    void* d_phihat[3] = {nullptr, nullptr, nullptr};      // T[Nout[d]]
    int32_t* d_index_map[3] = {nullptr, nullptr, nullptr};// out index -> oversampled index
    int32_t* d_inv_map[3] = {nullptr, nullptr, nullptr};  // oversampled index -> out index or -1
    void* d_us = nullptr;              // C grids, contiguous; T[Nover] (real) or complex
    void* d_uhat = nullptr;            // C spectra complex<T>[Nspec] (real plans only)
    int64_t grid_elems = 0;            // elements (of Z) per component in d_us
    int64_t spec_elems = 0;            // complex elements per component in d_uhat (or d_us)
    int64_t pspec_elems = 0;           // complex elements between components of the pruned-path spectrum (compact for complex plans)

    // bin sort
    int64_t Np = -1;
    int64_t Np_capacity = 0;
    int64_t binrank_capacity = 0;      // bytes of d_binrank (sized per point set: see nufft_set_points)
    uint32_t* d_counts = nullptr;      // [ntiles + 1]  histogram
    bool counts_clean = false;         // d_counts is all zero (plan creation; every completed set_points leaves it so)
    uint32_t* d_offsets = nullptr;     // [ntiles + 1]  exclusive scan
    nufft::CoarseSort slab{};          // two-level slab sort (binsort.hip; D = 3 plans without a column-layer sort): table + flag words allocated,
                                       // the slab height chosen per point set (nufft_set_points)
    int slab_fill = 85;                // a slab's average load, percent of what a level-2 workgroup holds, at most (NUFFT_SLAB_FILL)
    int64_t slab_min_points = 0;       // smaller point sets take the fine sort with global atomics (NUFFT_SLAB_MIN_POINTS)
    // adaptive sort choice on plans of the column-layer sort: the rings' per-point-set decisions come back through host-mapped memory
    // (CoarseSort::feedback); after two point sets in a row that a ring handed to the tile kernels (the column-layer attempt then ends in the
    // fine sort with global atomics: 1.8 ms at 1.7e7 folded-normal points against 1.1 ms for the slab sort) set_points takes the slab sort
    // directly, and returns to the column-layer sort as soon as both rings would serve a set again
    uint32_t* sort_feedback = nullptr;       // host-mapped {flag a, flag b, sequence}
    uint32_t sort_seq = 0, sort_seq_seen = 0;
    int sort_miss_streak = 0;
    bool sort_prefer_slab = false;
    bool coarse_now = false;                 // the current point set went through the column-layer path (set_points)
    bool dense_available = false;      // the spreading window's dense-set engine exists for this plan (dmarch_kernels.h)
    bool dense_now = false;            // ... and serves the current point set (set_points: mean bin load >= dense_min)
    int dense_min = 1 << 30;
    int sort_column_pred[2] = {0, 0};  // host-only plans: the column-layer sort a device plan of these parameters would take (predict_sort_column)
    int slab_max_keys = 0;             // keys the level-1 table was allocated for: min(kCoarseMaxKeys, nb[1] * nb[2])
    nufft::CoarseSort coarse{};        // column-layer sort (binsort.hip): enabled on plans whose two rings own the same columns; table allocated
    void* d_binrank = nullptr;         // uint2[Np]: (tile, rank)
    void* d_sorted = nullptr;          // PointRec<T, D>[Np]
    void* d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    // load balance (balance.hip): arrays over the tiles of both tilings, spreading tiles first
    struct Balance {
        uint32_t extra[2] = {0, 0};        // budget of extra slices = extra workgroups in the launch grids
        uint32_t* d_work = nullptr;        // [balance_work_words(nsp + nip)]: work per tile, then the partial sums
        uint32_t* d_nslices = nullptr;     // [nsp + nip + 1]
        uint32_t* d_desc_off = nullptr;    // [nsp + nip + 1]
        void* d_desc = nullptr;            // uint2[nsp + extra_sp + nip + extra_ip]
        uint32_t* d_slots = nullptr;       // [2] slots in use
        void* d_tmp = nullptr;
        size_t tmp_bytes = 0;
    } bal;
    bool balance_enabled = true;
    int64_t workspace_bytes = 0;

    // pruned FFT path (fft_lines.hip): dimension 1 by rocFFT (1-D batched r2c / c2r), higher dimensions by
    // pruned strided passes fused with the deconvolution
    bool pruned_fft = false;
    bool compact_dim1 = false;         // dimension 1 by real_lines_kernel with a compact spectrum (row length N_out1)
    int64_t spec_row = 0;             // row stride (complex elements) of the dimension-1 spectrum and of tmp2 on the pruned path:
                                       // N_out1 padded to 128 bytes for real plans with the compact spectrum (aligned strided passes)
    rocfft_plan_t* fft1_fw = nullptr;
    rocfft_plan_t* fft1_bw = nullptr;
    void* d_tmp2 = nullptr;            // complex<T>[N_out1 * N_out2 * Ñ3] (3-D only)
    void* d_tw_fw[3] = {nullptr, nullptr, nullptr};   // complex<T>[Ñ_d] twiddles exp(-2πi m/Ñ_d)
    void* d_tw_bw[3] = {nullptr, nullptr, nullptr};
    void* d_invphi[3] = {nullptr, nullptr, nullptr};  // T[N_out_d]: 1 / (ϕ̂_d 2^scale_exp_d)
    void* d_one = nullptr;             // T[1] = 1

    // rocFFT
    rocfft_plan_t* fft_fw = nullptr;
    rocfft_plan_t* fft_bw = nullptr;
    rocfft_execution_info_t* fft_info = nullptr;
    void* d_fft_work = nullptr;
    size_t fft_work_bytes = 0;

    // timing
    bool timing = false;
    void* ev_begin[NUFFT_NUM_STAGES] = {};
    void* ev_end[NUFFT_NUM_STAGES] = {};
    bool ev_valid[NUFFT_NUM_STAGES] = {};
};
