// Type-1 spreading on register-resident patches accumulated by the FP64 matrix pipe (gfx950, wave64).
//
// Replaces spread_from_points_shmem_kernel! (reference src/spreading/gpu.jl:237-377) + fill_with_zeros
// (src/NonuniformFFTs.jl:161-167) for 3-D grids of 4-cell bins.  Same arithmetic as the reference — every point adds
// v * w1[j1] * w2[j2] * w3[j3] to the (2M)^3 cells of its stencil — organised as a contraction over points:
//
//     G[(x, y), z] += sum_p  A[(x, y), p] * B[p, z],    A = w1_p[x] * w2_p[y],   B = v_p * w3_p[z]
//
// which v_mfma_f64_4x4x4_4b_f64 evaluates for a 4 x 4 x 4 cube of cells and four points per instruction (four
// independent 4 x 4 x 4 blocks: block = y row of the cube, i = x, j = z, k = point).  Measured on MI355X
// (scripts/microbench6.hip, profiles/round2_microbench.md): 18 cycles per instruction per SIMD, against 8.5 cycles per
// CU for one ds_add_f64 wave instruction (64 cells of ONE point) of the LDS-tile kernel in tile_kernels.h.
//
//   * A wave owns a patch of PBX x PBY cube columns (16 x 4 PBY cells) and marches along dimension 3 through a
//     segment of cube layers.  The accumulators of NCB = number of cube layers a stencil can touch live in
//     registers (MFMA C/D operands); when a bin layer is finished its oldest cube layer is complete, leaves through
//     a small LDS transposition as full 128-byte rows, and the ring of accumulators shifts.  No atomics anywhere
//     (LDS or global), no zero fill of the grid: every cell is written exactly once.
//   * Output-driven like the LDS-tile kernel: a patch visits the points of every bin whose stencils can reach
//     it — per bin layer PBY + NCB - 1 rows of PBX + NCB - 1 bins, each row one contiguous run of the bin-sorted
//     array.  The matrix work of a point is not duplicated by that (each cube belongs to one patch); only its
//     window evaluation and operand set-up are.
//   * Window values are evaluated with the group mapping of tile_kernels.h (WindowEval: next_pow2(2M) lanes own
//     a point, polynomial coefficients in registers, or the direct forms) into zero-padded LDS rows; the operand
//     of a cube is then a plain LDS read at (cube offset + lane coordinate - stencil start): cells outside the
//     stencil read the padding.  Dimensions 2 and 3 are static offsets from one address (the row of bins and
//     the cube layer fix the cube offsets), dimension 1 clamps its index into the padding.
//   * Values arrive gathered in sorted order (gather_values_kernel), records and values of the next chunk of
//     points are prefetched into registers while the current chunk is processed.
//
// Float32 plans evaluate their windows in Float32 and accumulate in Float64 here (as the LDS-tile kernel does); ComplexF32
// plans normally run patch32_kernels.h instead (FP32 matrix pipe, Float32 accumulators, as the reference accumulates).
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "device_common.h"
#include "nufft_mi355x.h"
#include "tile_kernels.h"

namespace nufft {

// Register budget of a patch: accumulators (cubes) per wave and the waves per SIMD the kernel is compiled for.
// Two waves per SIMD and 48 accumulators by default; one wave per SIMD with 120 accumulators (a larger patch: fewer
// visits, each of which re-evaluates 3 x 2M window values per point and sets the operands up again; and whole runs of
// the sorted array per chunk, see chunk_points) where a stencil reaches 4+ cubes per dimension AND carries two
// components, or 7 cubes.  Measured (spread stage at 256^3 -> 512^3, Np = 1e7, two waves vs one): ComplexF64 m = 4 4.9 vs
// 5.8 ms, m = 5 11.5 vs 10.9, m = 6 19.3 vs 18.0, m = 7 20.5 vs 18.0 (with the larger chunks); Float64 m = 7 11.9 vs 12.8,
// m = 8 11.4 vs 12.1, m = 9 18.5 vs 21.2, m = 10 39.3 vs 29.4; C3 (ComplexF32, m = 8) 168 vs 136 ms.
// NUFFT_PATCH_ACC_CAP / NUFFT_PATCH_OCC override both for every instantiation (ablation builds).
// (three components — real plans with ntransforms = 3, below — always take the large configuration: with 48 accumulators
// their patch would be a single row of cube columns.)
constexpr __host__ __device__ int patch_acc_cap(int ncomp, int M) {
#if defined(NUFFT_PATCH_ACC_CAP)
    return NUFFT_PATCH_ACC_CAP;
#else
    return ((ncomp == 2 && M >= 5) || M >= 10 || ncomp >= 3) ? 120 : 48;
#endif
}
constexpr __host__ __device__ int patch_occupancy(int ncomp, int M) {
#if defined(NUFFT_PATCH_OCC)
    return NUFFT_PATCH_OCC;
#else
    return ((ncomp == 2 && M >= 5) || M >= 10 || ncomp >= 3) ? 1 : 2;
#endif
}

constexpr __host__ __device__ int floor_div4(int a) { return a >= 0 ? a / 4 : -((-a + 3) / 4); }

// rows of cube columns per patch: NC * NCB * PBX * PBY accumulators (two VGPRs each) must leave room for the rest
constexpr __host__ __device__ int patch_rows(int ncomp, int ncb, int cap) {
    int r = cap / (ncomp * ncb * 4);
    return r < 1 ? 1 : (r > 4 ? 4 : r);
}

// NC components per point: 1 (real), 2 (complex, interleaved in values and grid) — or, PL = planar, the NC = ntransforms
// separate real value vectors / grids of a real plan spread TOGETHER: the window evaluation, the operand set-up and the
// A = w1 w2 operand are shared, only B = v_c w3 and the accumulators are per component (the reference re-evaluates the
// windows per component, TODO at src/spreading/gpu.jl:293 / src/interpolation/gpu.jl:273).
template <int NC, int M, bool PL = false>
struct PatchCfg {
    static constexpr int L = 2 * M;
    static constexpr int CLO = floor_div4(1 - M);      // cubes a stencil reaches relative to the bin of its point
    static constexpr int CHI = floor_div4(3 + M);
    static constexpr int NCB = CHI - CLO + 1;
    static constexpr int PBX = 4;
    static constexpr int PBY = patch_rows(NC, NCB, patch_acc_cap(NC, M));
    static constexpr int NRB = PBY + NCB - 1;           // rows of bins visited per bin layer
    static constexpr int NACC = NC * NCB * PBX * PBY;
    static constexpr int PADB = 4 - M - 4 * CLO;        // zeros in front of / behind the 2M window values of a row
    static constexpr int PADA = 4 * CHI + 3 - M;
    static constexpr int LW = PADB + L + PADA;
    // the row of dimension 1 is padded for every (lane x, stencil start) pair of the patch, so that its reads need no clamp
    static constexpr int PADXB = 4 * (PBX - CLO) - M;
    static constexpr int PADXA = 4 * PBX + 4 * CHI - M - 1;
    static constexpr int LWX = PADXB + L + PADXA;
    static constexpr int G = next_pow2(L);              // lanes per point during window evaluation
    static constexpr int PPW = kWave / G;
    // points per chunk (staged in LDS): as much of a run of the sorted array as the LDS holds — every chunk pays the
    // commit / prefetch / coefficient reload once (C3: 157 -> 137 ms going from 16 to 48 points).  One wave per SIMD =
    // one workgroup per CU with 160 KiB, two waves per SIMD = two workgroups with 80 KiB each.
    static constexpr int chunk_points() {
        const int occ = patch_occupancy(NC, M);
        const int budget = (occ == 1 ? 160 : 80) * 1024 - 512;
        for (int ch = (occ == 1 ? 64 : 32); ch > 16; ch -= 8)
            if (4 * ((ch + 1) * ((LWX + 2 * LW) * 8 + META) + ch * 48 + 32) + 3 * (M + 4) * L * 8 + 64 <= budget) return ch;
        return 16;
    }
    static constexpr int META = round_up(16 + 8 * NC, 16) < 32 ? 32 : round_up(16 + 8 * NC, 16);   // bytes: {sx, offy, offz, rbx} + NC values
    static constexpr int CH = chunk_points();
    static constexpr int PSTRIDE = (LWX + 2 * LW) * 8 + META;   // bytes per staged point
    static constexpr int WBYTES = (CH + 1) * PSTRIDE;   // + the all-zero point
    static constexpr int ROWLEN = (PL ? 16 : 16 * NC) + 2;   // reals per row of the transposition buffer (+2: banks); planar: one component at a time
    static constexpr int TBYTES = 4 * 4 * PBY * ROWLEN * 8;
    static constexpr int STAGE_PT = 32 + 16;            // staged record (3 coordinates as double + pad)
    static constexpr int WAVE_BYTES = round_up((WBYTES > TBYTES ? WBYTES : TBYTES), 16) + round_up(CH * STAGE_PT, 16);
    static constexpr int NPOLY = M + 4;                 // coefficients per sub-interval (src/Kernels/kaiser_bessel_backwards.jl:98)
    static_assert(PADB >= 1 && PADA >= 1, "padding");
    static constexpr int table_bytes(int real_bytes) { return round_up(3 * NPOLY * L * real_bytes, 16); }
    static constexpr int lds_bytes(int real_bytes, int nwaves) { return table_bytes(real_bytes) + nwaves * WAVE_BYTES; }
};

constexpr int kPatchWaves = 4;                           // waves per workgroup (independent tasks)

struct PatchGeom {
    int npx, npy, nseg, segl;       // patch columns, segments along dimension 3, cube layers per segment
    int ntasks;
};

template <typename T>
struct PlanarPtrs { const T* p[4]; };

template <typename T>
struct PatchArgs {
    TileArgs<T> t;                  // geometry, sorted records, bin offsets, window parameters, grids
    PatchGeom pg;
    const T* vsorted[kMaxCompPerLaunch];   // values in sorted order (gather_values_kernel), NC reals per point
    unsigned long long* prof;              // NUFFT_PATCH_PROFILE builds: cycles per phase, summed over the waves
    const uint32_t* enabled;               // device flag written by set_points (patch_split_kernel); null: always run
    const uint2* tasktab;                  // per point set: {patch column, end layer << 16 | first layer} per task (balance.hip)
};

// Values in sorted order: vs[p] = v[idx[p]] (* weight[idx[p]]: callbacks.nonuniform, src/spreading/gpu.jl:289)
template <typename T, int NC, int REC_BYTES>
__global__ __launch_bounds__(256) void gather_values_kernel(const unsigned char* __restrict__ recs, int idx_off, int64_t np,
                                                           const T* __restrict__ vin, const T* __restrict__ weights,
                                                           T* __restrict__ vout, const uint32_t* __restrict__ enabled) {
    if (enabled && *enabled == 0u) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) {
        const int32_t idx = *reinterpret_cast<const int32_t*>(recs + p * REC_BYTES + idx_off);
        const T w = weights ? weights[idx] : T(1);
        if constexpr (NC == 1) {
            vout[p] = vin[idx] * w;
        } else {
            using V2 = typename std::conditional<sizeof(T) == 8, double2, float2>::type;
            V2 v = reinterpret_cast<const V2*>(vin)[idx];
            v.x *= w; v.y *= w;
            reinterpret_cast<V2*>(vout)[p] = v;
        }
    }
}

// Planar components (ntransforms = NC real value vectors): vs[p * NC + c] = v_c[idx[p]] — one interleaved buffer, so that
// the patch kernel prefetches the NC values of a point with one access pattern
template <typename T, int NC, int REC_BYTES>
__global__ __launch_bounds__(256) void gather_planar_kernel(const unsigned char* __restrict__ recs, int idx_off, int64_t np,
                                                           PlanarPtrs<T> vin, const T* __restrict__ weights, T* __restrict__ vout,
                                                           const uint32_t* __restrict__ enabled) {
    if (enabled && *enabled == 0u) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) {
        const int32_t idx = *reinterpret_cast<const int32_t*>(recs + p * REC_BYTES + idx_off);
        const T w = weights ? weights[idx] : T(1);
#pragma unroll
        for (int c = 0; c < NC; ++c) vout[p * NC + c] = vin.p[c][idx] * w;
    }
}

// N doubles at a stride of S doubles from one LDS address: pairs by ds_read2_b64 (offsets in units of 8 bytes) into
// p[(N + 1) / 2]; an odd count reads its last value twice (no copy out of a register with a read in flight)
typedef double v2f64 __attribute__((ext_vector_type(2)));
template <int O0, int O1>
__device__ __forceinline__ void lds_read2_b64(v2f64& d, uint32_t addr) {
    asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(d) : "v"(addr), "n"(O0), "n"(O1));
}
template <int N, int S, int I = 0>
__device__ __forceinline__ void lds_read_strided64(v2f64 (&p)[(N + 1) / 2], uint32_t addr) {
    if constexpr (I + 1 < N) {
        lds_read2_b64<I * S, (I + 1) * S>(p[I / 2], addr);
        lds_read_strided64<N, S, I + 2>(p, addr);
    } else if constexpr (I < N) {
        lds_read2_b64<I * S, I * S>(p[I / 2], addr);
    }
}

// f(std::integral_constant<int, r>) for the run-time row r in [I, N)
template <int I, int N, typename F>
__device__ __forceinline__ void dispatch_row(int r, F&& f) {
    if constexpr (I < N) {
        if (r == I) f(std::integral_constant<int, I>{});
        else dispatch_row<I + 1, N>(r, f);
    }
}

__device__ __forceinline__ double mfma444(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

template <typename T, bool CPLX, int M, bool OTHERK, int NP = 0>
__global__ __launch_bounds__(kPatchWaves * kWave, patch_occupancy(NP > 0 ? NP : (CPLX ? 2 : 1), M)) void spread_patch_kernel(PatchArgs<T> a) {
    constexpr bool PLANAR = NP > 0;                    // NP real components spread together (comp_id 0 carries all of them)
    static_assert(!PLANAR || !CPLX, "planar components are real");
    constexpr int NC = PLANAR ? NP : (CPLX ? 2 : 1);
    using P = PatchCfg<NC, M, PLANAR>;
    constexpr int L = P::L, CLO = P::CLO, CHI = P::CHI, NCB = P::NCB, PBX = P::PBX, PBY = P::PBY, NRB = P::NRB;
    constexpr int PADB = P::PADB, PADXB = P::PADXB, LW = P::LW, LWX = P::LWX, CH = P::CH, PSTRIDE = P::PSTRIDE;
    constexpr int WX = 0, WY = LWX * 8, WZ = (LWX + LW) * 8, MT = (LWX + 2 * LW) * 8;      // byte offsets inside a staged point
    using WE = WindowEval<T, 1, 3, M, P::G, OTHERK>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const Geom& g = a.t.g;
    const PatchGeom& pg = a.pg;

    if (a.enabled && *a.enabled == 0u) return;      // this point set goes to the LDS-tile kernel (set_points: balance.hip)
    // polynomial coefficients of the window: one copy per workgroup in LDS (re-read by every chunk evaluation)
    T* ctab = reinterpret_cast<T*>(smem);
    for (int i = threadIdx.x; i < 3 * P::NPOLY * L; i += kPatchWaves * kWave) ctab[i] = a.t.coefs[i];
    __syncthreads();

    // ---- task: (patch column px, py; segment) ----
    const int nwg = (int)gridDim.x;
    const int wg = xcd_remap_chunked((int)blockIdx.x, nwg, a.t.xcd_chunk);
    const int task = wg * kPatchWaves + wave;
    if (task >= pg.ntasks) return;
    const int comp_id = blockIdx.y;
    // the task: a patch column and its segment of cube layers — from set_points' table (segments of about equal point
    // count, balance.hip), or segments of equal length
    int px, py, z0, z1;
    if (a.tasktab) {
        const uint2 te = a.tasktab[task];
        px = (int)te.x % pg.npx; py = (int)te.x / pg.npx;
        z0 = (int)(te.y & 0xffffu); z1 = (int)(te.y >> 16);
        if (z1 <= z0) return;                           // a task that received no layers
    } else {
        const int seg = task / (pg.npx * pg.npy);
        px = task % pg.npx; py = (task / pg.npx) % pg.npy;
        z0 = seg * pg.segl; z1 = min(z0 + pg.segl, g.nb[2]);
    }
    const int ncx = min(PBX, g.nb[0] - px * PBX), ncy = min(PBY, g.nb[1] - py * PBY);   // cube columns that exist
    const int X0 = px * PBX * 4;
    const int bx0 = px * PBX, by0 = py * PBY;

    unsigned char* wmem = smem + P::table_bytes((int)sizeof(T)) + wave * P::WAVE_BYTES;                 // staged points / transposition buffer
    unsigned char* stage = wmem + round_up((P::WBYTES > P::TBYTES ? P::WBYTES : P::TBYTES), 16);
    const uint32_t wbase = (uint32_t)(uintptr_t)wmem;

    auto zero_wmem = [&]() __attribute__((always_inline)) {
        for (int o = lane * 16; o < P::WBYTES; o += kWave * 16) *reinterpret_cast<uint4*>(wmem + o) = make_uint4(0, 0, 0, 0);
    };
    zero_wmem();

    // ---- accumulators: [component][ring slot][cube row][cube column] ----
    double acc[NC][NCB][PBY][PBX];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int s = 0; s < NCB; ++s)
#pragma unroll
            for (int y = 0; y < PBY; ++y)
#pragma unroll
                for (int x = 0; x < PBX; ++x) acc[c][s][y][x] = 0.0;

    // window evaluation roles (group mapping)
    const int grp = lane / P::G, q = lane % P::G;
    // matrix roles
    const int mk = lane >> 4, mb = (lane >> 2) & 3, mi = lane & 3;

    const PointRec<T, 3>* sorted = static_cast<const PointRec<T, 3>*>(a.t.sorted);
    const T* vs = a.vsorted[comp_id];
    T* grid = a.t.grid[comp_id];

    // ---- runs of the sorted array: per bin layer, row r (relative bin row r - CHI) has up to two pieces (periodic
    //      wrap along dimension 1); lane t < 4 NRB holds bound (r = t >> 2, piece = (t >> 1) & 1, end = t & 1) ----
    const int gx0 = bx0 - CHI, gx1 = bx0 + ncx - 1 - CLO;            // visited bins along dimension 1 (unwrapped)
    auto load_bounds = [&](int bz) __attribute__((always_inline)) -> uint32_t {
        const int r = lane >> 2, piece = (lane >> 1) & 1, isend = lane & 1;
        const int rb = r - CHI;
        uint32_t val = 0;
        if (lane < 4 * NRB && rb <= ncy - 1 - CLO) {
            int lo, hi;                                              // bins [lo, hi] of this piece, or empty
            if (gx0 < 0) { lo = piece ? 0 : gx0 + g.nb[0]; hi = piece ? gx1 : g.nb[0] - 1; }
            else if (gx1 >= g.nb[0]) { lo = piece ? 0 : gx0; hi = piece ? gx1 - g.nb[0] : g.nb[0] - 1; }
            else { lo = gx0; hi = piece ? -1 : gx1; }
            if (hi >= lo) {
                int by = by0 + rb;
                if (by < 0) by += g.nb[1];
                if (by >= g.nb[1]) by -= g.nb[1];
                int bzw = bz % g.nb[2];
                if (bzw < 0) bzw += g.nb[2];
                const int64_t row = ((int64_t)bzw * g.nb[1] + by) * g.nb[0];
                val = a.t.offsets[row + (isend ? hi + 1 : lo)];
            }
        }
        return val;
    };

    const int bz_first = z0 - CHI, bz_last = z1 - 1 - CLO;

    struct Cursor { int bz, u; uint32_t p, pe; };
    uint32_t bnd = load_bounds(bz_first);
    uint32_t bnd_next = load_bounds(bz_first + 1);
    // next chunk after c (c.u = -1, c.p = c.pe = 0 to start); returns false at the end of the segment
    auto advance = [&](Cursor& c) __attribute__((always_inline)) -> bool {
        c.p += CH;
        if (c.p < c.pe) return true;
        for (;;) {
            ++c.u;
            if (c.u == 2 * NRB) {
                c.u = 0;
                ++c.bz;
                if (c.bz > bz_last) return false;
                bnd = bnd_next;
                bnd_next = load_bounds(c.bz + 1);
            }
            c.p = (uint32_t)__builtin_amdgcn_readlane((int)bnd, 2 * c.u);
            c.pe = (uint32_t)__builtin_amdgcn_readlane((int)bnd, 2 * c.u + 1);
            if (c.p < c.pe) return true;
        }
    };

    // ---- prefetch of a chunk: lane l < n holds record and value of point c.p + l ----
    PointRec<T, 3> pf_rec;
    T pf_v[NC];
    auto issue_prefetch = [&](const Cursor& c) __attribute__((always_inline)) {
        const uint32_t n = min((uint32_t)CH, c.pe - c.p);
        const uint32_t pp = c.p + min((uint32_t)lane, n - 1);
        pf_rec = sorted[pp];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) pf_v[cc] = vs[(int64_t)pp * NC + cc];
    };
    // lane l < CH = point l of the chunk: cell, cell fraction (staged for the window evaluation) and the point's
    // meta data {-8 sx, byte offsets of dimensions 2 and 3, bin along dimension 1} + value, written next to its windows
    auto commit_prefetch = [&]() __attribute__((always_inline)) {
        if (lane < CH) {
            int cell[3];
            double* s = reinterpret_cast<double*>(stage + lane * P::STAGE_PT);
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                cell[d] = cell_of(pf_rec.r[d], g.Nover[d]);
                s[d] = (double)(pf_rec.r[d] - T(cell[d]));
            }
            int sx = cell[0] - (M - 1) - X0;                           // stencil start relative to the patch, unwrapped
            if (sx > g.Nover[0] / 2) sx -= g.Nover[0];
            if (sx < -(g.Nover[0] / 2)) sx += g.Nover[0];
            int4 m;
            m.x = (PADXB - sx) * 8;                                    // first read of dimension 1: cube column 0, lane x = 0
            m.y = (PADB + M - 1 - (cell[1] & 3) + 4 * CLO) * 8;        // cube offset CLO, lane row 0
            m.z = (PADB + M - 1 - (cell[2] & 3) + 4 * CLO) * 8;
            m.w = (sx + (M - 1)) >> 2;                                 // bin of the point relative to the patch
            unsigned char* pw = wmem + lane * PSTRIDE + MT;
            *reinterpret_cast<int4*>(pw) = m;
            *reinterpret_cast<double2*>(pw + 16) = make_double2((double)pf_v[0], NC >= 2 ? (double)pf_v[NC >= 2 ? 1 : 0] : 0.0);
            if constexpr (NC >= 3) *reinterpret_cast<double*>(pw + 32) = (double)pf_v[NC >= 3 ? 2 : 0];
        }
    };

    // ---- retire the oldest cube layer (cz = bz + CLO) once bin layer bz is finished, shift the ring ----
    auto retire = [&](int bz) __attribute__((always_inline)) {
        const int cz = bz + CLO;
#ifndef NUFFT_PATCH_DIRECT_RETIRE
#define NUFFT_PATCH_DIRECT_RETIRE 1
#endif
        if (NUFFT_PATCH_DIRECT_RETIRE && sizeof(T) == 8 && !(PLANAR && NC >= 3) && cz >= z0 && cz < z1) {
            // Float64 grids: every lane stores its own cell (both components of a complex cell together).  The four
            // lanes x = 0..3 of a cube row write 32 (64) contiguous bytes and the four cube columns of the patch complete
            // the 128-byte line in L2 — no LDS transposition, no wipe of the staged points behind it (at one wave per SIMD
            // the transposition's two LDS round trips are serial time).  Measured (spread stage, ms): ComplexF64 m = 4 5.58 -> 5.40,
            // m = 6 18.2 -> 17.9, Float64 m = 6 13.05 -> 12.92, two planar components 5.80 -> 5.67, C2 by patches 4.21 -> 4.06;
            // three planar components spill with it (7.4 -> 8.9 ms) and keep the transposition.
            const int di = lane >> 4, db = (lane >> 2) & 3, dj = lane & 3;
            constexpr int NR = PLANAR ? NC : 1, NI = PLANAR ? 1 : NC;
            const int64_t gz = (int64_t)cz * 4 + dj;
#pragma unroll
            for (int rd = 0; rd < NR; ++rd) {
                T* gr = PLANAR ? a.t.grid[rd] : grid;
#pragma unroll
                for (int y = 0; y < PBY; ++y) {
                    if (y < ncy) {
                        T* row = gr + ((gz * g.Nover[1] + (int64_t)by0 * 4 + 4 * y + db) * g.Nover[0] + X0 + di) * NI;
#pragma unroll
                        for (int x = 0; x < PBX; ++x) {
                            if (x < ncx) {
                                if constexpr (NI == 1) row[4 * x] = (T)acc[PLANAR ? rd : 0][0][y][x];
                                else *reinterpret_cast<double2*>(row + 4 * x * NI) = make_double2(acc[0][0][y][x], acc[NC > 1 ? 1 : 0][0][y][x]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);          // one row's addresses at a time (register pressure)
                }
            }
        } else if (cz >= z0 && cz < z1) {
            wave_lds_fence();
            double* tb = reinterpret_cast<double*>(wmem);
            // D layout of v_mfma_f64_4x4x4_4b: lane 16 i + 4 b + j holds cell (x = i, y = b, z = j) of the cube
            const int di = lane >> 4, db = (lane >> 2) & 3, dj = lane & 3;
            // interleaved components (complex) leave in one round; planar ones (separate grids) one component per round
            constexpr int NR = PLANAR ? NC : 1;            // rounds
            constexpr int NI = PLANAR ? 1 : NC;            // components interleaved within a round
#pragma unroll
            for (int rd = 0; rd < NR; ++rd) {
                if (rd > 0) wave_lds_fence();
#pragma unroll
                for (int c = 0; c < NI; ++c)
#pragma unroll
                    for (int y = 0; y < PBY; ++y)
#pragma unroll
                        for (int x = 0; x < PBX; ++x)
                            tb[(dj * (4 * PBY) + 4 * y + db) * P::ROWLEN + (4 * x + di) * NI + c] = acc[PLANAR ? rd : c][0][y][x];
                wave_lds_fence();
                // rows of 16 NI reals: 8 lanes per row (2 NI reals each), 8 rows per wave instruction
                constexpr int NROWS = 4 * 4 * PBY;
                const int sub = lane >> 3, e0 = (lane & 7) * 2 * NI;
                const int nxr = ncx * 4 * NI, nyr = ncy * 4;
                T* gr = PLANAR ? a.t.grid[rd] : grid;
#pragma unroll
                for (int r0 = 0; r0 < NROWS; r0 += 8) {
                    const int row = r0 + sub;
                    const int pl = row / (4 * PBY), yy = row % (4 * PBY);
                    if (yy < nyr && e0 < nxr) {
                        const double* src = tb + row * P::ROWLEN + e0;
                        const int64_t gz = (int64_t)cz * 4 + pl, gy = (int64_t)by0 * 4 + yy;
                        T* dst = gr + ((gz * g.Nover[1] + gy) * g.Nover[0] + X0) * NI + e0;
                        if constexpr (NI == 1) {
                            if constexpr (sizeof(T) == 8) *reinterpret_cast<double2*>(dst) = make_double2(src[0], src[1]);
                            else *reinterpret_cast<float2*>(dst) = make_float2((float)src[0], (float)src[1]);
                        } else {
                            if constexpr (sizeof(T) == 8) {
                                reinterpret_cast<double2*>(dst)[0] = make_double2(src[0], src[1]);
                                reinterpret_cast<double2*>(dst)[1] = make_double2(src[2], src[3]);
                            } else {
                                *reinterpret_cast<float4*>(dst) = make_float4((float)src[0], (float)src[1], (float)src[2], (float)src[3]);
                            }
                        }
                    }
                }
            }
            wave_lds_fence();
            zero_wmem();                    // the buffer aliases the staged points: restore their zero padding
            wave_lds_fence();
        }
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int y = 0; y < PBY; ++y)
#pragma unroll
                for (int x = 0; x < PBX; ++x) {
#pragma unroll
                    for (int s = 0; s + 1 < NCB; ++s) acc[c][s][y][x] = acc[c][s + 1][y][x];
                    acc[c][NCB - 1][y][x] = 0.0;
                }
    };

    // ---- one chunk of n <= CH staged points of bin row R: window evaluation, then K-batches of four points ----
    auto eval_chunk = [&](int n) __attribute__((always_inline)) {
        // the polynomial coefficients live in registers only while a chunk is evaluated (they are re-read from the
        // LDS table per chunk): the matrix phase needs the room for its accumulators
        WE we;
        we.init(a.t, q, ctab);
        wave_lds_fence();
#pragma unroll 1
        for (int t0 = 0; t0 < n; t0 += P::PPW) {
            const int pt = t0 + grp;
            const double* s = reinterpret_cast<const double*>(stage + min(pt, n - 1) * P::STAGE_PT);
            const T X[3] = {(T)s[0], (T)s[1], (T)s[2]};
            T v[WE::NSLOT];
            we.eval_regs(a.t, X, v);
            unsigned char* pw = wmem + pt * PSTRIDE;
            if (pt < n) {
#pragma unroll
                for (int sl = 0; sl < WE::NSLOT; ++sl)
                    if (we.has[sl]) {
                        const int off = we.dsel[sl] == 0 ? WX + (PADXB + we.jsel[sl]) * 8 : (we.dsel[sl] == 1 ? WY : WZ) + (PADB + we.jsel[sl]) * 8;
                        *reinterpret_cast<double*>(pw + off) = (double)v[sl];
                    }
            }
        }
        wave_lds_fence();
    };

    // K-batches of four points (k = lane >> 4; the all-zero point pads the last one), software-pipelined: the LDS
    // reads of batch i + 1 (operands) and i + 2 (point meta data) are in flight while the MFMAs of batch i issue.
    auto batches = [&](auto RBc, int n) __attribute__((always_inline)) {
        constexpr int RB = decltype(RBc)::value - CHI;                 // relative bin row: cubes RB + CLO .. RB + CHI
        typedef int v4i __attribute__((ext_vector_type(4)));
        typedef double v2d __attribute__((ext_vector_type(2)));
        v4i m;                       // {sx, byte offset dim 2, byte offset dim 3, bin along dim 1 relative to the patch}
        v2d vv;                      // value (re, im) / components 0, 1
        double v3 = 0.0, v3n = 0.0;  // component 2 (planar, NC = 3)
        v2d w3p[(NCB + 1) / 2], w2p[(NCB + 1) / 2], w1p[PBX / 2];      // operand values in register pairs (ds_read2_b64)
        double vre = 0.0, vim = 0.0;
        uint32_t cxmask = 0u;                                          // cube columns the batch can touch
        auto issue_meta = [&](int b0) __attribute__((always_inline)) {
            const int pidx = b0 + mk < n ? b0 + mk : CH;
            const uint32_t ad = wbase + (uint32_t)(pidx * PSTRIDE + MT);
            asm volatile("ds_read_b128 %0, %1" : "=v"(m) : "v"(ad));
            asm volatile("ds_read_b128 %0, %1 offset:16" : "=v"(vv) : "v"(ad));
            if constexpr (NC >= 3) asm volatile("ds_read_b64 %0, %1 offset:32" : "=v"(v3n) : "v"(ad));
        };
        // operands of the batch whose meta data has arrived in (m, vv)
        auto issue_ops = [&](int b0) __attribute__((always_inline)) {
            const int pidx = b0 + mk < n ? b0 + mk : CH;
            const uint32_t pb = wbase + (uint32_t)(pidx * PSTRIDE);
            const int nvalid = max(1, min(4, n - b0));
            // cube columns this batch can touch (points are sorted by bin along dimension 1)
            {
                const int lo = max(__builtin_amdgcn_readlane(m.w, 0) + CLO, 0);
                const int hi = min(__builtin_amdgcn_readlane(m.w, 16 * (nvalid - 1)) + CHI, PBX - 1);
                cxmask = hi >= lo ? (2u << hi) - (1u << lo) : 0u;
#if defined(NUFFT_PATCH_NOGUARD)
                cxmask = (1u << PBX) - 1u;
#endif
            }
            vre = vv.x; vim = vv.y;
            if constexpr (NC >= 3) v3 = v3n;
            // dimension 3: ring slot s <-> cube offset CLO + s; dimension 2: cube offset CLO + o (static offsets from
            // one address each); dimension 1: the window index is clamped into the zero padding
            lds_read_strided64<NCB, 4>(w3p, pb + WZ + (uint32_t)m.z + (uint32_t)mi * 8);
            lds_read_strided64<NCB, 4>(w2p, pb + WY + (uint32_t)m.y + (uint32_t)mb * 8);
            lds_read_strided64<PBX, 4>(w1p, pb + WX + (uint32_t)m.x + (uint32_t)mi * 8);
        };
        auto wait_all = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(m), "+v"(vv));
            if constexpr (NC >= 3) asm volatile("" : "+v"(v3n));
#pragma unroll
            for (int s = 0; s < (NCB + 1) / 2; ++s) asm volatile("" : "+v"(w3p[s]), "+v"(w2p[s]));
#pragma unroll
            for (int cx = 0; cx < PBX / 2; ++cx) asm volatile("" : "+v"(w1p[cx]));
        };
        // (the waits sit at the END of an iteration: registers that an inline-asm read has been issued into must not
        // cross the loop back edge, where the compiler may copy them before the data has arrived)
        issue_meta(0);
        wait_all();
        issue_ops(0);
        issue_meta(4);
        wait_all();
#pragma unroll 1
        for (int b0 = 0; b0 < n; b0 += 4) {
            // operands of this batch: A = w1 w2 per cube column, B = v w3 per ring slot
            double A[NCB][PBX], bz_[NC][NCB];
#pragma unroll
            for (int s = 0; s < NCB; ++s) {
                const double w3s = w3p[s / 2][s % 2];
                bz_[0][s] = w3s * vre;
                if constexpr (NC >= 2) bz_[NC >= 2 ? 1 : 0][s] = w3s * vim;
                if constexpr (NC >= 3) bz_[NC >= 3 ? 2 : 0][s] = w3s * v3;
            }
#pragma unroll
            for (int o = 0; o < NCB; ++o)
#pragma unroll
                for (int cx = 0; cx < PBX; ++cx) A[o][cx] = (RB + CLO + o >= 0 && RB + CLO + o < PBY) ? w1p[cx / 2][cx % 2] * w2p[o / 2][o % 2] : 0.0;
            const uint32_t mask = cxmask;
            issue_ops(b0 + 4);
            issue_meta(b0 + 8);
#if !defined(NUFFT_PATCH_NOSCHEDBAR)
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int cx = 0; cx < PBX; ++cx) {
                if (mask & (1u << cx)) {
#pragma unroll
                    for (int o = 0; o < NCB; ++o) {
                        const int cy = RB + CLO + o;
                        if (cy >= 0 && cy < PBY) {
#pragma unroll
                            for (int c = 0; c < NC; ++c)
#pragma unroll
                                for (int s = 0; s < NCB; ++s) acc[c][s][cy][cx] = mfma444(A[o][cx], bz_[c][s], acc[c][s][cy][cx]);
                        }
                    }
                }
            }
            wait_all();
        }
    };

#if defined(NUFFT_PATCH_PROFILE)
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define NUFFT_PH(i) do { const unsigned long long tn = __builtin_readcyclecounter(); tph[i] += tn - tlast; tlast = tn; } while (0)
#else
#define NUFFT_PH(i) do { } while (0)
#endif
    // ---- main loop over the chunks of the segment ----
    Cursor nxt{bz_first, -1, 0u, 0u};
    bool has = advance(nxt);
    if (has) issue_prefetch(nxt);
    int bz_done = bz_first;
    while (has) {
        const Cursor cur = nxt;
        NUFFT_PH(0);
        while (bz_done < cur.bz) { retire(bz_done); ++bz_done; }      // (wipes the staged points: before the commit)
        NUFFT_PH(1);
        wave_lds_fence();
        commit_prefetch();
        NUFFT_PH(2);
        has = advance(nxt);
        if (has) issue_prefetch(nxt);
        NUFFT_PH(0);
        const int n = (int)min((uint32_t)CH, cur.pe - cur.p);
        eval_chunk(n);
        NUFFT_PH(3);
        dispatch_row<0, NRB>(cur.u >> 1, [&](auto Rc) __attribute__((always_inline)) { batches(Rc, n); });
        NUFFT_PH(4);
    }
    while (bz_done <= bz_last) { retire(bz_done); ++bz_done; }
    NUFFT_PH(1);
#if defined(NUFFT_PATCH_PROFILE)
    if (lane == 0 && a.prof)
        for (int i = 0; i < 5; ++i) atomicAdd(a.prof + i, tph[i]);
#endif
}

}  // namespace nufft
