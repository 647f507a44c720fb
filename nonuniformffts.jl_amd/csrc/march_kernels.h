// Type-2 interpolation on a z-marching LDS ring (gfx950, wave64): the second interpolation engine.
//
// Replaces interpolate_to_points_shmem_kernel! (reference src/interpolation/gpu.jl:211-328, tile load :331-355) for 3-D
// plans, with the same arithmetic per point (every point gathers its (2M)^3 stencil once).  interp_tile_kernel loads a
// padded box (n + 2M - 1)^3 per tile, so every grid cell is fetched 2.6x (C2: 20 x 20 x 16 interior) to 12.5x (C3:
// 16 x 12 x 8, ComplexF32, M = 8) through L2.  Here a workgroup owns a COLUMN of the grid — (n1 + 2M - 1) x (n2 + 2M - 1)
// cells in x, y — and marches along z through a segment of bin layers: LDS holds a ring of RZ = 2M - 1 + 4 planes, the
// window of the stencils of one bin layer (4 planes of cells); per layer 4 new planes replace the 4 oldest.  The halo is
// paid in x and y only (C2: 32 x 32 interior, 1.49x; C3: 16 x 16, 3.75x), and the planes of the next layer are fetched
// into registers while the points of the current layer are gathered, so the load latency is off the critical path.
//
// Points: the bins of a layer inside the column are runs of the bin-sorted array (one per row of bins); waves pull
// passes of PPW points from an LDS counter.  The gather itself is the one of interp_tile_kernel (group mapping, DPP
// broadcasts of the window values for real data, DPP / permlane reduction) with the plane index taken modulo RZ.
//
// Tasks: a workgroup's column and segment of bin layers come from a table that set_points builds per point set on the
// device (balance.hip): segments of equal length for uniform point sets, of about equal point count (column quantiles,
// at most kSegMax layers) otherwise.  A point is gathered once, by the task of its own column and layer, so short
// segments in dense regions cost only window loads.  Point sets whose heaviest task would still hold the chip up —
// and grids with too few columns to fill it — go to interp_tile_kernel, which shares heavy tiles between workgroups:
// a device flag decides, no host read-back.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "device_common.h"
#include "nufft_mi355x.h"
#include "tile_kernels.h"

namespace nufft {

struct MarchGeom {
    int ntx, nty, nseg, segl;       // columns along x, y; segments along z and bin layers per segment when cut evenly
    int ntasks;                     // entries of the task table = workgroups of the launch
    const uint32_t* flag;           // per point set (set_points, balance.hip): 1 = the ring serves it, 0 = interp_tile_kernel
    const uint2* tasktab;           // per point set: {column, end layer << 16 | first layer} per task
    int n1, n2;                     // spreading ring (smarch_kernels.h): the column chosen for this grid (<= its compile-time one)
    void* halo;                     // ... its halo variant: side buffer of the stencil reach (reals), component stride in reals
    int64_t halo_comp;
    int parts;                      // spreading ring, complex data through the real kernel (smarch_kernels.h): 2 = blockIdx.y counts (component,
                                    // part) pairs — part 0 / 1 spreads the real / imaginary parts of the values into the real / imaginary
                                    // parts of the interleaved grid (value and cell stride 2 reals); 0 / 1: plain components
    const uint32_t* coarse_a;       // interpolation ring on plans of the column-layer sort (CoarseSort, kernels.h): both nonzero = this point set
    const uint32_t* coarse_b;       // is column-layer sorted — interp_march_staged_kernel serves it, interp_march_kernel returns (null: no such plan)
    uint32_t* halo_state;           // spreading ring, halo variant: device word the kernel sets to 1 when it has written the side buffer (0 when the
                                    // tile kernel served the point set): "the reach has not been added to the grid yet" — the consumers are gated on it,
                                    // so the state travels with the stream / a replayed hipGraph, not with the host (plan.cpp: halo_hint)
};

constexpr int kMarchMaxRows = 16;   // rows of bins of a column (n2 <= 64)

template <typename T, bool CPLX, int M, bool POLY = true, bool STG = false>       // STG: configuration of interp_march_staged_kernel
struct MarchCfg {
    static constexpr int NC = CPLX ? 2 : 1;
    static constexpr int L = 2 * M, HALO = L - 1;
#ifndef NUFFT_MARCH_KL
#define NUFFT_MARCH_KL 1
#endif
    static constexpr int KL = NUFFT_MARCH_KL;           // bin layers per phase (one barrier pair and one plane fetch per phase)
    static constexpr int BZ = 4 * KL;                   // planes per phase
    static constexpr int RZ = HALO + BZ;                // ring depth
    // Workgroup size: 16 waves per CU leave 128 registers per lane, 8 waves 256.  With one instantiation per window evaluation (the other
    // mode's code and registers gone) almost every kernel fits 128 registers and gains from the second wave per SIMD — interpolation
    // stage, 256^3 -> 512^3, Np = 1e7, 8 -> 16 waves (profiles/round4_c_interp_threads.log): Float64 m = 4 1.52 -> 1.27 ms (Direct 1.85 ->
    // 1.56), m = 5 3.80 -> 3.00; Float32 m = 8 4.84 -> 3.67; ComplexF64 m = 4 4.19 -> 2.76; ComplexF32 m = 4 1.70 -> 1.25.  The
    // exceptions spill (polynomial window, wide stencils: the 2M window values and M + 3 coefficients per slot stay in registers):
    // Float64 m = 6 (34 registers spilled) 4.01 -> 4.69, m = 8 (68) 5.83 -> 15.1; ComplexF64 m = 8 20.4 -> 30.2 — those keep 8 waves with
    // 256 registers each.  ComplexF32 m = 8 (C3's kernel) spilled 18 registers with its row groups of 8 (6.07 -> 6.06 ms) and none with
    // groups of 4: C3 interpolation 52.3 -> 45.0 ms.
    static constexpr int threads_rule() {
#if defined(NUFFT_MARCH_FORCE_1024)
        return 1024;
#elif defined(NUFFT_MARCH_FORCE_512)
        return 512;
#else
        // staged kernel, polynomial window, Float64: the 24 coefficients (48 registers) beside the prefetched planes and the record piece do
        // not fit 128 registers — 22 spilled, reloaded in the serial section between the two barriers of a chunk: 1.67 ms at C2 against
        // 1.26 ms for the plain kernel; 12 waves have 170 registers each
#ifndef NUFFT_STAGED_POLY_THREADS
#define NUFFT_STAGED_POLY_THREADS 768
#endif
        if (STG && POLY && sizeof(T) == 8) return (!CPLX && M >= 6) ? 512 : NUFFT_STAGED_POLY_THREADS;      // (m = 6, 7: 15 / 9 registers spilled at 12 waves)
        // round 6, the staged kernel for every (T, M): where it spilled at 16 waves (hipcc -S: Float64 Direct() m = 3, 6, 7: 15 / 17 / 6 registers;
        // Float32 m <= 3: 6 - 36, m = 6 polynomial: 10; ComplexF32 m <= 3: 21 - 38, m = 4 / 6 polynomial: 10 / 16) it is slower than the plain ring
        // (interpolation 256^3 -> 512^3, Np = 1e7: Float32 m = 2 polynomial 0.97 against 0.55 ms, Float64 m = 3 Direct() 1.64 against 1.42) — 12 waves there
        if (STG && sizeof(T) == 8 && !POLY) return (M == 3 || M >= 6) ? 768 : 1024;
        if (STG && sizeof(T) == 4 && !CPLX) return (M <= 3 || (POLY && M == 6)) ? 768 : 1024;
        if (STG && sizeof(T) == 4 && CPLX) return (M <= 3 || (POLY && (M == 4 || M == 6))) ? 768 : 1024;
        if (!POLY) return 1024;                                 // Direct(): no spills at 128 registers in any instantiation
        if (sizeof(T) == 8) return M <= (CPLX ? 6 : 5) ? 1024 : 512;
        return 1024;                                            // Float32, ComplexF32 (m = 8: with row groups of 4, below)
#endif
    }
    static constexpr int THREADS_PREFERRED = threads_rule();
    // Complex data with at most 16 lanes per stencil row: ONE lane per j1 gathers both components (64- / 128-bit LDS reads,
    // v_pk_fma_f32 for ComplexF32) — half the wave instructions per point of the (j1, component) lane mapping, and the
    // window values can stay in registers as for real data.
    // (ComplexF32 only: 128-bit reads of ComplexF64 pairs run the LDS at a quarter of its rate, scripts/microbench7.hip —
    // measured here: ComplexF64 m = 4 interpolation 5.5 ms paired against 4.4 ms with the tile kernel)
    static constexpr bool PAIR = CPLX && sizeof(T) == 4 && (next_pow2(L) == 8 || next_pow2(L) == 16);   // (window values in registers: REGW)
    using GP = Grp<PAIR ? 1 : NC, M>;
    static constexpr bool REGW = (!CPLX || PAIR) && (GP::G == 8 || GP::G == 16);   // window values stay in registers (DPP broadcasts)
    static_assert(!PAIR || REGW, "the paired gather exists in the register-window form only");
    // A row of 16 lanes reads 16 contiguous nodes = HALF the banks its 32-lane LDS group spans, so the two points that
    // share a group collide on every read unless their rows happen to be complementary (C3: LDS array busy twice the
    // ideal time, 80 % of the kernel).  ZP = 2: both rows of a group belong to ONE point and gather alternate stencil
    // planes; the plane stride is padded to an odd multiple of the row's bytes, so the two rows always cover the
    // group's banks exactly once.  (Two points per wave pass instead of four: the window evaluation is done by both rows.)
#ifndef NUFFT_MARCH_ZP
#define NUFFT_MARCH_ZP 2
#endif
    static constexpr int ZP = (REGW && GP::G == 16 && L == 16) ? NUFFT_MARCH_ZP : 1;   // (measured: M = 5..7 lose 6-17 % with their 10-14 of 16 lanes)
    static constexpr int PPW = GP::PPW / ZP;            // points per wave pass
    static constexpr int ROW_BYTES = GP::G * (PAIR ? 2 : 1) * (int)sizeof(T);
    // Staged kernel, Float64 real data, one row of G lanes per point: the records of a chunk are put in BANK order, not bin order — the 32-lane
    // LDS group of a 64-bit read holds 32 / G points whose G-lane rows (64 or 128 bytes) collide unless their start addresses differ by
    // exact multiples of the row's bytes; with points of one bin (or in any spatial order) they almost never do, and half of the gather's
    // LDS array cycles are conflicts (profiles/round4_sq_counters.md, round5_c_*_sq.json: SQ_LDS_BANK_CONFLICT 2.1e8 of 4.2e8).  The start
    // address of a point's rows is (s0 + RS s1) reals modulo the 32-real bank period — the same for every stencil row and plane once the
    // plane stride is a multiple of the period — so points with equal phase modulo G and distinct phase / G form conflict-free groups.
#ifndef NUFFT_STAGED_BANK_ORDER
#define NUFFT_STAGED_BANK_ORDER 0
#endif
    static constexpr bool BANK_ORDER = STG && NUFFT_STAGED_BANK_ORDER && sizeof(T) == 8 && !CPLX && REGW && ZP == 1 && (GP::G == 8 || GP::G == 16);
    static constexpr int pad_plane(int ps) {            // plane stride in reals: = ROW_BYTES (mod 2 ROW_BYTES) bytes
        if (BANK_ORDER) return (ps + 31) / 32 * 32;     // a whole number of bank periods: the phase of a point is the same in every plane
        if (ZP == 1) return ps;
        const int b = ps * (int)sizeof(T), m = 2 * ROW_BYTES;
        return ps + ((ROW_BYTES - b % m + m) % m) / (int)sizeof(T);
    }
    // Row stride in reals.  At M = 6 (12 of 16 lanes per row) a stride congruent to the row's width modulo the 128-byte half of the bank
    // period spreads the rows of the four points of an LDS group better (measured, interpolation stage 256^3 -> 512^3, Np = 1e7: Float64 3.98 ->
    // 3.60 ms, Float32 2.89 -> 2.75, ComplexF32 3.74 -> 3.31; ComplexF64 8.86 -> 8.94 and every other M within +- 3 % or worse: Float32
    // m = 8 3.71 -> 4.24): padded there only.
#ifndef NUFFT_MARCH_RS_PAD
#define NUFFT_MARCH_RS_PAD (M == 6 && !(CPLX && sizeof(T) == 8))
#endif
    static constexpr int row_stride_of(int n1) { return (NUFFT_MARCH_RS_PAD) ? padded_row_stride(NC * (n1 + HALO), NC * L, (int)sizeof(T)) : NC * (n1 + HALO); }
    static constexpr int strip_bytes() { return REGW ? 0 : round_up(GP::PPW * 3 * L * (int)sizeof(T), 16); }   // (REGW = false: ZP = 1)
    // [runs of the segment: layer x row][passes of the longest run per layer][pass counter, flag]
    // (staged kernel: a layer of the column is ONE run — one row per layer)
    static constexpr int RUN_ROWS = STG ? 1 : kMarchMaxRows;
    static constexpr int table_bytes(int segl) { return round_up(RUN_ROWS * segl * 8 + segl * 4 + 64, 16); }
    static constexpr int kSegMax = 64;
    // staged kernel: the stage of the next chunk's records (16 bytes per thread), two sets of 64 bin counters
    static constexpr int staged_extra_for(int threads) { return STG ? 16 * threads + 2 * 64 * 4 + 64 : 0; }
    static constexpr int fixed_bytes_for(int threads) { return table_bytes(kSegMax) + threads / kWave * strip_bytes() + 64 + staged_extra_for(threads); }
    // column interior (n1, n2): multiples of the bin edge, minimal halo amplification within the LDS budget
    struct Dims { int n1, n2; };
    static constexpr Dims search(int threads) {
        Dims best{0, 0};
        double best_cost = 1e300;
        for (int n2 = 4; n2 <= 4 * kMarchMaxRows; n2 += 4)
            for (int n1 = 4; n1 <= 64; n1 += 4) {
                const long bytes = (long)pad_plane(row_stride_of(n1) * (n2 + HALO)) * RZ * (long)sizeof(T) + fixed_bytes_for(threads);
                if (bytes > 163840 - 256) continue;
                // registers of the per-thread plane prefetch (the planes of the next layer are in flight during the gather)
                if (((long)BZ * NC * (n1 + HALO) * (n2 + HALO) + threads - 1) / threads * (long)(sizeof(T) / 4) > 32) continue;
                double cost = (double)(n1 + HALO) / n1 * (double)(n2 + HALO) / n2;
                // columns whose edge divides the common power-of-two grid sizes leave no partial column
                if (512 % n1) cost *= 1.03;
                if (512 % n2) cost *= 1.03;
                cost -= 1e-6 * n1;
                if (cost < best_cost) { best_cost = cost; best = Dims{n1, n2}; }
            }
        return best;
    }
    // (16 waves carry twice the strips: where no column fits beside them — ComplexF64, M = 9, Direct — 8 waves it is)
    static constexpr int THREADS = (THREADS_PREFERRED == 1024 && search(1024).n1 == 0) ? 512 : THREADS_PREFERRED;
    static constexpr int NW = THREADS / kWave;
    static constexpr int fixed_bytes() { return fixed_bytes_for(THREADS); }
    static constexpr Dims DIMS = search(THREADS);
    static constexpr int N1 = DIMS.n1, N2 = DIMS.n2;
    static constexpr int P1 = N1 + HALO, P2 = N2 + HALO;
    static constexpr int RS = row_stride_of(N1);        // row stride in reals
    static constexpr int PS = RS * P2;                  // reals per plane
    static constexpr int PSP = pad_plane(PS);           // plane stride in the ring
    static constexpr int RING_BYTES = round_up(RZ * PSP * (int)sizeof(T), 16);
    static constexpr int lds_bytes() { return RING_BYTES + fixed_bytes(); }
    static constexpr int NPF = (BZ * PS + THREADS - 1) / THREADS;    // prefetched reals per thread and layer
    static constexpr bool FITS = N1 > 0;                // (ComplexF64 at M = 10: not even a 4 x 4 column fits 160 KiB)
    // STAGED variant (column-layer sorted point sets: the points of a bin layer of the column are ONE run, in no particular order):
    // the records of the next chunk of points travel global -> registers -> LDS while the current chunk is gathered, and are put
    // in bin order on the way (counting sort over <= 64 keys), so that the points of a wave pass share stencil rows as they do
    // after the fine sort.  One 16-byte piece per thread; a chunk never spans layers.
    static constexpr int REC_BYTES = (int)sizeof(PointRec<T, 3>);
    static constexpr int PIECES = REC_BYTES / 16;
    static constexpr int STAGE_RECS = THREADS / PIECES;
    static constexpr int STAGE_BYTES = STAGE_RECS * REC_BYTES;          // = 16 THREADS
    static constexpr int NBX = FITS ? N1 / 4 : 1, NBY = FITS ? N2 / 4 : 1;
    static constexpr int key_shift(int nbx, int nby, bool for_y) {      // coarsen the bins until at most 64 keys are left
        int sx = 0, sy = 0;
        while (((nbx + (1 << sx) - 1) >> sx) * ((nby + (1 << sy) - 1) >> sy) > 64) { if (((nbx + (1 << sx) - 1) >> sx) >= ((nby + (1 << sy) - 1) >> sy)) ++sx; else ++sy; }
        return for_y ? sy : sx;
    }
    static constexpr int KSX = key_shift(NBX, NBY, false), KSY = key_shift(NBX, NBY, true), KNX = (NBX + (1 << KSX) - 1) >> KSX;
    // (STG: the column search above has left room for the stage and the counters; they sit behind the strips)
    static constexpr int staged_extra_bytes() { return staged_extra_for(THREADS); }
    static constexpr int staged_lds_bytes() { return lds_bytes(); }
    static constexpr int stage_offset() { return lds_bytes() - staged_extra_bytes(); }
    static constexpr bool FITS_STAGED = STG && FITS && staged_lds_bytes() <= 163840 - 256 && (REC_BYTES % 16 == 0);
};

// A task is a column and a segment of its bin layers (at most kSegMax) from set_points' table: segments of about equal
// point count, or of equal length for uniform point sets (balance.hip).
template <typename T, bool CPLX, int M, bool POLY>
__global__ __launch_bounds__((MarchCfg<T, CPLX, M, POLY>::THREADS)) void interp_march_kernel(TileArgs<T> a, MarchGeom mg) {
    constexpr bool STAGED_KERNEL = false;
#include "march_setup.inc"
    constexpr int KL = C::KL;
    int pm = 0;                                         // (BZ * phase) mod RZ: slot of the first plane of the window
    const int nphase = (nlay + KL - 1) / KL;
    for (int ph = 0; ph < nphase; ++ph) {
        // ---- planes of the next phase into registers (they replace the BZ oldest once this phase is done) ----
        T pf[NPF];
        const bool more = ph + 1 < nphase;
        if (more) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                pf[u] = T(0);
                if (pf_el[u] >= 0) {
                    int gz = zbase + RZ + BZ * ph + (pf_el[u] >> 24);
                    if (gz >= g.Nover[2]) gz -= g.Nover[2];
                    pf[u] = grid[(int64_t)gz * plane_reals + pf_off[u]];
                }
            }
        }
        // ---- points of this phase's bin layers: passes of PPW points, pulled from a counter; item -> (layer, row, pass of
        //      that row), up to the pass count of the phase's longest run (shorter rows yield empty items) ----
        const int lay0 = ph * KL, nl = min(KL, nlay - lay0);
        int mp = 0;
#pragma unroll
        for (int kl = 0; kl < KL; ++kl) mp = max(mp, kl < nl ? maxp[lay0 + kl] : 0);
        const int nrl = nrows * nl;
        const int nitems = mp * nrl;
        for (;;) {
            int item = 0;
            if (lane == 0) item = atomicAdd(counter, 1);
            item = __builtin_amdgcn_readfirstlane(item);
            if (item >= nitems) break;
            const int rl = item % nrl, kl = rl / nrows;
            const uint2 pr = runs[(lay0 + kl) * C::RUN_ROWS + rl % nrows];
            const uint32_t p0 = pr.x + (uint32_t)(item / nrl) * PPW, p1 = pr.y;
            if (p0 >= p1) continue;
            const uint32_t p = p0 + grp;
            const bool have = p < p1;
            const PointRec<T, 3> rec = sorted[min(p, p1 - 1)];
#include "march_gather.inc"
        }
        lds_barrier();                                   // every wave has finished with this phase's window (no wait for the value stores)
        if (more) {
            if (tid == 0) counter[0] = 0;
            // the BZ new planes take the slots of the BZ oldest: slots pm .. pm + BZ - 1 (mod RZ)
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                if (pf_el[u] >= 0) {
                    int slot = pm + (pf_el[u] >> 24);
                    if (slot >= RZ) slot -= RZ;
                    ring[slot * PSP + (pf_el[u] & 0xffffff)] = pf[u];
                }
            }
            pm += BZ;
            if (pm >= RZ) pm -= RZ;
            lds_barrier();
        }
    }
}

// The same kernel for column-layer sorted point sets (CoarseSort, kernels.h): a bin layer of the column is ONE run of the sorted
// array (row 0 of the table; the other rows are empty), in no particular order, cut into chunks of STAGE_RECS records.  While the waves
// gather the chunk staged in LDS, every thread holds one 16-byte piece of the next chunk in registers; at the chunk boundary the
// records are counted by bin (LDS atomics before the barrier, the 64-key scan redone by every wave after it: no extra barrier) and
// written to the stage in bin order — the points of a wave pass then share stencil rows as they do after the fine sort (without
// the ordering the 64-bit LDS reads of a pass conflict more: + 0.28 ms at C2, round 4), and no pass waits for a record from memory.
template <typename T, bool CPLX, int M, bool POLY>
__global__ __launch_bounds__((MarchCfg<T, CPLX, M, POLY, true>::THREADS)) void interp_march_staged_kernel(TileArgs<T> a, MarchGeom mg) {
    constexpr bool STAGED_KERNEL = true;
#include "march_setup.inc"
    constexpr int KL = C::KL;
    static_assert(KL == 1, "staged records: one bin layer per phase");
    int pm = 0;                                         // (BZ * layer) mod RZ: slot of the first plane of the window
    constexpr int PIECES = C::PIECES, SREC = C::STAGE_RECS;
    typedef uint32_t U4 __attribute__((ext_vector_type(4)));
    unsigned char* stage = smem + C::stage_offset();
    uint32_t* cnt = reinterpret_cast<uint32_t*>(stage + C::STAGE_BYTES);        // [2][64]
    const PointRec<T, 3>* staged = reinterpret_cast<const PointRec<T, 3>*>(stage);
    const int myrec = tid / PIECES, mysub = tid % PIECES;
    // (run bounds through readfirstlane: uniform values in scalar registers — as vector registers they cost this kernel, which sits
    // at the 128-register limit of 16 waves per CU, spills in the serial section between the two barriers of a chunk)
    auto run_of = [&](int lay, int& r0, int& n) __attribute__((always_inline)) {
        const uint2 pr = runs[lay * C::RUN_ROWS];
        r0 = __builtin_amdgcn_readfirstlane((int)pr.x);
        n = __builtin_amdgcn_readfirstlane((int)(pr.y - pr.x));
    };
    auto chunk_records = [&](int lay, int ch) __attribute__((always_inline)) -> int {
        int r0, n;
        run_of(lay, r0, n);
        const int left = n - ch * SREC;
        return left < SREC ? (left > 0 ? left : 0) : SREC;
    };
    auto chunks_of = [&](int lay) __attribute__((always_inline)) -> int {
        int r0, n;
        run_of(lay, r0, n);
        return n > 0 ? (n + SREC - 1) / SREC : 1;
    };
    // the piece of this thread in chunk `ch` of layer `lay` (n: records of the chunk; uniform) — issued, not waited for
    auto load_piece = [&](int lay, int ch, int n) __attribute__((always_inline)) -> U4 {
        U4 v = U4{0u, 0u, 0u, 0u};
        if (myrec < n) {
            int r0, nn;
            run_of(lay, r0, nn);
            v = *reinterpret_cast<const U4*>(reinterpret_cast<const unsigned char*>(sorted + (uint32_t)(r0 + ch * SREC)) + (size_t)tid * 16);
        }
        return v;
    };
    // bin key of the record whose first piece this thread holds (r1, r2 sit in the first 16 bytes for both precisions)
    auto key_of = [&](const U4& v) __attribute__((always_inline)) -> int {
        T r0, r1;
        if constexpr (sizeof(T) == 8) {
            r0 = __builtin_bit_cast(T, ((unsigned long long)v[1] << 32) | v[0]);
            r1 = __builtin_bit_cast(T, ((unsigned long long)v[3] << 32) | v[2]);
        } else {
            r0 = __builtin_bit_cast(T, v[0]);
            r1 = __builtin_bit_cast(T, v[1]);
        }
        const int s0 = cell_of(r0, g.Nover[0]) - org1, s1 = cell_of(r1, g.Nover[1]) - org2;
#ifdef NUFFT_STAGED_NOSORT
        return 0;                                       // (ablation build: the records stay in arrival order)
#endif
        if constexpr (C::BANK_ORDER) {
            // key = (phase mod G) (32 / G) + phase / G: the sub-classes of one alignment class are neighbouring keys
            const int t = (s0 + s1 * RS) & 31;
            return (t % GP::G) * (32 / GP::G) + t / GP::G;
        }
        return (((s0 >> 2) >> C::KSX) + ((s1 >> 2) >> C::KSY) * C::KNX) & 63;
    };
    // counting sort, second half: start of every key from the counters (each wave scans them for itself), then the piece goes to its place
    auto place_piece = [&](const U4& v, int key, uint32_t rank, int n, const uint32_t* cn) __attribute__((always_inline)) {
        const uint32_t c = cn[lane];
        uint32_t dest;
        if constexpr (C::BANK_ORDER) {
            // alignment class = Q neighbouring keys (its Q sub-classes): the first m = min count records of every sub-class interleave into
            // conflict-free groups [class][rank][sub-class]; what is left over follows behind all groups, in key order
            constexpr int Q = 32 / GP::G;
            uint32_t m = min(c, (uint32_t)__shfl_xor((int)c, 1, kWave));
            if constexpr (Q == 4) m = min(m, (uint32_t)__shfl_xor((int)m, 2, kWave));
            uint32_t qi = (lane % Q == 0) ? Q * m : 0u, li = c - m;
            const uint32_t q0 = qi, l0 = li;
            for (int o = 1; o < kWave; o <<= 1) {
                const uint32_t tq = __shfl_up(qi, o, kWave), tl = __shfl_up(li, o, kWave);
                if (lane >= o) { qi += tq; li += tl; }
            }
            const uint32_t nquad = (uint32_t)__builtin_amdgcn_readlane((int)qi, kWave - 1);
            const uint32_t mk = (uint32_t)__shfl((int)m, key, kWave);
            const uint32_t qb = (uint32_t)__shfl((int)(qi - q0), key & ~(Q - 1), kWave), lb = (uint32_t)__shfl((int)(li - l0), key, kWave);
            dest = rank < mk ? qb + Q * rank + (uint32_t)(key & (Q - 1)) : nquad + lb + (rank - mk);
        } else {
        uint32_t incl = c;
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += t;
        }
        const uint32_t start = __shfl(incl - c, key, kWave);
        dest = start + rank;
        }
        if constexpr (PIECES > 1) dest = __shfl(dest, lane & ~(PIECES - 1), kWave);       // the record's other pieces follow its first
        if (myrec < n) *reinterpret_cast<U4*>(stage + (size_t)dest * C::REC_BYTES + mysub * 16) = v;
    };
    int lay = 0, ch = 0, par = 1;                       // current layer and chunk; counter set of the next chunk
    int cur_n = chunk_records(0, 0);                    // records staged
    {
        if (tid < 128) cnt[tid] = 0u;
        __syncthreads();
        const U4 v = load_piece(0, 0, cur_n);
        int key = 0;
        uint32_t rank = 0u;
        if (mysub == 0 && myrec < cur_n) { key = key_of(v); rank = atomicAdd(&cnt[key], 1u); }
        __syncthreads();
        place_piece(v, key, rank, cur_n, cnt);
        __syncthreads();
    }
    for (;;) {
        // ---- what comes next: another chunk of this layer, or the first chunk of the next one with its four new planes ----
        int nlay2 = lay, nch2 = ch + 1;
        if (nch2 >= chunks_of(lay)) { nlay2 = lay + 1; nch2 = 0; }
        const bool has_next = nlay2 < nlay, more = has_next && nlay2 != lay;
        const int next_n = has_next ? chunk_records(nlay2, nch2) : 0;
        U4 pc = U4{0u, 0u, 0u, 0u};
        if (has_next) pc = load_piece(nlay2, nch2, next_n);
        T pf[NPF];
        if (more) {
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                pf[u] = T(0);
                if (pf_el[u] >= 0) {
                    int gz = zbase + RZ + BZ * lay + (pf_el[u] >> 24);
                    if (gz >= g.Nover[2]) gz -= g.Nover[2];
                    pf[u] = grid[(int64_t)gz * plane_reals + pf_off[u]];
                }
            }
        }
        // ---- the staged chunk: passes of PPW points, pulled from a counter ----
        const int nitems = (cur_n + PPW - 1) / PPW;
        for (;;) {
            int item = 0;
            if (lane == 0) item = atomicAdd(counter, 1);
            item = __builtin_amdgcn_readfirstlane(item);
            if (item >= nitems) break;
            constexpr int kl = 0;
            const int p = item * PPW + grp;
            const bool have = p < cur_n;
            const PointRec<T, 3> rec = staged[min(p, cur_n - 1)];
#include "march_gather.inc"
        }
        // ---- the next chunk is counted by bin as soon as this wave's pieces are here ----
        int key = 0;
        uint32_t rank = 0u;
        if (has_next && mysub == 0 && myrec < next_n) { key = key_of(pc); rank = atomicAdd(&cnt[par * 64 + key], 1u); }
        lds_barrier();                                   // every wave has finished with the stage (and, if the layer ends, with its window)
        if (!has_next) break;
        place_piece(pc, key, rank, next_n, cnt + par * 64);
        if (tid < 64) cnt[(par ^ 1) * 64 + tid] = 0u;
        if (tid == 0) counter[0] = 0;
        if (more) {
            // the BZ new planes take the slots of the BZ oldest: slots pm .. pm + BZ - 1 (mod RZ)
#pragma unroll
            for (int u = 0; u < NPF; ++u) {
                if (pf_el[u] >= 0) {
                    int slot = pm + (pf_el[u] >> 24);
                    if (slot >= RZ) slot -= RZ;
                    ring[slot * PSP + (pf_el[u] & 0xffffff)] = pf[u];
                }
            }
            pm += BZ;
            if (pm >= RZ) pm -= RZ;
        }
        lay = nlay2; ch = nch2; cur_n = next_n; par ^= 1;
        lds_barrier();
    }
}

}  // namespace nufft
