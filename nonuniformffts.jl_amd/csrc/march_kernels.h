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
};

constexpr int kMarchMaxRows = 16;   // rows of bins of a column (n2 <= 64)

template <typename T, bool CPLX, int M, bool POLY = true>
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
    static constexpr int pad_plane(int ps) {            // plane stride in reals: = ROW_BYTES (mod 2 ROW_BYTES) bytes
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
    static constexpr int table_bytes(int segl) { return round_up(kMarchMaxRows * segl * 8 + segl * 4 + 64, 16); }
    static constexpr int kSegMax = 64;
    static constexpr int fixed_bytes_for(int threads) { return table_bytes(kSegMax) + threads / kWave * strip_bytes() + 64; }
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
};

// A task is a column and a segment of its bin layers (at most kSegMax) from set_points' table: segments of about equal
// point count, or of equal length for uniform point sets (balance.hip).
template <typename T, bool CPLX, int M, bool POLY>
__global__ __launch_bounds__((MarchCfg<T, CPLX, M, POLY>::THREADS)) void interp_march_kernel(TileArgs<T> a, MarchGeom mg) {
    using C = MarchCfg<T, CPLX, M, POLY>;
    constexpr int kMarchThreads = C::THREADS;
    using GP = typename C::GP;
    constexpr int NC = C::NC, L = C::L, RZ = C::RZ, BZ = C::BZ, RS = C::RS, PS = C::PS, P1 = C::P1, P2 = C::P2;
    constexpr int N1 = C::N1, N2 = C::N2, NPF = C::NPF, PSP = C::PSP, ZP = C::ZP, PPW = C::PPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    if (*mg.flag == 0u) return;                         // interp_tile_kernel serves this point set
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const Geom& g = a.g;
    const int task = xcd_remap_chunked((int)blockIdx.x, (int)gridDim.x, a.xcd_chunk);
    if (task >= mg.ntasks) return;
    const int comp_id = blockIdx.y;
    const uint2 te = mg.tasktab[task];
    const int tx = (int)te.x % mg.ntx, ty = (int)te.x / mg.ntx;
    const int zb0 = (int)(te.y & 0xffffu), zb1 = (int)(te.y >> 16);
    if (zb1 <= zb0) return;                             // a task that received no layers
    const int org1 = tx * N1, org2 = ty * N2;
    const int neff1 = min(N1, g.Nover[0] - org1), neff2 = min(N2, g.Nover[1] - org2);
    const int nlay = zb1 - zb0;
    const int nrows = (neff2 + 3) >> 2;                 // rows of bins of the column
    const int bx0 = org1 >> 2, nbx = (neff1 + 3) >> 2, by0 = org2 >> 2;

    T* ring = reinterpret_cast<T*>(smem);
    uint2* runs = reinterpret_cast<uint2*>(smem + C::RING_BYTES);                 // [layer][row] -> [p0, p1) of the sorted array
    int* maxp = reinterpret_cast<int*>(runs + kMarchMaxRows * C::kSegMax);        // [layer] -> passes of its longest run
    int* counter = reinterpret_cast<int*>(smem + C::RING_BYTES + C::table_bytes(C::kSegMax) - 64);   // [0] pass counter, [1] any point
    T* strip_wave = reinterpret_cast<T*>(smem + C::RING_BYTES + C::table_bytes(C::kSegMax) + wave * C::strip_bytes());

    // ---- runs of the segment, and whether it holds any point at all ----
    if (tid < 2) counter[tid] = 0;
    for (int i = tid; i < nlay; i += kMarchThreads) maxp[i] = 0;
    __syncthreads();
    {
        int any = 0;
        for (int i = tid; i < nlay * nrows; i += kMarchThreads) {
            const int lay = i / nrows, row = i % nrows;
            const int64_t bin0 = ((int64_t)(zb0 + lay) * g.nb[1] + by0 + row) * g.nb[0] + bx0;
            const uint2 pr = make_uint2(a.offsets[bin0], a.offsets[bin0 + nbx]);
            runs[lay * kMarchMaxRows + row] = pr;
            any |= pr.x != pr.y;
            atomicMax(&maxp[lay], (int)((pr.y - pr.x + PPW - 1) / PPW));
        }
        if (any) counter[1] = 1;
    }
    __syncthreads();
    if (counter[1] == 0) return;

    const T* grid = a.grid[comp_id];
    const int zbase = 4 * zb0 - (M - 1);                // first plane of the segment's first window (may be negative)
    const int o1 = org1 - (M - 1), o2 = org2 - (M - 1);

    // lane roles: G lanes per point, lane q = (j1, component) — or j1 alone with both components per lane (PAIR)
    constexpr bool PAIR = C::PAIR;
    constexpr int NCL = PAIR ? 1 : NC;                  // components that have lanes of their own
    typedef T VT2 __attribute__((ext_vector_type(2)));
    using VT = typename std::conditional<PAIR, VT2, T>::type;      // what one lane gathers per stencil node
    auto vfma = [](VT x, T w, VT acc) __attribute__((always_inline)) -> VT {
        if constexpr (PAIR) return __builtin_elementwise_fma(x, VT{w, w}, acc);
        else return fma(x, w, acc);
    };
    const int grp = lane / (GP::G * ZP), q = lane % GP::G;
    const int zp = ZP > 1 ? (lane / GP::G) % ZP : 0;   // which of the point's rows: stencil planes zp, zp + ZP, ...
    const bool lane_active = q < GP::W1;
    const int comp = q % NCL, j1 = (q / NCL) % L;
    T* strip = strip_wave + grp * (3 * L);
    using WEv = WindowEval<T, NCL, 3, M, GP::G, false>;
    WEv we;
    const EvalArgs<T, POLY ? NUFFT_EVAL_FAST_APPROXIMATION : NUFFT_EVAL_DIRECT> am(a);       // (the instantiation fixes the evaluation mode)
    we.init(am, q);
    if constexpr (ZP == 2) {
        // the second row holds the dimension-3 values with neighbouring pairs exchanged: a row broadcast from lane 2 jj
        // then hands row 0 value 2 jj and row 1 value 2 jj + 1 — each row the weight of its own plane
        if (zp) {
#pragma unroll
            for (int sl = 0; sl < WEv::NSLOT; ++sl)
                if (we.dsel[sl] == 2) {
                    we.jsel[sl] ^= 1;
#pragma unroll
                    for (int c = 0; c < WEv::NP; ++c) we.cs[sl][c] = a.coefs[(2 * WEv::NP + c) * L + we.jsel[sl]];
                }
        }
    }
    const PointRec<T, 3>* sorted = static_cast<const PointRec<T, 3>*>(a.sorted);
    T* vout = a.vout[comp_id];
    __syncthreads();

    // Layer-invariant part of the plane prefetch: element e = tid + u * THREADS of the BZ new planes is real `el` of plane k; its
    // offset inside a plane of the grid (periodic in x, y) and its place in an LDS plane never change from phase to phase — only
    // the plane index does.  (Recomputing them per phase — two divisions and two periodic wraps per element — was 50 of the 184
    // vector instructions per point at C3, where a layer of a 16 x 16 column holds only ~95 points, and 13 of 43 at C2.)
    int pf_off[NPF], pf_el[NPF];                        // grid offset within a plane; k << 24 | el  (-1: no element)
    const int64_t plane_reals = (int64_t)g.Nover[1] * g.Nover[0] * NC;
#pragma unroll
    for (int u = 0; u < NPF; ++u) {
        const int e = tid + u * kMarchThreads;
        pf_off[u] = 0;
        pf_el[u] = -1;
        if (e < BZ * PS) {
            const int k = e / PS, el = e % PS, r = el / RS, xx = el % RS;
            pf_off[u] = (wrap_index(o2 + r, g.Nover[1]) * g.Nover[0] + wrap_index(o1 + xx / NC, g.Nover[0])) * NC + xx % NC;
            pf_el[u] = (k << 24) | el;
        }
    }
    // ---- first window: RZ planes straight into the ring (plane zbase + k in slot k), BZ planes at a time with the same tables ----
    for (int b0 = 0; b0 < RZ; b0 += BZ) {
#pragma unroll
        for (int u = 0; u < NPF; ++u) {
            const int pl = b0 + (pf_el[u] >> 24);
            if (pf_el[u] >= 0 && pl < RZ)
                ring[pl * PSP + (pf_el[u] & 0xffffff)] = grid[(int64_t)wrap_index(zbase + pl, g.Nover[2]) * plane_reals + pf_off[u]];
        }
    }
    __syncthreads();
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
            const uint2 pr = runs[(lay0 + kl) * kMarchMaxRows + rl % nrows];
            const uint32_t p0 = pr.x + (uint32_t)(item / nrl) * PPW, p1 = pr.y;
            if (p0 >= p1) continue;
            const uint32_t p = p0 + grp;
            const bool have = p < p1;
            const PointRec<T, 3> rec = sorted[min(p, p1 - 1)];
            int s[3];
            T X[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int c = cell_of(rec.r[d], g.Nover[d]);
                X[d] = rec.r[d] - T(c);
                s[d] = c;
            }
            s[0] -= org1; s[1] -= org2;                   // first stencil node in padded-column coordinates
            int s3 = pm + 4 * kl + (s[2] & 3) + zp;      // ring slot of this lane's first stencil plane
            if (s3 >= RZ) s3 -= RZ;
            T wv[WEv::NSLOT];
            T w1;
            if constexpr (C::REGW) {
                we.eval_regs(am, X, wv);
                w1 = wv[0];
            } else {
                wave_lds_fence();
                we.eval_to_strip(am, X, strip, q);
                wave_lds_fence();
                w1 = strip[j1];
            }
            auto wfetch = [&](int d, int j) __attribute__((always_inline)) -> T {
                const int kk = d * L + j;
                const T x = wv[kk / GP::G];
                if constexpr (GP::G == 16) return row_bcast(x, kk % GP::G);
                const T lo = row_bcast(x, kk % GP::G), hi = row_bcast(x, kk % GP::G + 8);
                return (lane & 8) ? hi : lo;
            };
            const T* base = ring + (s[0] + j1) * NC + comp + s[1] * RS;
            int poff = s3 * PSP;                         // plane offset of the lane's stencil plane (wraps at RZ * PSP)
            T acc = T(0);
            VT accv = VT(0);
            if constexpr (C::REGW) {
                T w2[L];
#pragma unroll
                for (int j = 0; j < L; ++j) w2[j] = wfetch(1, j);
                // rows per group of the hand-scheduled form (0: compiler-scheduled reads): what compiles without spills
                // (m = 4 at 16 waves: groups of 4 spill 8 registers for Float64, 1.31 against 1.26 ms; Float32 gains 3 %: left at 2)
                // (m = 8 at 16 waves: 8 rows in flight twice over spill 18 registers, groups of 4 fit — C3: 49.8 against 45.0 ms; groups of 2: 47.4)
                constexpr int R = L <= 8 ? ((CPLX || C::THREADS != 512 || L % 4 != 0) ? 2 : 4) : (L % 8 == 0 ? (C::THREADS == 512 ? 8 : 4) : (L % 4 == 0 ? 4 : 0));
                static_assert(R == 0 || L % R == 0, "row groups must tile the stencil");
                if constexpr (R > 0) {
                // hand-scheduled LDS reads (as in interp_tile_kernel): groups of R rows with immediate offsets from the
                // plane's own address (the ring wraps between planes), the next group in flight while this one is consumed
                constexpr int GPP = L / R, NG = (L / ZP) * GPP, RB = RS * (int)sizeof(T);

                const uint32_t a0 = (uint32_t)(uintptr_t)base;
                VT buf[2][R];
                lds_read_rows<VT, R, 0, RB>(buf[0], a0 + (uint32_t)poff * (uint32_t)sizeof(T), std::make_integer_sequence<int, R>{});
                // (ComplexF32: two accumulation chains per plane — a v_pk_fma_f32 that depends on the previous one issues
                // every 6.2 cycles instead of 4.5 even with other waves to fill the gap, scripts/microbench10.hip)
#ifndef NUFFT_MARCH_NT_ALL
#define NUFFT_MARCH_NT_ALL 0
#endif
                constexpr int NT = (PAIR || NUFFT_MARCH_NT_ALL) ? 2 : 1;
                VT t2[NT];
#pragma unroll
                for (int c = 0; c < NT; ++c) t2[c] = VT(0);
#pragma unroll
                for (int gi = 0; gi < NG; ++gi) {
                    if (gi + 1 < NG) {
                        if ((gi + 1) % GPP == 0) {                 // next group starts the lane's next plane
                            poff += ZP * PSP;
                            if (poff >= RZ * PSP) poff -= RZ * PSP;
                        }
                        const uint32_t ad = a0 + (uint32_t)poff * (uint32_t)sizeof(T) + (uint32_t)(((gi + 1) % GPP) * R * RB);
                        lds_read_rows<VT, R, 0, RB>(buf[(gi + 1) & 1], ad, std::make_integer_sequence<int, R>{});
                        lds_wait_rows<R>(buf[gi & 1]);
                    } else {
                        lds_wait_rows<0>(buf[gi & 1]);
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) t2[r % NT] = vfma(buf[gi & 1][r], w2[(gi % GPP) * R + r], t2[r % NT]);
                    if (gi % GPP == GPP - 1) {
                        VT tp = t2[0];
                        if constexpr (NT == 2) tp += t2[1];
                        accv = vfma(tp, wfetch(2, ZP * (gi / GPP)), accv);
#pragma unroll
                        for (int c = 0; c < NT; ++c) t2[c] = VT(0);
                    }
                }
                } else {
                    // wide stencils: the unrolled hand-scheduled form (2M x M groups) spills; the compiler schedules the reads
#pragma unroll
                    for (int j3 = 0; j3 < L; j3 += ZP) {
                        const VT* plane = reinterpret_cast<const VT*>(base + poff);
                        VT t2 = VT(0);
#pragma unroll
                        for (int j2 = 0; j2 < L; ++j2) t2 = vfma(*reinterpret_cast<const VT*>(reinterpret_cast<const T*>(plane) + j2 * RS), w2[j2], t2);
                        accv = vfma(t2, wfetch(2, j3), accv);
                        poff += ZP * PSP;
                        if (poff >= RZ * PSP) poff -= RZ * PSP;
                    }
                }
                if constexpr (PAIR) {
                    const bool okl = have && lane_active;
                    const T re = group_sum<T, GP::G * ZP, false>(okl ? accv[0] * w1 : T(0));
                    const T im = group_sum<T, GP::G * ZP, false>(okl ? accv[1] * w1 : T(0));
                    if (have && q == 0 && zp == 0) *reinterpret_cast<VT*>(vout + (int64_t)rec.idx * 2) = VT{re * a.prefactor, im * a.prefactor};
                    continue;
                } else {
                    acc = accv;
                }
                acc = (have && lane_active) ? acc * w1 : T(0);
            } else if (have && lane_active) {
                T w2[L];
#pragma unroll
                for (int j = 0; j < L; ++j) w2[j] = strip[L + j];
#pragma unroll
                for (int j3 = 0; j3 < L; ++j3) {
                    const T* plane = base + poff;
                    T t2 = T(0);
#pragma unroll
                    for (int j2 = 0; j2 < L; ++j2) t2 = fma(plane[j2 * RS], w2[j2], t2);
                    acc = fma(t2, strip[2 * L + j3], acc);
                    poff += PSP;
                    if (poff >= RZ * PSP) poff -= RZ * PSP;
                }
                acc *= w1;
            }
            acc = group_sum<T, GP::G * ZP, CPLX>(acc);
            if (have && q < NC && zp == 0) vout[(int64_t)rec.idx * NC + q] = acc * a.prefactor;
        }
        __syncthreads();                                 // every wave has finished with this phase's window
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
            __syncthreads();
        }
    }
}

}  // namespace nufft
