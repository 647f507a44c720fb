// Type-1 spreading on a z-marching LDS ring (gfx950, wave64): the third spreading engine.
//
// Replaces spread_from_points_shmem_kernel! (reference src/spreading/gpu.jl:237-377, flush :381-434) and the zero fill in
// front of it (src/NonuniformFFTs.jl:161-167) for 3-D plans, with the same arithmetic per point.  spread_tile_kernel holds
// the interior of a box in LDS and visits every point within M cells of it: 2.08 visits per point at C2 (24 x 28 x 24),
// each with its own window evaluation, scalar clipping control per stencil plane and LDS round trips.  Here a workgroup
// owns a COLUMN of the grid — n1 x n2 cells in x, y, interior only (output-driven as before: no halo in LDS, no global
// atomics, every cell written once) — and marches along z through a segment of bin layers.  LDS holds a ring of
// RZ = 2M - 1 + 4 planes: exactly the planes the stencils of one bin layer (4 planes of cells) can touch.  After a layer
// the 4 oldest planes are complete: they leave with coalesced stores, are zeroed, and become the 4 newest.  Points are
// clipped in x and y only (C2: 40 x 36 column, 1.42 visits per point); along z a stencil always lies inside the ring, so
// its 2M planes are added with immediate offsets from per-slot code (switch on the ring slot of the first plane: no
// per-plane control).  Only the first and last layers of a segment clip along z (planes that belong to the neighbouring
// segments).
//
// Points: the bins of a layer that can touch the column are runs of the bin-sorted array (one per row of bins, two where
// the column sits at the periodic boundary in x); a wave pulls chunks of PPW points from an LDS counter, the next chunk's
// records in flight behind the current one.  Window evaluation (group mapping) and accumulation (face mapping, one point
// per wave instruction, ds_add_f64 on conflict-free rows) are those of spread_tile_kernel.
//
// Tasks: column x segment of bin layers from the table set_points builds per point set on the device (balance.hip):
// equal-length segments for uniform sets, column quantiles otherwise; point sets whose heaviest task would hold the chip
// up — and grids with too few columns — go to spread_tile_kernel, which shares heavy tiles between workgroups (device
// flag, no host read-back).
#pragma once

#include <hip/hip_runtime.h>

#include <utility>

#include "device_common.h"
#include "march_kernels.h"
#include "nufft_mi355x.h"
#include "tile_kernels.h"

namespace nufft {

#ifndef NUFFT_SMARCH_ABL
#define NUFFT_SMARCH_ABL 0          // ablation builds: 1 = no LDS atomics, 2 = no point visits, 3 = no retire stores
#endif

template <typename T, bool CPLX, int M>
struct SMarchCfg {
    static constexpr int NC = CPLX ? 2 : 1;
    static constexpr int L = 2 * M;
    static constexpr int RZ = L + 3;                    // ring depth: the planes one bin layer's stencils can touch
    static constexpr int HLO = (M + 3) / 4;             // layers of points below a segment whose stencils reach into it
    static constexpr int HHI = 1 + (M - 2) / 4;         // ... and above it
    static constexpr int THREADS = 1024;
    static constexpr int NW = THREADS / kWave;
    using GP = Grp<NC, M>;
    static constexpr int FACE = GP::W1 * L;             // (component, j1, j2) elements of a stencil face
    static constexpr int NPASS = (FACE + kWave - 1) / kWave;
    static constexpr int kMaxRuns = 64;                 // runs of the sorted array per layer (rows of bins x 2)
    static constexpr int strip_bytes() { return round_up(GP::PPW * 3 * L * (int)sizeof(T), 16); }
    // per layer, double-buffered: runs (uint2), cumulative chunk counts (uint32); then counters
    static constexpr int table_bytes() { return 2 * (kMaxRuns * 8 + kMaxRuns * 4) + kMaxRuns * 8 + 64; }
    static constexpr int fixed_bytes() { return table_bytes() + NW * strip_bytes(); }
    static constexpr int row_stride(int n1) { return padded_row_stride(NC * n1, NC * L, 8); }
    struct Dims { int n1, n2; };
    static constexpr int bin_rows(int n) { return tile_bin_rows_bound(true, n, 4, M); }
    // column interior (n1, n2): multiples of the bin edge, fewest point visits within the LDS budget
    static constexpr Dims search() {
        Dims best{0, 0};
        double best_cost = 1e300;
        for (int n2 = 4; n2 <= 64; n2 += 4)
            for (int n1 = 4; n1 <= 64; n1 += 4) {
                const long bytes = (long)row_stride(n1) * n2 * RZ * 8 + fixed_bytes();
                if (bytes > 163840 - 256) continue;
                if (2 * bin_rows(n2) > kMaxRuns) continue;
                // visits per point on a 512-cell axis (the partial last column counts)
                auto axis = [](int n) {
                    const int full = 512 / n, rest = 512 - full * n;
                    return (double)(full * (n + L - 1) + (rest ? rest + L - 1 : 0)) / 512.0;
                };
                double cost = axis(n1) * axis(n2);
                cost -= 1e-6 * n1;
                if (cost < best_cost) { best_cost = cost; best = Dims{n1, n2}; }
            }
        return best;
    }
    static constexpr Dims DIMS = search();
    static constexpr int N1 = DIMS.n1, N2 = DIMS.n2;
    static constexpr bool FITS = N1 > 0;
    static constexpr int RS = FITS ? row_stride(N1) : 2;    // row stride in doubles
    static constexpr int PS = RS * (FITS ? N2 : 2);         // plane stride in doubles
    static constexpr int PSB = PS * 8;                      // ... in bytes
    static constexpr int RING_BYTES = round_up(RZ * PSB, 16);
    static constexpr int lds_bytes() { return RING_BYTES + fixed_bytes(); }
    // immediate-offset atomics: KB planes per base address (16-bit offsets), NB bases cover the ring
    static constexpr int KB = 65535 / PSB + 1;
    static constexpr int NB = (RZ + KB - 1) / KB;
    // per-slot code for the 2M planes of a point: RZ cases x NPASS x 2M atomics — only where that stays small
    static constexpr bool SLOTSW = RZ * NPASS * L <= 400 && NB <= 4;
};

// the L planes of a point whose first plane sits in ring slot S: plane J in slot (S + J) % RZ, as an immediate offset
// from the lane's address in the first slot of that slot's base group
template <typename C, int S, typename T, int... J>
__device__ __forceinline__ void smarch_add_planes(const uint32_t (&vb)[C::NB], T w, const T (&w3)[C::L], std::integer_sequence<int, J...>) {
    (lds_add_imm<(((S + J) % C::RZ) % C::KB) * C::PSB>(vb[((S + J) % C::RZ) / C::KB], (double)(w * w3[J])), ...);
}
// binary dispatch on the (wave-uniform) slot
template <typename C, int LO, int HI, typename T>
__device__ __forceinline__ void smarch_slot_dispatch(int slot, const uint32_t (&vb)[C::NB], T w, const T (&w3)[C::L]) {
    if constexpr (HI - LO == 1) {
        smarch_add_planes<C, LO>(vb, w, w3, std::make_integer_sequence<int, C::L>{});
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (slot < MID) smarch_slot_dispatch<C, LO, MID>(slot, vb, w, w3);
        else smarch_slot_dispatch<C, MID, HI>(slot, vb, w, w3);
    }
}

template <typename T, bool CPLX, int M>
__global__ __launch_bounds__(1024) void spread_march_kernel(TileArgs<T> a, MarchGeom mg) {
    using C = SMarchCfg<T, CPLX, M>;
    using GP = typename C::GP;
    constexpr int NC = C::NC, L = C::L, RZ = C::RZ, RS = C::RS, PS = C::PS, PSB = C::PSB, N1 = C::N1, N2 = C::N2;
    constexpr int NPASS = C::NPASS, FACE = C::FACE, PPW = GP::PPW, HLO = C::HLO, HHI = C::HHI, THREADS = C::THREADS;
    constexpr int kMaxRuns = C::kMaxRuns;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    if (*mg.flag == 0u) return;                         // spread_tile_kernel serves this point set
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const Geom& g = a.g;
    const int task = xcd_remap_chunked((int)blockIdx.x, (int)gridDim.x, a.xcd_chunk);
    if (task >= mg.ntasks) return;
    const int comp_id = blockIdx.y;
    const uint2 te = mg.tasktab[task];
    const int tx = (int)te.x % mg.ntx, ty = (int)te.x / mg.ntx;
    const int zb0 = (int)(te.y & 0xffffu), zb1 = (int)(te.y >> 16);
    if (zb1 <= zb0) return;                             // a task that received no layers
    // the column of this grid: mg.n1 x mg.n2 <= N1 x N2 cells (plan creation picks what fills the chip best)
    const int org1 = tx * mg.n1, org2 = ty * mg.n2;
    const int neff1 = min(mg.n1, g.Nover[0] - org1), neff2 = min(mg.n2, g.Nover[1] - org2);
    const int nlay = zb1 - zb0;
    const int nq = 4 * nlay;                            // planes this task owns: q = 0 .. nq - 1 (plane 4 zb0 + q of the grid)
    const int nli = nlay + HLO + HHI;                   // layers of points it visits

    double* ring = reinterpret_cast<double*>(smem);
    const uint32_t ring_base = (uint32_t)(uintptr_t)ring;
    uint2* runs_tab = reinterpret_cast<uint2*>(smem + C::RING_BYTES);                               // [2][kMaxRuns]
    uint32_t* cum_tab = reinterpret_cast<uint32_t*>(smem + C::RING_BYTES + 2 * kMaxRuns * 8);        // [2][kMaxRuns] inclusive chunk counts
    uint2* run_bins = reinterpret_cast<uint2*>(smem + C::RING_BYTES + 2 * (kMaxRuns * 8 + kMaxRuns * 4));   // [kMaxRuns] {first bin within a layer, bins}
    int* counter = reinterpret_cast<int*>(smem + C::RING_BYTES + 2 * (kMaxRuns * 8 + kMaxRuns * 4) + kMaxRuns * 8); // [0] chunk counter, [2], [3]: chunks of the layer per buffer
    T* strip_wave = reinterpret_cast<T*>(smem + C::RING_BYTES + C::table_bytes() + wave * C::strip_bytes());

    // ---- bins whose points can touch the column: rows of bins along y, one or two runs along x.  Their first bin (within
    //      a layer of bins) and length go to an LDS table once; wave 0 turns them into runs of the sorted array per layer ----
    int nruns;
    {
        const BinSegs seg0 = bin_segments(org1 - M, org1 + neff1 + M - 1, g.Nover[0], 2, g.nb[0]);
        const BinSegs seg1 = bin_segments(org2 - M, org2 + neff2 + M - 1, g.Nover[1], 2, g.nb[1]);
        nruns = seg1.total() * seg0.n;                  // <= kMaxRuns (SMarchCfg::search)
        if (tid < kMaxRuns) {
            uint2 d = make_uint2(0u, 0u);
            if (tid < nruns) {
                const int sg = tid % seg0.n, r2 = tid / seg0.n;
                d = make_uint2((uint32_t)(seg1.bin(r2) * g.nb[0] + (sg ? seg0.lo[1] : seg0.lo[0])), (uint32_t)(sg ? seg0.len[1] : seg0.len[0]));
            }
            run_bins[tid] = d;
        }
    }
    // table of layer `li` into buffer `buf` (wave 0, all 64 lanes): runs, inclusive chunk counts, their total
    auto write_table = [&](int li, int buf) __attribute__((always_inline)) {
        int lay = zb0 - HLO + li;
        if (lay < 0) lay += g.nb[2];
        if (lay >= g.nb[2]) lay -= g.nb[2];
        const uint2 d = run_bins[lane];
        const int64_t bin0 = (int64_t)lay * g.nb[1] * g.nb[0] + d.x;
        const uint2 pr = make_uint2(a.offsets[bin0], a.offsets[bin0 + d.y]);      // (lanes >= nruns: an empty run)
        const uint32_t ch = (pr.y - pr.x + PPW - 1) / PPW;
        uint32_t incl = ch;
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += v;
        }
        runs_tab[buf * kMaxRuns + lane] = pr;
        cum_tab[buf * kMaxRuns + lane] = incl;
        if (lane == kWave - 1) counter[2 + buf] = (int)incl;
    };

    // ---- zero the ring, table of the first layer ----
    {
        typedef double D2 __attribute__((ext_vector_type(2)));
        D2* r2p = reinterpret_cast<D2*>(ring);
        for (int i = tid; i < RZ * PS / 2; i += THREADS) r2p[i] = D2{0.0, 0.0};
    }
    __syncthreads();
    if (wave == 0) {
        write_table(0, 0);
        if (lane == 0) counter[0] = 0;
    }

    // evaluation roles
    const int grp = lane / GP::G, q = lane % GP::G;
    T* strip = strip_wave + grp * (3 * L);
    WindowEval<T, NC, 3, M, GP::G, false> we;
    we.init(a, q);
    // accumulation roles
    int j1f[NPASS], j2f[NPASS], cmpf[NPASS];
    bool actf[NPASS];
    uint32_t lane_addr[NPASS];                          // LDS byte address of the lane's face element for a stencil that starts at (0, 0), slot 0
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int e = lane + ps * kWave;
        actf[ps] = e < FACE;
        const int e1 = e % GP::W1;
        j2f[ps] = (e / GP::W1) % L;
        cmpf[ps] = e1 % NC;
        j1f[ps] = e1 / NC;
        lane_addr[ps] = ring_base + (uint32_t)((j1f[ps] * NC + cmpf[ps] + j2f[ps] * RS) * 8);
    }
    const PointRec<T, 3>* sorted = static_cast<const PointRec<T, 3>*>(a.sorted);
    const T* vin = a.vin[comp_id];
    T* grid = a.grid[comp_id];
    __syncthreads();

    int pm = 0;                                         // ring slot of the first plane of the current layer's window
    for (int li = 0; li < nli; ++li) {
        const int buf = li & 1;
        // wave 0: the runs of the next layer (its two loads cost one wave a microsecond per layer)
        if (wave == 0 && li + 1 < nli) write_table(li + 1, buf ^ 1);

        const int nchunks = counter[2 + buf];
        const int wq = 4 * (li - HLO) - (M - 1);        // first plane of the layer's window (in owned-plane coordinates)
        const bool clipz = wq < 0 || wq + RZ > nq;      // planes of this window belong to other segments

        auto lookup = [&](int item, uint32_t& p0, uint32_t& p1) __attribute__((always_inline)) {
            const uint32_t cum_l = cum_tab[buf * kMaxRuns + lane];  // inclusive chunk counts (lanes >= nruns: the total)
            const unsigned long long mk = __ballot((uint32_t)item < cum_l);
            const int i = (int)__builtin_ctzll(mk);
            const uint32_t before = i ? (uint32_t)__builtin_amdgcn_readlane((int)cum_l, i - 1) : 0u;
            const uint2 run = runs_tab[buf * kMaxRuns + i];
            p0 = run.x + ((uint32_t)item - before) * PPW;
            p1 = run.y;
        };
        auto pull = [&]() __attribute__((always_inline)) -> int {
            int item = 0;
            if (lane == 0) item = atomicAdd(counter, 1);
            return __builtin_amdgcn_readfirstlane(item);
        };

        // one chunk of up to PPW points
        auto chunk = [&](auto clip_c, const PointRec<T, 3>& rec, T vmine, bool have) __attribute__((always_inline)) {
            constexpr bool CLIPZ = decltype(clip_c)::value;
            int s[2];
            T X[3];
            bool ok = have;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int c = cell_of(rec.r[d], g.Nover[d]);
                X[d] = rec.r[d] - T(c);
                int sd = c - (M - 1) - (d == 0 ? org1 : org2);      // first stencil node in column coordinates
                const int ne = d == 0 ? neff1 : neff2;
                if (sd > ne - 1) sd -= g.Nover[d];                  // periodic image next to this column
                if (sd < -(L - 1)) sd += g.Nover[d];
                ok = ok && (sd >= -(L - 1)) && (sd <= ne - 1);
                s[d] = sd;
            }
            const int c3 = cell_of(rec.r[2], g.Nover[2]);
            X[2] = rec.r[2] - T(c3);
            const int dz = c3 & 3;
            int slot3 = pm + dz;                        // ring slot of the first stencil plane
            if (slot3 >= RZ) slot3 -= RZ;
            const int q0 = wq + dz;                     // its plane (owned-plane coordinates)
            if constexpr (CLIPZ) ok = ok && (q0 + L - 1 >= 0) && (q0 < nq);
            const unsigned long long okmask = __ballot(ok);
            if (okmask == 0ull) return;                 // nothing of this chunk touches the column
            wave_lds_fence();
            we.template eval_to_strip<0>(a, X, strip, q);
            wave_lds_fence();
#if NUFFT_SMARCH_ABL != 2
            auto do_point = [&](int gi, const T (&w1v)[NPASS], const T (&w2v)[NPASS], T w3a) __attribute__((always_inline)) {
                const int src = gi * GP::G;             // first lane of the point's group
                if (!((okmask >> src) & 1ull)) return;
                const int S1 = __builtin_amdgcn_readlane(s[0], src);
                const int S2 = __builtin_amdgcn_readlane(s[1], src);
                const int SL = __builtin_amdgcn_readlane(slot3, src);
                const T Vre = readlane_t(vmine, src);
                const T Vim = CPLX ? readlane_t(vmine, src + (CPLX ? 1 : 0)) : T(0);
                const T* sp = strip_wave + gi * (3 * L);
                T w3[L];
                {
                    T w3b = T(0);
                    if constexpr (L > 16) w3b = sp[2 * L + min(16 + (lane & 15), L - 1)];
                    if constexpr (L <= 16) {
                        row_bcast_all(w3a, w3, 0, std::make_integer_sequence<int, L>{});
                    } else {
#pragma unroll
                        for (int j = 0; j < L; ++j) w3[j] = j < 16 ? row_bcast(w3a, j) : row_bcast(w3b, j - 16);
                    }
                }
                const uint32_t soff = (uint32_t)((S1 * NC + S2 * RS) * 8);
                unsigned planes = (1u << L) - 1u;
                if constexpr (CLIPZ) {
                    const int Q0 = __builtin_amdgcn_readlane(q0, src);
                    const int lo3 = max(0, -Q0), hi3 = min(L, nq - Q0);
                    planes = ((1u << hi3) - 1u) & ~((1u << lo3) - 1u);
                }
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    const int l1 = S1 + j1f[ps], l2 = S2 + j2f[ps];
                    const bool lane_ok = actf[ps] && (unsigned)l1 < (unsigned)neff1 && (unsigned)l2 < (unsigned)neff2;
                    const T w = w1v[ps] * (CPLX ? (cmpf[ps] ? Vim : Vre) : Vre) * w2v[ps];
                    const uint32_t v0 = lane_addr[ps] + soff;       // the lane's element in slot 0
                    if (lane_ok) {
                        if constexpr (!CLIPZ && C::SLOTSW) {
                            uint32_t vb[C::NB];
#pragma unroll
                            for (int b = 0; b < C::NB; ++b) vb[b] = v0 + (uint32_t)(b * C::KB * PSB);
                            smarch_slot_dispatch<C, 0, RZ>(SL, vb, w, w3);
                        } else {
                            int slot = SL;
#pragma unroll
                            for (int j3 = 0; j3 < L; ++j3) {
                                if (!CLIPZ || (planes & (1u << j3))) {
                                    lds_atomic_add(ring + ((v0 - ring_base) >> 3) + slot * PS, (double)(w * w3[j3]));
                                }
                                slot = slot + 1 == RZ ? 0 : slot + 1;
                            }
                        }
                    }
                }
            };
            if constexpr (NUFFT_SPREAD_ASM_STRIP && NPASS == 1 && L <= 16 && PPW % 2 == 0) {
                // the three strip reads of a point issued together for two points at a time (see spread_tile_kernel)
                const uint32_t sb1 = (uint32_t)(uintptr_t)(strip_wave + j1f[0]);
                const uint32_t sb2 = (uint32_t)(uintptr_t)(strip_wave + L + j2f[0]);
                const uint32_t sb3 = (uint32_t)(uintptr_t)(strip_wave + 2 * L + min(lane & 15, L - 1));
                constexpr int PB = 3 * L * (int)sizeof(T);
                for_each_pair<0, PPW>([&](auto G0c) {
                    constexpr int g0 = decltype(G0c)::value;
                    T pre[6];
                    lds_read_imm<T, (g0 + 0) * PB>(pre[0], sb1);
                    lds_read_imm<T, (g0 + 0) * PB>(pre[1], sb2);
                    lds_read_imm<T, (g0 + 0) * PB>(pre[2], sb3);
                    lds_read_imm<T, (g0 + 1) * PB>(pre[3], sb1);
                    lds_read_imm<T, (g0 + 1) * PB>(pre[4], sb2);
                    lds_read_imm<T, (g0 + 1) * PB>(pre[5], sb3);
                    lds_wait_rows<0>(pre);
                    { const T a1[1] = {pre[0]}, a2[1] = {pre[1]}; do_point(g0, a1, a2, pre[2]); }
                    { const T a1[1] = {pre[3]}, a2[1] = {pre[4]}; do_point(g0 + 1, a1, a2, pre[5]); }
                });
            } else {
#pragma unroll
                for (int gi = 0; gi < PPW; ++gi) {
                    const T* sp = strip_wave + gi * (3 * L);
                    T w1v[NPASS], w2v[NPASS];
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) {
                        w1v[ps] = sp[j1f[ps]];
                        w2v[ps] = sp[L + j2f[ps]];
                    }
                    const T w3a = sp[2 * L + min(lane & 15, L - 1)];
                    do_point(gi, w1v, w2v, w3a);
                }
            }
#else
            asm volatile("" ::"v"(s[0]), "v"(s[1]), "v"(slot3), "v"(vmine));
#endif
        };

        // ---- chunks of this layer, pulled from the counter; the next chunk's records are requested before the current one
        //      is processed, its values right after ----
        {
            int item = pull();
            uint32_t p0 = 0, p1 = 0;
            bool valid = item < nchunks;
            PointRec<T, 3> rec{};
            T vcur = T(0);
            if (valid) {
                lookup(item, p0, p1);
                rec = sorted[min(p0 + (uint32_t)grp, p1 - 1)];
                if (q < NC) {
                    vcur = vin[(int64_t)rec.idx * NC + q];
                    if (a.weights) vcur *= a.weights[rec.idx];      // callbacks.nonuniform(v, n), src/spreading/gpu.jl:289
                }
            }
            while (valid) {
                const int itn = pull();
                const bool validn = itn < nchunks;
                uint32_t n0 = 0, n1 = 0;
                PointRec<T, 3> recn = rec;
                if (validn) {
                    lookup(itn, n0, n1);
                    recn = sorted[min(n0 + (uint32_t)grp, n1 - 1)];
                }
                const bool have = p0 + (uint32_t)grp < p1;
                if (clipz) chunk(std::true_type{}, rec, vcur, have);
                else chunk(std::false_type{}, rec, vcur, have);
                if (validn && q < NC) {
                    vcur = vin[(int64_t)recn.idx * NC + q];
                    if (a.weights) vcur *= a.weights[recn.idx];
                }
                rec = recn; p0 = n0; p1 = n1; valid = validn;
            }
        }
        // the immediate-offset atomics are inline assembly: the compiler does not know that they are in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- retire: the 4 oldest planes of the window are complete; store the owned ones, zero them, recycle ----
        {
            typedef double D2 __attribute__((ext_vector_type(2)));
            typedef T T2 __attribute__((ext_vector_type(2)));
            constexpr int RP = NC * N1 / 2;             // pairs per row
            constexpr int NPAIR = N2 * RP;
            if (tid == 0) counter[0] = 0;
            if (wq + 4 > 0 && wq < nq) {
                for (int e = tid; e < 4 * NPAIR; e += THREADS) {
                    const int pl = e / NPAIR, er = e % NPAIR, r = er / RP, xp = er % RP;
                    const int qq = wq + pl;
                    if (qq < 0 || qq >= nq || r >= neff2 || 2 * xp >= NC * neff1) continue;
                    int slot = pm + pl;
                    if (slot >= RZ) slot -= RZ;
                    D2* src = reinterpret_cast<D2*>(ring + slot * PS + r * RS + 2 * xp);
                    const D2 v = *src;
                    *src = D2{0.0, 0.0};
#if NUFFT_SMARCH_ABL != 3
                    const int gz = 4 * zb0 + qq;        // < Nover[2]: the task owns these planes
                    const int64_t row = (int64_t)gz * g.Nover[1] + org2 + r;
                    *reinterpret_cast<T2*>(grid + (row * g.Nover[0] + org1) * NC + 2 * xp) = T2{(T)v.x, (T)v.y};
#else
                    asm volatile("" ::"v"(v));
#endif
                }
            }
        }
        __syncthreads();
        pm += 4;
        if (pm >= RZ) pm -= RZ;
    }
}

}  // namespace nufft
