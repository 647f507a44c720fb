// Type-1 spreading on a z-marching LDS window (gfx950, wave64): the third spreading engine.
//
// Replaces spread_from_points_shmem_kernel! (reference src/spreading/gpu.jl:237-377, flush :381-434) and the zero fill in
// front of it (src/NonuniformFFTs.jl:161-167) for 3-D plans, with the same arithmetic per point.  spread_tile_kernel holds
// the interior of a box in LDS and visits every point within M cells of it: 2.08 visits per point at C2 (24 x 28 x 24),
// each with its own window evaluation, scalar clipping control per stencil plane and LDS round trips.  Here a workgroup
// owns a COLUMN of the grid — n1 x n2 cells in x, y, interior only (output-driven as before: no halo in LDS, no global
// atomics, every cell written once) — and marches along z through a segment of bin layers.  LDS holds a window of
// RZ = 2M - 1 + 4 planes: exactly the planes the stencils of one bin layer (4 planes of cells) can touch, plane k of the
// window in slot k.  After a layer the 4 lowest planes are complete: they leave with coalesced stores, the other 2M - 1
// move down four slots and the top four are zeroed (one pass of the workgroup over its LDS, ~3 % of a layer's time).
// Because the window never rotates, the slot of a stencil's first plane is just (cell_z mod 4): it is folded into the
// point's LDS offset, and the 2M planes of a point are added with immediate offsets from that one address — no per-plane
// or per-slot control flow at all.  Points are clipped in x and y only (C2: 32 x 32 column, 1.49 visits per point); only
// the first and last layers of a segment clip along z (planes that belong to the neighbouring segments).
//
// Points: the bins of a layer that can touch the column are runs of the bin-sorted array (one per row of bins, two where
// the column sits at the periodic boundary in x).  Every wave keeps the runs of the current layer in registers (one run
// per lane, the next layer's bounds in flight) and takes the chunks wave, wave + 16, ... of the layer: no shared counter,
// no LDS table; the next chunk's records are requested before the current one is processed.  Window evaluation (group
// mapping) and accumulation (face mapping, one point per wave instruction, ds_add_f64 on conflict-free rows) are those
// of spread_tile_kernel.  For real data at M = 4 (the 8 x 8 face is exactly one wave) the point's value is folded into its
// dimension-1 window values, and its clipping mask — an EXEC mask, computed for the eight points of a chunk at once by
// their group lanes — and LDS offset reach the accumulation as three v_readlane: 22 vector and ~8 scalar instructions
// per point visit besides the eight atomics.
//
// Halo variant (template flags HX = HY = true; round 4c).  The column above is OUTPUT-driven in x and y: it holds its own cells only
// and visits every point whose stencil reaches them (1.49 visits per point at 32 x 32).  The halo variant's window also holds the
// stencil's reach beyond the column (free in x: the bank-aligned row stride of a 32-cell column is 40 cells = 32 + 2M; 2M - 1 rows
// in y) and the column visits only ITS OWN points: every point is spread exactly once, unclipped, every atomic with all its lanes.
// The cells of the reach belong to the neighbouring columns: they leave, with plain coalesced stores like the column's own cells,
// into a SIDE BUFFER (one record per column and plane: n2 strips of the x reach, then the 2M - 1 rows of the y reach), and the
// consumer of the grid adds them: the dimension-1 FFT pass of real plans while it loads a line (fft_lines.hip, real_lines_kernel:
// + 0.5 G of reads), or smarch_halo_add_kernel for the stage-level entry point and the plans whose first pass is not that kernel.
// (Round 4b accumulated the reach with global float atomics onto zeroed bands of the grid instead: 1.2e8 atomics per launch at
// 66 G/s — 3.71 + 0.31 ms against 2.44 ms; with plain stores the same kernel took 1.94 ms.)
//
// Complex data through the real kernel (MarchGeom::parts = 2; round 5).  The complex instantiations accumulate (component, j1, j2)
// faces: 128 lanes = two wave instructions per plane at M = 4, and the interleaved window halves the column (ComplexF64 m = 4: 5.06 ms
// against 1.87 ms for real data).  Real and imaginary parts are independent real transforms of the same points, so a complex plan
// can run the REAL kernel twice per component — two workgroups per task, side by side on one XCD — reading the part's half of every value (stride 2
// reals) and storing the part's half of every cell of the interleaved grid (two 8-byte stores instead of one 16-byte store per
// pair); the side buffer of the halo variant stays planar, one per (component, part), and its consumers add it to the real or
// imaginary parts of the lines (fft_lines.hip).  The 8 x 8-face FAST path, the 32 x 32 column and the halo variant of real data carry over.
//
// Tasks: column x segment of bin layers from the table set_points builds per point set on the device (balance.hip):
// equal-length segments for uniform sets, column quantiles otherwise; point sets whose heaviest task would hold the chip
// up — and grids with too few columns — go to spread_tile_kernel, which shares heavy tiles between workgroups (device
// flag, no host read-back).
#pragma once

#include <hip/hip_runtime.h>

#include <utility>

#include "column_tasks.h"
#include "device_common.h"
#include "march_kernels.h"
#include "nufft_mi355x.h"
#include "tile_kernels.h"

namespace nufft {

#ifndef NUFFT_SMARCH_SPLIT_TAIL
#define NUFFT_SMARCH_SPLIT_TAIL 0   // (see the main loop)
#endif
#ifndef NUFFT_SMARCH_PRIO
#define NUFFT_SMARCH_PRIO 3         // wave priority during the accumulation of a chunk (s_setprio; 0: none).  Round 6: the waves that feed the LDS atomic pipe go ahead of
                                    // those that evaluate windows — C2 spread 1.97 -> 1.90 ms Direct(), 1.90 -> 1.87 polynomial, C4 5.84 -> 5.55, ComplexF64 m = 4 3.96 -> 3.79, the
                                    // reference protocol's folded-normal sets 3.66 -> 3.34; levels 1 / 2 / 3 within 1 % of each other (scripts/r6_ae.sh)
#endif
#ifndef NUFFT_SMARCH_AHEAD
#define NUFFT_SMARCH_AHEAD 0        // 1: a wave prepares its first chunk of the next layer before the retire pass (round 6)
#endif
#ifndef NUFFT_SMARCH_DEFER
#define NUFFT_SMARCH_DEFER 0        // 1: the retire pass's global stores issued behind its second barrier (experiment, round 6)
#endif
#ifndef NUFFT_SMARCH_ABL
#define NUFFT_SMARCH_ABL 0          // ablation builds: 1 = no LDS atomics, 2 = no point visits, 3 = no retire stores, 4 = no shift
#endif

template <typename T, bool CPLX, int M, bool HX = false, bool HY = false>
struct SMarchCfg {
    static constexpr int NC = CPLX ? 2 : 1;
    static constexpr int L = 2 * M;
    static constexpr int RZ = L + 3;                    // window depth: the planes one bin layer's stencils can touch
    static constexpr int HLO = (M + 3) / 4;             // layers of points below a segment whose stencils reach into it
    static constexpr int HHI = 1 + (M - 2) / 4;         // ... and above it
    static constexpr int THREADS = 1024;
    static constexpr int NW = THREADS / kWave;
    using GP = Grp<NC, M>;
    static constexpr int FACE = GP::W1 * L;             // (component, j1, j2) elements of a stencil face
    static constexpr int NPASS = (FACE + kWave - 1) / kWave;
    static constexpr int kMaxRuns = 64;                 // runs of the sorted array per layer (rows of bins x 2): one per lane
    // real data, M = 4: the face is one wave (8 x 8 lanes); masks and offsets come from the group lanes
    static constexpr bool FAST = !CPLX && M == 4;
    static constexpr int strip_bytes() { return round_up(GP::PPW * 3 * L * (int)sizeof(T), 16); }
    static constexpr int fixed_bytes() { return NW * strip_bytes() + 64; }
    // window = column + the stencil's reach where the dimension is input-driven: XLO cells below (M - 1 rounded up to even, so
    // that pairs of cells stay 16-byte aligned in the grid) and M above in x; M - 1 rows below and M above in y
    static constexpr int XLO = HX ? ((M - 1) + ((M - 1) & 1)) : 0, XHI = HX ? M : 0;
    static constexpr int YLO = HY ? (M - 1) : 0, YHI = HY ? M : 0;
    static constexpr int row_stride(int n1) { return (padded_row_stride(NC * (n1 + XLO + XHI), NC * L, 8) + 1) & ~1; }   // even: rows of aligned pairs
    struct Dims { int n1, n2; };
    static constexpr int bin_rows(int n) { return tile_bin_rows_bound(true, n, 4, M); }
    // column interior (n1, n2): multiples of the bin edge, fewest point visits within the LDS budget
    static constexpr Dims search() {
        Dims best{0, 0};
        double best_cost = 1e300;
        for (int n2 = 4; n2 <= 64; n2 += 4)
            for (int n1 = 4; n1 <= 64; n1 += 4) {
                const long bytes = (long)row_stride(n1) * (n2 + YLO + YHI) * RZ * 8 + fixed_bytes();
                if (bytes > 163840 - 256) continue;
                if (2 * bin_rows(n2) > kMaxRuns) continue;
                // visits per point on a 512-cell axis (the partial last column counts); input-driven dimensions: 1, and the
                // share of cells that leave with atomics instead (weighted as a fifth of a visit)
                auto axis = [](int n, bool halo) {
                    const int full = 512 / n, rest = 512 - full * n;
                    if (halo) return 1.0 + 0.2 * (double)((full + (rest ? 1 : 0)) * (L - 1)) / 512.0;
                    return (double)(full * (n + L - 1) + (rest ? rest + L - 1 : 0)) / 512.0;
                };
                double cost = axis(n1, HX) * axis(n2, HY);
                cost -= 1e-6 * n1;
                if (cost < best_cost) { best_cost = cost; best = Dims{n1, n2}; }
            }
        return best;
    }
    static constexpr Dims DIMS = search();
    static constexpr int N1 = DIMS.n1, N2 = DIMS.n2;
    static constexpr bool FITS = N1 > 0;
    static constexpr int RS = FITS ? row_stride(N1) : 2;    // row stride in doubles
    static constexpr int WY = (FITS ? N2 : 2) + YLO + YHI;  // rows of the window
    static constexpr int PS = RS * WY;                      // plane stride in doubles
    static constexpr int PSB = PS * 8;                      // ... in bytes
    static constexpr int RING_BYTES = round_up(RZ * PSB, 16);
    static constexpr int lds_bytes() { return RING_BYTES + fixed_bytes(); }
    // immediate-offset atomics: PLB planes per base address (16-bit offsets), NBASE bases cover a stencil's 2M planes
    static constexpr int PLB0 = 65535 / PSB + 1;
    static constexpr int PLB = PLB0 < 4 ? PLB0 : PLB0 / 4 * 4;
    static constexpr int NBASE = (L + PLB - 1) / PLB;
};

// the L planes of a point from the lane's address in the point's first slot: plane J at J * PSB — as an immediate offset
// from the base of its group of PLB planes.  CLIPZ: only the planes of the bit mask (first / last layers of a segment).
template <typename C, bool CLIPZ, int J, typename T>
__device__ __forceinline__ void smarch_add_plane(const uint32_t (&vb)[C::NBASE], T w, const T (&w3)[C::L], unsigned planes) {
#if NUFFT_SMARCH_ABL == 1
    const double v = (double)(w * w3[J]);
    asm volatile("" ::"v"(vb[J / C::PLB]), "v"(v));
#else
    if (!CLIPZ || (planes & (1u << J))) lds_add_imm<(J % C::PLB) * C::PSB>(vb[J / C::PLB], (double)(w * w3[J]));
#endif
}
template <typename C, bool CLIPZ, typename T, int... J>
__device__ __forceinline__ void smarch_add_planes(const uint32_t (&vb)[C::NBASE], T w, const T (&w3)[C::L], unsigned planes,
                                                  std::integer_sequence<int, J...>) {
    (smarch_add_plane<C, CLIPZ, J>(vb, w, w3, planes), ...);
}

// FAST path (2M = 8 planes, all from two base addresses): planes J0 .. J0 + 3 with their window values in w3q
template <typename C, bool CLIPZ, int J0, typename T>
__device__ __forceinline__ void smarch_add_four(const uint32_t (&vb)[C::NBASE], T w, const T (&w3q)[4], unsigned planes) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int J = J0 + k;
#if NUFFT_SMARCH_ABL == 1
        const double v = (double)(w * w3q[k]);
        asm volatile("" ::"v"(vb[J / C::PLB]), "v"(v));
#else
        if (!CLIPZ || (planes & (1u << J))) {
            const uint32_t addr = vb[J / C::PLB];
            const double v = (double)(w * w3q[k]);
            switch (J % C::PLB) {                         // (J is a constant after unrolling)
                case 0: lds_add_imm<0 * C::PSB>(addr, v); break;
                case 1: lds_add_imm<1 * C::PSB>(addr, v); break;
                case 2: lds_add_imm<2 * C::PSB>(addr, v); break;
                default: lds_add_imm<3 * C::PSB>(addr, v); break;
            }
        }
#endif
    }
}

// grid += side buffer (the general consumer: one thread per real of the grid that receives anything gathers from the up to eight
// neighbouring columns).
template <typename T>
__global__ __launch_bounds__(256) void smarch_halo_add_kernel(T* grid, const T* halo, int64_t grid_comp, int64_t halo_comp, Geom g, HaloLayout h,
                                                             const uint32_t* flag) {
    if (*flag == 0u) return;
    const int nc = h.nc, row_reals = g.Nover[0] * nc;
    const int64_t rows = (int64_t)g.Nover[1] * g.Nover[2];
    T* gr = grid + (int64_t)blockIdx.y * grid_comp;
    const T* hb = halo + (int64_t)blockIdx.y * halo_comp;
    for (int64_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const int y = (int)(row % g.Nover[1]), z = (int)(row / g.Nover[1]);
        const int ty = y / h.n2, ly = y - ty * h.n2;
        const T* hz = hb + (int64_t)z * h.plane;
        // cells that receive anything: in a row within the y reach of a column boundary all of them, else only the xhi first and
        // xlo last cells of every column (41 % of the grid at 32 x 32, m = 4)
        const bool yaff = ly < h.yhi || ly >= h.n2 - h.ylo;
        const int per = yaff ? h.n1 : h.xlo + h.xhi;
        for (int t = threadIdx.x; t < h.ntx * per * nc; t += blockDim.x) {
            const int comp = t % nc, c = t / nc;
            const int tx = c / per, k = c - tx * per;
            const int lx = yaff ? k : (k < h.xhi ? k : h.n1 - h.xlo + (k - h.xhi));
            const int xr = (tx * h.n1 + lx) * nc + comp;
            T sum = T(0);
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int ry = ly - dy * h.n2;              // row relative to the source column's first row
                if (ry < -h.ylo || ry >= h.n2 + h.yhi) continue;
                int sy = ty + dy;
                if (sy < 0) sy += h.nty;
                if (sy >= h.nty) sy -= h.nty;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    if (dx == 0 && dy == 0) continue;
                    const int rx = lx - dx * h.n1;
                    if (rx < -h.xlo || rx >= h.n1 + h.xhi) continue;
                    int sx = tx + dx;
                    if (sx < 0) sx += h.ntx;
                    if (sx >= h.ntx) sx -= h.ntx;
                    sum += hz[((int64_t)sy * h.ntx + sx) * h.rec + halo_record_offset(h, (h.xlo + rx) * nc + comp, ry)];
                }
            }
            if (sum != T(0)) gr[row * row_reals + xr] += sum;
        }
    }
}

template <typename T, bool CPLX, int M, bool POLY, bool HX = false, bool HY = false>
__global__ __launch_bounds__(1024) void spread_march_kernel(TileArgs<T> a, MarchGeom mg) {
    using C = SMarchCfg<T, CPLX, M, HX, HY>;
    using GP = typename C::GP;
    using WE = WindowEval<T, C::NC, 3, M, GP::G, false>;
    constexpr int NC = C::NC, L = C::L, RZ = C::RZ, RS = C::RS, PS = C::PS, PSB = C::PSB;
    constexpr int NPASS = C::NPASS, FACE = C::FACE, PPW = GP::PPW, HLO = C::HLO, HHI = C::HHI, THREADS = C::THREADS, NW = C::NW;
    constexpr bool FAST = C::FAST;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    if ((HX || HY) && mg.halo_state && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *mg.halo_state = *mg.flag != 0u ? 1u : 0u;      // the side buffer this launch writes is pending (no side buffer: the tiles serve the set)
    if (*mg.flag == 0u) return;                         // spread_tile_kernel serves this point set
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid / kWave);      // (wave: a scalar)
    const Geom& g = a.g;
    // (complex data through this real kernel, see the header: the two parts of a task are workgroups 8 apart of a launch of 2 x tasks (rounded
    // up to 8) — dispatched back to back on the SAME XCD, so that the halves they write into every 16-byte cell meet in that XCD's L2)
    const bool split = !CPLX && mg.parts == 2;
    const int comp_id = (int)blockIdx.y;
    const int slot = (int)blockIdx.x >> 3;
    const int part = split ? slot & 1 : 0;
    const int vgs = split ? 2 : 1;                      // reals between consecutive values / cells of this part
    const int yrow = split ? 2 * comp_id + part : comp_id;      // planar side buffer of this (component, part)
    const int vblock = split ? (slot >> 1) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
    const int task = xcd_remap_chunked(vblock, split ? (int)gridDim.x >> 1 : (int)gridDim.x, a.xcd_chunk);
    if (task >= mg.ntasks) return;
    const uint2 te = mg.tasktab[task];
    const int tx = (int)te.x % mg.ntx, ty = (int)te.x / mg.ntx;
    const int zb0 = (int)(te.y & 0xffffu), zb1 = (int)(te.y >> 16);
    if (zb1 <= zb0) return;                             // a task that received no layers
    // the column of this grid: mg.n1 x mg.n2 <= N1 x N2 cells (plan creation picks what fills the chip best)
    const int org1 = tx * mg.n1, org2 = ty * mg.n2;
    const int neff1 = min(mg.n1, g.Nover[0] - org1), neff2 = min(mg.n2, g.Nover[1] - org2);
    // the window in LDS: the column, plus the stencil's reach in the input-driven dimensions
    const int wx0 = org1 - C::XLO, wy0 = org2 - C::YLO;         // first cell of the window (may be negative: periodic)
    const int wnx = neff1 + C::XLO + C::XHI, wny = neff2 + C::YLO + C::YHI;
    const int nlay = zb1 - zb0;
    const int nq = 4 * nlay;                            // planes this task owns: q = 0 .. nq - 1 (plane 4 zb0 + q of the grid)
    const int nli = nlay + HLO + HHI;                   // layers of points it visits

    double* ring = reinterpret_cast<double*>(smem);
    const uint32_t ring_base = (uint32_t)(uintptr_t)ring;
    T* strip_wave = reinterpret_cast<T*>(smem + C::RING_BYTES + wave * C::strip_bytes());

    // ---- bins whose points can touch the column: rows of bins along y, one or two runs along x; lane i keeps run i:
    //      its first bin within a layer of bins and its length (lanes >= nruns: an empty run) ----
    uint32_t rb_bin = 0u, rb_len = 0u;
    {
        // (an input-driven dimension visits the column's own bins only: one run, no periodic image)
        const BinSegs seg0 = HX ? bin_segments(org1, org1 + neff1, g.Nover[0], 2, g.nb[0]) : bin_segments(org1 - M, org1 + neff1 + M - 1, g.Nover[0], 2, g.nb[0]);
        const BinSegs seg1 = HY ? bin_segments(org2, org2 + neff2, g.Nover[1], 2, g.nb[1]) : bin_segments(org2 - M, org2 + neff2 + M - 1, g.Nover[1], 2, g.nb[1]);
        const int nruns = seg1.total() * seg0.n;        // <= kMaxRuns (SMarchCfg::search)
        if (lane < nruns) {
            const int sg = lane % seg0.n, r2 = lane / seg0.n;
            rb_bin = (uint32_t)(seg1.bin(r2) * g.nb[0] + (sg ? seg0.lo[1] : seg0.lo[0]));
            rb_len = (uint32_t)(sg ? seg0.len[1] : seg0.len[0]);
        }
    }
    // bounds of the lane's run in layer `li` of the task
    auto load_run = [&](int li, uint32_t& r0, uint32_t& r1) __attribute__((always_inline)) {
        int lay = zb0 - HLO + li;
        if (lay < 0) lay += g.nb[2];
        if (lay >= g.nb[2]) lay -= g.nb[2];
        const uint32_t* o = a.offsets + ((int64_t)lay * g.nb[1] * g.nb[0] + rb_bin);
        r0 = o[0];
        r1 = o[rb_len];
    };

    // ---- zero the window ----
    {
        typedef double D2 __attribute__((ext_vector_type(2)));
        D2* r2p = reinterpret_cast<D2*>(ring);
        for (int i = tid; i < RZ * PS / 2; i += THREADS) r2p[i] = D2{0.0, 0.0};
    }
    uint32_t nx0, nx1;
    load_run(0, nx0, nx1);

    // evaluation roles (the instantiation fixes the evaluation mode)
    const EvalArgs<T, POLY ? NUFFT_EVAL_FAST_APPROXIMATION : NUFFT_EVAL_DIRECT> am(a);
    const int grp = lane / GP::G, q = lane % GP::G;
    T* strip = strip_wave + grp * (3 * L);
    WE we;
    we.init(am, q);
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
    const T* vin = a.vin[comp_id] + part;
    T* grid = a.grid[comp_id] + part;
    // halo variant: the side buffer's records (columns of mg.n1 x mg.n2 cells: the plan takes it only for grids they divide)
    const HaloLayout hl = make_halo_layout(mg.n1, mg.n2, M, NC, mg.ntx, mg.nty);
    __syncthreads();

#if NUFFT_SMARCH_DEFER
    typedef double HeldD2 __attribute__((ext_vector_type(2)));
    HeldD2 held[4];
#endif
    // what accum needs of a prepared chunk (its window values sit in the wave's strip)
    struct ChunkState { unsigned long long okmask; uint32_t soff; unsigned long long lmask; int q0, s0, s1; T vmine; };
    // Round 6: a wave prepares its FIRST chunk of the next layer (records requested a chunk into this layer; windows evaluated and staged in
    // its strip) before the two barriers of the retire pass, while the slower waves finish — after the pass the atomics start at once
    // instead of behind sixteen simultaneous window evaluations (the strip is the wave's own; the pass touches the window only).
    constexpr bool AHEAD = NUFFT_SMARCH_AHEAD && !NUFFT_SMARCH_SPLIT_TAIL && NUFFT_SMARCH_ABL == 0;
    bool have_carry = false;
    ChunkState carry{};
    for (int li = 0; li < nli; ++li) {
        // ---- the runs of this layer (requested a layer ago): chunks per run, inclusive scan over the lanes ----
        const uint32_t p0_l = nx0, p1_l = nx1;
        uint32_t cum_l = (p1_l - p0_l + PPW - 1) / PPW;
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t v = __shfl_up(cum_l, o, kWave);
            if (lane >= o) cum_l += v;
        }
        if (li + 1 < nli) load_run(li + 1, nx0, nx1);
        const int nchunks = __builtin_amdgcn_readlane((int)cum_l, kWave - 1);
        const int wq = 4 * (li - HLO) - (M - 1);        // first plane of the layer's window (in owned-plane coordinates)
        const bool clipz = wq < 0 || wq + RZ > nq;      // planes of this window belong to other segments

        auto lookup_in = [&](int item, uint32_t cum, uint32_t r0, uint32_t r1, uint32_t& p0, uint32_t& p1) __attribute__((always_inline)) {
            const unsigned long long mk = __ballot((uint32_t)item < cum);
            const int i = (int)__builtin_ctzll(mk);
            const uint32_t before = i ? (uint32_t)__builtin_amdgcn_readlane((int)cum, i - 1) : 0u;
            p0 = (uint32_t)__builtin_amdgcn_readlane((int)r0, i) + ((uint32_t)item - before) * PPW;
            p1 = (uint32_t)__builtin_amdgcn_readlane((int)r1, i);
        };
        auto lookup = [&](int item, uint32_t& p0, uint32_t& p1) __attribute__((always_inline)) { lookup_in(item, cum_l, p0_l, p1_l, p0, p1); };
        auto value_of = [&](const PointRec<T, 3>& r) __attribute__((always_inline)) -> T {
            T v = T(0);
            if (FAST || q < NC) {                       // (FAST: every lane of the group holds the point's value)
#if NUFFT_SMARCH_ABL == 6
                return T(r.idx);
#endif
                v = vin[(int64_t)r.idx * (NC * vgs) + (FAST ? 0 : q)];
                if (a.weights) v *= a.weights[r.idx];   // callbacks.nonuniform(v, n), src/spreading/gpu.jl:289
            }
            return v;
        };

        // one chunk of up to PPW points
        // (part, nparts: the chunks of a layer's last, incomplete round are shared by nparts waves each — every wave evaluates the windows
        // of the whole chunk, wave `part` adds the points gi with gi nparts / PPW == part)
        // prep: the chunk's windows into the wave's strip, what the accumulation needs from its records into `st` (wq_: first plane of the
        // window of the chunk's layer — this layer's, or the NEXT one's for the chunk a wave prepares ahead of the retire pass, see below)
        auto prep = [&](auto clip_c, const PointRec<T, 3>& rec, T vmine, bool have, int wq_, ChunkState& st) __attribute__((always_inline)) {
            constexpr bool CLIPZ = decltype(clip_c)::value;
            int s[2];
            T X[3];
            bool ok = have;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int c = cell_of(rec.r[d], g.Nover[d]);
                X[d] = rec.r[d] - T(c);
                int sd = c - (M - 1) - (d == 0 ? wx0 : wy0);        // first stencil node in window coordinates
                if ((d == 0 && !HX) || (d == 1 && !HY)) {           // output-driven: clipped to the column
                    const int ne = d == 0 ? neff1 : neff2;
                    if (sd > ne - 1) sd -= g.Nover[d];              // periodic image next to this column
                    if (sd < -(L - 1)) sd += g.Nover[d];
                    ok = ok && (sd >= -(L - 1)) && (sd <= ne - 1);
                }                                                   // (input-driven: the point's own column, 0 <= sd, sd + L <= window)
                s[d] = sd;
            }
            const int c3 = cell_of(rec.r[2], g.Nover[2]);
            X[2] = rec.r[2] - T(c3);
            const int dz = c3 & 3;                      // slot of the first stencil plane
            const int q0 = wq_ + dz;                    // its plane (owned-plane coordinates)
            if constexpr (CLIPZ) ok = ok && (q0 + L - 1 >= 0) && (q0 < nq);
            const unsigned long long okmask = __ballot(ok);
            st.okmask = okmask;
            if (okmask == 0ull) return;                 // nothing of this chunk touches the column
            // the point's LDS offset: stencil start in x, y and the slot of its first plane
            const uint32_t soff = (uint32_t)((s[0] * NC + s[1] * RS + dz * PS) * 8);
            unsigned long long lmask = 0ull;            // FAST: lanes (j1, j2) of the 8 x 8 face inside the column
            if constexpr (FAST && !(HX && HY)) {
                const int lo1 = HX ? 0 : min(L, max(0, -s[0])), hi1 = HX ? L : max(lo1, min(L, neff1 - s[0]));    // (a point that misses the column: empty ranges)
                const int lo2 = HY ? 0 : min(L, max(0, -s[1])), hi2 = HY ? L : max(lo2, min(L, neff2 - s[1]));
                const uint32_t m1 = ((1u << hi1) - 1u) & ~((1u << lo1) - 1u);
                const uint32_t col = m1 * 0x01010101u;
                const unsigned long long rows = hi2 > lo2 ? (~0ull >> (64 - 8 * (hi2 - lo2))) << (8 * lo2) : 0ull;
                lmask = ok ? (rows & (((unsigned long long)col << 32) | col)) : 0ull;
            }
            T wv[WE::NSLOT];
            we.eval_regs(am, X, wv);
            if constexpr (FAST) wv[0] *= vmine;         // slot 0 = dimension 1 (G = L = 8): the value rides on w1
            wave_lds_fence();
#pragma unroll
            for (int sl = 0; sl < WE::NSLOT; ++sl)
                if (we.has[sl]) strip[q + sl * GP::G] = wv[sl];
            wave_lds_fence();
            st.soff = soff; st.lmask = lmask; st.q0 = q0; st.s0 = s[0]; st.s1 = s[1]; st.vmine = vmine;
        };
        // accum: the staged chunk's points into the window
        auto accum = [&](auto clip_c, const ChunkState& st, int part, int nparts) __attribute__((always_inline)) {
            constexpr bool CLIPZ = decltype(clip_c)::value;
            const unsigned long long okmask = st.okmask;
            if (okmask == 0ull) return;
            const uint32_t soff = st.soff;
            const unsigned long long lmask = st.lmask;
            const int q0 = st.q0;
            const int s[2] = {st.s0, st.s1};
            const T vmine = st.vmine;
#if NUFFT_SMARCH_PRIO
            __builtin_amdgcn_s_setprio(NUFFT_SMARCH_PRIO);      // (experiment, round 6: the waves that feed the LDS atomic pipe ahead of those that evaluate windows)
#endif
#if NUFFT_SMARCH_ABL != 2
            auto do_point = [&](int gi, const T (&w1v)[NPASS], const T (&w2v)[NPASS], T w3a) __attribute__((always_inline)) {
                const int src = gi * GP::G;             // first lane of the point's group
                if (!((okmask >> src) & 1ull)) return;
                const uint32_t so = (uint32_t)__builtin_amdgcn_readlane((int)soff, src);
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
                unsigned planes = (1u << L) - 1u;
                if constexpr (CLIPZ) {
                    const int Q0 = __builtin_amdgcn_readlane(q0, src);
                    const int lo3 = max(0, -Q0), hi3 = min(L, nq - Q0);
                    planes = ((1u << hi3) - 1u) & ~((1u << lo3) - 1u);
                }
                if constexpr (FAST && HX && HY) {
                    // every lane of the 8 x 8 face lies inside the window: no mask
                    const T w = w1v[0] * w2v[0];
                    uint32_t vb[C::NBASE];
#pragma unroll
                    for (int b = 0; b < C::NBASE; ++b) vb[b] = lane_addr[0] + so + (uint32_t)(b * C::PLB * PSB);
                    T w3[L];
                    row_bcast_all(w3a, w3, 0, std::make_integer_sequence<int, L>{});
                    smarch_add_planes<C, CLIPZ>(vb, w, w3, planes, std::make_integer_sequence<int, L>{});
                } else if constexpr (FAST) {
                    const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)lmask, src);
                    const uint32_t mhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(lmask >> 32), src);
                    const T w = w1v[0] * w2v[0];
                    uint32_t vb[C::NBASE];
#pragma unroll
                    for (int b = 0; b < C::NBASE; ++b) vb[b] = lane_addr[0] + so + (uint32_t)(b * C::PLB * PSB);
#if !defined(NUFFT_SMARCH_W3_SPLIT)
                    if (__builtin_amdgcn_inverse_ballot_w64(((unsigned long long)mhi << 32) | mlo))
                        smarch_add_planes<C, CLIPZ>(vb, w, w3, planes, std::make_integer_sequence<int, L>{});
#else
                    // the window values of dimension 3 four planes at a time (8 registers less; measured slower)
                    static_assert(!FAST || (L == 8 && C::PLB == 4), "FAST: two groups of four planes");
                    T w3q[4];
                    row_bcast4<0>(w3a, w3q);
                    if (__builtin_amdgcn_inverse_ballot_w64(((unsigned long long)mhi << 32) | mlo)) smarch_add_four<C, CLIPZ, 0>(vb, w, w3q, planes);
                    row_bcast4<4>(w3a, w3q);
                    if (__builtin_amdgcn_inverse_ballot_w64(((unsigned long long)mhi << 32) | mlo)) smarch_add_four<C, CLIPZ, 4>(vb, w, w3q, planes);
#endif
                } else {
                    const int S1 = __builtin_amdgcn_readlane(s[0], src);
                    const int S2 = __builtin_amdgcn_readlane(s[1], src);
                    const T Vre = readlane_t(vmine, src);
                    const T Vim = CPLX ? readlane_t(vmine, src + (CPLX ? 1 : 0)) : T(0);
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) {
                        const int l1 = S1 + j1f[ps], l2 = S2 + j2f[ps];
                        const bool lane_ok = actf[ps] && (HX || (unsigned)l1 < (unsigned)neff1) && (HY || (unsigned)l2 < (unsigned)neff2);
                        const T w = w1v[ps] * (CPLX ? (cmpf[ps] ? Vim : Vre) : Vre) * w2v[ps];
                        uint32_t vb[C::NBASE];
#pragma unroll
                        for (int b = 0; b < C::NBASE; ++b) vb[b] = lane_addr[ps] + so + (uint32_t)(b * C::PLB * PSB);
                        if (lane_ok) smarch_add_planes<C, CLIPZ>(vb, w, w3, planes, std::make_integer_sequence<int, L>{});
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
                    if (nparts > 1 && g0 * nparts / PPW != part) return;
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
                    if (nparts > 1 && gi * nparts / PPW != part) continue;
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
            asm volatile("" ::"v"(s[0]), "v"(s[1]), "v"(soff), "v"(lmask));
#endif
#if NUFFT_SMARCH_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        auto chunk = [&](auto clip_c, const PointRec<T, 3>& rec, T vmine, bool have, int part, int nparts) __attribute__((always_inline)) {
            ChunkState st;
            prep(clip_c, rec, vmine, have, wq, st);
            accum(clip_c, st, part, nparts);
        };

        // ---- chunks wave, wave + NW, ... of this layer; the next chunk's records are requested before the current one is
        //      processed, its values right after.  (Tried, round 5: the first chunk of the NEXT layer requested before the two barriers of
        //      the retire pass, so that no wave starts a layer waiting for records — C2 spread stage 1.89 -> 1.87 ms with the polynomial
        //      window, 1.98 -> 2.05 with Direct(): nothing to gain, the other waves already cover that latency.) ----
        {
            // the last, incomplete round of the layer (rem < NW chunks left): 2 or 4 waves share a chunk, so that the LDS atomic pipe does not
            // idle behind a few waves — a layer of a 32 x 32 column holds ~38 chunks at C2: rounds of 16, 16 and 6.  Measured (round 5, C2
            // spread stage): Direct() 1.98 -> 2.11 ms, polynomial window 1.90 -> 1.93: every sharing wave evaluates the windows of the
            // whole chunk, which costs more than the idle lanes of the tail — off
#ifndef NUFFT_SMARCH_SPLIT_TAIL
#define NUFFT_SMARCH_SPLIT_TAIL 0
#endif
            const int full = nchunks / NW * NW, rem = nchunks - full;
            const int tparts = (NUFFT_SMARCH_SPLIT_TAIL && PPW >= 4 && rem > 0) ? (rem * 4 <= NW ? 4 : (rem * 2 <= NW ? 2 : 1)) : 1;
            auto item_of = [&](int it, int& part, int& np) __attribute__((always_inline)) -> int {      // chunk of the wave's item `it` (-1: none)
                part = 0; np = 1;
                if (it < full) return it;
                const int w = it - full;            // = wave, in the tail round
                if (w >= rem * tparts) return -1;
                part = w % tparts; np = tparts;
                return full + w / tparts;
            };
            int it = wave, part = 0, nparts = 1;
            int item = item_of(it, part, nparts);
            uint32_t p0 = 0, p1 = 0;
            bool valid = item >= 0;
            PointRec<T, 3> rec{};
            T vcur = T(0);
            const bool carried = AHEAD && have_carry;           // (then item == wave is valid: the chunk was found in this layer's runs a layer ago)
            if (valid && !carried) {
                lookup(item, p0, p1);
                rec = sorted[min(p0 + (uint32_t)grp, p1 - 1)];
                vcur = value_of(rec);
            }
            // the wave's first chunk of the NEXT layer: requested once this layer's first chunk is done (the next layer's runs, requested at the
            // top of this layer, have arrived by then), prepared behind the last chunk
            bool ahead_req = false, ahead_valid = false, ahead_have = false;
            PointRec<T, 3> rec_a{};
            T v_a = T(0);
            auto request_ahead = [&]() __attribute__((always_inline)) {
                ahead_req = true;
                if (li + 1 >= nli) return;
                uint32_t cum_n = (nx1 - nx0 + PPW - 1) / PPW;
                for (int o = 1; o < kWave; o <<= 1) {
                    const uint32_t v = __shfl_up(cum_n, o, kWave);
                    if (lane >= o) cum_n += v;
                }
                const int nchunks_n = __builtin_amdgcn_readlane((int)cum_n, kWave - 1);
                if (wave < nchunks_n) {
                    uint32_t a0, a1;
                    lookup_in(wave, cum_n, nx0, nx1, a0, a1);
                    rec_a = sorted[min(a0 + (uint32_t)grp, a1 - 1)];
                    v_a = value_of(rec_a);
                    ahead_have = a0 + (uint32_t)grp < a1;
                    ahead_valid = true;
                }
            };
            bool first = true;
            while (valid) {
                it += NW;
                int partn = 0, npartsn = 1;
                item = it < full + NW ? item_of(it, partn, npartsn) : -1;
                const bool validn = item >= 0;
                uint32_t n0 = 0, n1 = 0;
                PointRec<T, 3> recn = rec;
                if (validn) {
                    lookup(item, n0, n1);
                    recn = sorted[min(n0 + (uint32_t)grp, n1 - 1)];
                }
                if (AHEAD && first && carried) {
                    if (clipz) accum(std::true_type{}, carry, part, nparts);
                    else accum(std::false_type{}, carry, part, nparts);
                } else {
                    const bool have = p0 + (uint32_t)grp < p1;
                    if (clipz) chunk(std::true_type{}, rec, vcur, have, part, nparts);
                    else chunk(std::false_type{}, rec, vcur, have, part, nparts);
                }
                first = false;
                part = partn; nparts = npartsn;
                // (requesting the next values in the middle of the visits instead — half a chunk more slack — measured slower:
                // 2.50 against 2.45 ms at C2; the value gather costs 0.1 ms in all, ablation 6)
                if (validn) vcur = value_of(recn);
                rec = recn; p0 = n0; p1 = n1; valid = validn;
                if (AHEAD && !ahead_req) request_ahead();
            }
            if constexpr (AHEAD) {
                if (!ahead_req) request_ahead();                // (a wave without a chunk in this layer)
                have_carry = false;
                if (ahead_valid) {
                    const int wqn = wq + 4;
                    if (wqn < 0 || wqn + RZ > nq) prep(std::true_type{}, rec_a, v_a, ahead_have, wqn, carry);
                    else prep(std::false_type{}, rec_a, v_a, ahead_have, wqn, carry);
                    have_carry = true;
                }
            }
        }
        // the immediate-offset atomics are inline assembly: the compiler does not know that they are in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if NUFFT_SMARCH_ABL != 5
        lds_barrier();

#if NUFFT_SMARCH_DEFER
        // (experiment, round 6: the stores of the finished planes — address arithmetic and issue — behind the second barrier, next to the
        // next layer's first chunks, instead of between the barriers where every wave waits for them)
        const bool defer = wny * ((NC * wnx + 1) / 2) <= THREADS;
        if (defer) {
#define NUFFT_RETIRE_SECTION 1
#include "smarch_retire.inc"
        } else {
#include "smarch_retire.inc"
        }
        lds_barrier();
        if (defer) {
            const int held_wq = wq;
#define NUFFT_RETIRE_SECTION 2
#include "smarch_retire.inc"
        }
#else
#include "smarch_retire.inc"
        lds_barrier();      // (not __syncthreads(): that would wait for the retire pass's global stores to be acknowledged, once per layer)
#endif
#endif
    }
}

}  // namespace nufft
