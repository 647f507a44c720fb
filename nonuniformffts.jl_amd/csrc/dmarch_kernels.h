// Type-1 spreading of DENSE point sets on the z-marching LDS window: the fourth spreading engine (round 6).
//
// Same stage as spread_march_kernel's halo variant (reference src/spreading/gpu.jl:237-377 + the zero fill of src/NonuniformFFTs.jl:161-167;
// smarch_kernels.h has the window, the side buffer, the tasks): a workgroup owns a column of the grid, marches along z with a window of
// RZ = 2M + 3 planes in LDS, every point is spread once by its own column.  What changes is how a point reaches the window.  There, every
// point costs 2M (M = 4) to 3 x 2M (M = 6) Float64 LDS atomics of 64 lanes — 8.4 LDS-array cycles each, the pipe that binds the kernel
// (67 of 115 CU-cycles per point at C2).  Here the points of a 4^3-cell BIN are first accumulated in REGISTERS by the FP64 matrix pipe:
// all stencils of a bin lie inside its footprint of FP^3 cells, FP = 2M + 3 = the window depth, and with zero-padded window rows
//      C[face position (x, y) of FP x FP][plane z of FP] += sum over points k:  (w1pad_k[x] w2pad_k[y]) (v_k w3pad_k[z])
// is a rank-4 update per batch of four points: NT = ceil(FP^2 / 16) instructions v_mfma_f64_16x16x4 (A = the face products of the four
// points, B = their value-weighted dimension-3 rows), accumulators 4 NT registers per lane.  The footprint then goes to the window ONCE
// per bin: 4 NT atomics whatever the bin holds.  Per point that is NT / 4 matrix instructions of 64 cycles on one of four matrix pipes
// (8 tiles at M = 4: 32 CU-cycles) plus 4 NT x 8.4 / n flush cycles for n points in the bin — scripts/microbench12.hip measured 56 / 43
// CU-cycles per point at n = 19 / 64 against 67 for the atomic stream, and 131 at C2's n = 4.8: an engine for dense sets at M = 4 (the
// densities of the reference's published benchmark, rho = Np / N^3 >= 0.3 — where type 1 was half of type 2), and for every density at
// M = 5, 6, where a point costs 20 / 36 atomics.  set_points picks it per point set from the mean bin load (plan.cpp).
//
// A wave takes the bins wave, wave + NW, ... of the column's current layer (their runs of the bin-sorted array in registers, the next
// layer's bounds in flight — as spread_march_kernel keeps its rows of bins).  Sixteen records at a time are loaded one per lane quad
// position (lane (k, i) of point group k = lane / 16 holds point 4 (i mod 4) + k of the sixteen), cell and cell fraction are computed
// where the record sits, and each of the four batches broadcasts its points inside the quads (DPP quad_perm).  Window values: the
// group mapping of the other kernels (WindowEval, 16 lanes per point), written into the wave's LDS strip [point][dimension][16] at
// offset (cell mod 4) inside zero-filled rows — the padding.  Operands: lane (k, i) reads its face products from two strip rows,
// tile row i holds face position 16 t + i.  The C/D layout of the instruction (column = lane mod 16, row = lane / 16 + 4 x register)
// puts the face positions 16 t + 4 r + 0..3 of accumulator register r into the four lane groups: a flush instruction adds four
// CONSECUTIVE cells of a window row on planes z = lane mod 16 — with the plane stride congruent to 2 modulo the 32 double-word banks
// no two lanes of a half-wave share a bank.
#pragma once
#ifndef NUFFT_DMARCH_PRIO
#define NUFFT_DMARCH_PRIO 0        // 1..3: wave priority while a wave issues a batch's matrix instructions / flushes a bin (experiment, round 6)
#endif

#include "smarch_kernels.h"

namespace nufft {

typedef double DMv4 __attribute__((ext_vector_type(4)));

template <typename T, int M>
struct DMarchCfg {
    using S = SMarchCfg<T, false, M, true, true>;       // the halo variant's window: column bound, row stride, reach, layers visited
    static constexpr int NC = 1;
    static constexpr int L = 2 * M, RZ = L + 3, FP = L + 3;
    static constexpr int NFACE = FP * FP, NT = (NFACE + 15) / 16;
    static_assert(FP <= 15, "a padded window row has 16 entries, the last one always zero");
#ifndef NUFFT_DMARCH_THREADS
#define NUFFT_DMARCH_THREADS 512
#endif
#ifndef NUFFT_DMARCH_ABL
#define NUFFT_DMARCH_ABL 0          // ablation builds, bit mask: 1 = no flush atomics, 2 = no matrix instructions, 4 = no window evaluation, 8 = no operand reads,
                                    // 16 = no record / value loads, 32 = no strip writes
#endif
    static constexpr int THREADS = NUFFT_DMARCH_THREADS, NW = THREADS / kWave;      // 8 waves: 256 registers per lane (4 NT accumulators: 120 at M = 6)
    static constexpr int HLO = S::HLO, HHI = S::HHI;
    static constexpr int XLO = S::XLO, XHI = S::XHI, YLO = S::YLO, YHI = S::YHI;
    static constexpr int N1 = S::N1, N2 = S::N2;
    static constexpr bool FITS = S::FITS && M <= 6;
    static constexpr int RS = S::RS, WY = S::WY;
    // plane stride = 2 (mod 32 doubles): the flush's 11 .. 15 planes x 2 consecutive cells of a half-wave on distinct banks (and even: pairs stay aligned)
    static constexpr int PS = RS * WY + ((2 - (RS * WY) % 32) + 32) % 32;
    static constexpr int PSB = PS * 8;
    static constexpr int RING_BYTES = round_up(RZ * PSB, 16);
    static constexpr int STRIP_DOUBLES = 4 * 3 * 16;    // [point of the batch][dimension][padded row]
    static constexpr int strip_bytes() { return STRIP_DOUBLES * 8; }
    static constexpr int TAB_WORDS = 16 * NT;           // [tile][lane group q][register r] -> offset (in reals) of the face position in a plane, or ~0
    static constexpr int lds_bytes() { return RING_BYTES + NW * strip_bytes() + TAB_WORDS * 4 + 64; }
    static_assert(!FITS || lds_bytes() <= 163840 - 256, "window + strips exceed the LDS");
};

// Orders the LDS traffic of one wave for the COMPILER only.  The hardware executes the LDS instructions of a wave in issue order (a ds_read
// behind a ds_write of the same wave sees the written value), so no s_waitcnt is needed between them — wave_lds_fence() (device_common.h)
// drained the LDS queue three times per batch here: 0.5 of 3.0 ms at rho = 1.
__device__ __forceinline__ void lds_order() { asm volatile("" ::: "memory"); }

// quad broadcast: every lane takes the value of lane B of its quad (DPP quad_perm [B, B, B, B])
template <int B>
__device__ __forceinline__ int quad_bcast(int x) { return __builtin_amdgcn_mov_dpp(x, B * 0x55, 0xf, 0xf, true); }
template <int B>
__device__ __forceinline__ float quad_bcast(float x) { return __builtin_bit_cast(float, quad_bcast<B>(__builtin_bit_cast(int, x))); }
template <int B>
__device__ __forceinline__ double quad_bcast(double x) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)quad_bcast<B>((int)(unsigned)u), hi = (unsigned)quad_bcast<B>((int)(unsigned)(u >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <typename T, int M, bool POLY>
__global__ __launch_bounds__((DMarchCfg<T, M>::THREADS)) void spread_march_dense_kernel(TileArgs<T> a, MarchGeom mg) {
    using C = DMarchCfg<T, M>;
    using WE = WindowEval<T, 1, 3, M, 16, false>;
    constexpr int NC = 1, L = C::L, RZ = C::RZ, FP = C::FP, NT = C::NT, RS = C::RS, PS = C::PS, PSB = C::PSB;
    constexpr int HLO = C::HLO, HHI = C::HHI, THREADS = C::THREADS, NW = C::NW;
    constexpr bool HX = true, HY = true;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    if (mg.halo_state && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *mg.halo_state = *mg.flag != 0u ? 1u : 0u;      // the side buffer this launch writes is pending (smarch_kernels.h)
    if (*mg.flag == 0u) return;                         // spread_tile_kernel serves this point set
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const Geom& g = a.g;
    // (complex data part by part through this real kernel: the two parts of a task side by side on one XCD — smarch_kernels.h)
    const bool split = mg.parts == 2;
    const int comp_id = (int)blockIdx.y;
    const int slot = (int)blockIdx.x >> 3;
    const int part = split ? slot & 1 : 0;
    const int vgs = split ? 2 : 1;
    const int yrow = split ? 2 * comp_id + part : comp_id;
    const int vblock = split ? (slot >> 1) * 8 + ((int)blockIdx.x & 7) : (int)blockIdx.x;
    const int task = xcd_remap_chunked(vblock, split ? (int)gridDim.x >> 1 : (int)gridDim.x, a.xcd_chunk);
    if (task >= mg.ntasks) return;
    const uint2 te = mg.tasktab[task];
    const int tx = (int)te.x % mg.ntx, ty = (int)te.x / mg.ntx;
    const int zb0 = (int)(te.y & 0xffffu), zb1 = (int)(te.y >> 16);
    if (zb1 <= zb0) return;
    const int org1 = tx * mg.n1, org2 = ty * mg.n2;
    const int neff1 = min(mg.n1, g.Nover[0] - org1), neff2 = min(mg.n2, g.Nover[1] - org2);      // (= mg.n1, mg.n2: the halo variant's columns divide the axes)
    const int wnx = neff1 + C::XLO + C::XHI, wny = neff2 + C::YLO + C::YHI;
    const int nlay = zb1 - zb0;
    const int nq = 4 * nlay;
    const int nli = nlay + HLO + HHI;

    double* ring = reinterpret_cast<double*>(smem);
    double* strip = reinterpret_cast<double*>(smem + C::RING_BYTES) + wave * C::STRIP_DOUBLES;
    uint32_t* facetab = reinterpret_cast<uint32_t*>(smem + C::RING_BYTES + NW * C::strip_bytes());

    // ---- bins of the column: lane j of a wave keeps bin wave + NW j (the column's own bins only: every point is spread once) ----
    const int nbx = neff1 >> 2, nby = neff2 >> 2, nbins = nbx * nby;
    const int mybin = wave + NW * lane;
    uint32_t rb_bin = 0u;
    const bool rb_ok = mybin < nbins;
    if (rb_ok) rb_bin = (uint32_t)(((org2 >> 2) + mybin / nbx) * g.nb[0] + (org1 >> 2) + mybin % nbx);
    const int nbw = nbins > wave ? (nbins - wave + NW - 1) / NW : 0;      // bins of this wave per layer (a scalar)
    auto load_run = [&](int li, uint32_t& r0, uint32_t& r1) __attribute__((always_inline)) {
        int lay = zb0 - HLO + li;
        if (lay < 0) lay += g.nb[2];
        if (lay >= g.nb[2]) lay -= g.nb[2];
        r0 = r1 = 0u;
        if (rb_ok) {
            const uint32_t* o = a.offsets + ((int64_t)lay * g.nb[1] * g.nb[0] + rb_bin);
            r0 = o[0];
            r1 = o[1];
        }
    };

    // ---- zero the window; the flush table ----
    {
        typedef double D2 __attribute__((ext_vector_type(2)));
        D2* r2p = reinterpret_cast<D2*>(ring);
        for (int i = tid; i < RZ * PS / 2; i += THREADS) r2p[i] = D2{0.0, 0.0};
        for (int i = tid; i < C::TAB_WORDS; i += THREADS) {
            const int t = i >> 4, qq = (i >> 2) & 3, r = i & 3;
            const int f = 16 * t + 4 * r + qq;          // face position of accumulator register r, lane group qq, tile t
            facetab[i] = f < C::NFACE ? (uint32_t)((f / FP) * RS + f % FP) : 0xffffffffu;
        }
    }
    uint32_t nx0, nx1;
    load_run(0, nx0, nx1);

    const EvalArgs<T, POLY ? NUFFT_EVAL_FAST_APPROXIMATION : NUFFT_EVAL_DIRECT> am(a);
    const int k = lane >> 4, i16 = lane & 15, qd = lane & 3;      // point group, lane of the group, position in the quad
    WE we;
    we.init(am, i16);
    // the lane's strip rows [dimension][16] of its point group
    double* srow = strip + k * 48;
    // operands: tile t, row i16 of the tile = face position 16 t + i16.  (C/D layout of v_mfma_f64_16x16x4: column = lane mod 16, row =
    // lane / 16 + 4 x register — the four lane groups of accumulator register r hold the CONSECUTIVE face positions 16 t + 4 r + 0..3)
    int fxy[NT];                                        // fx | fy << 8
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int f = 16 * t + i16;
        const int fx = f < C::NFACE ? f % FP : 15, fy = f < C::NFACE ? f / FP : 15;      // (entry 15 of a padded row is always zero)
        fxy[t] = fx | (fy << 8);
    }
    const PointRec<T, 3>* sorted = static_cast<const PointRec<T, 3>*>(a.sorted);
    const T* vin = a.vin[comp_id] + part;
    T* grid = a.grid[comp_id] + part;
    const HaloLayout hl = make_halo_layout(mg.n1, mg.n2, M, NC, mg.ntx, mg.nty);
    const uint4* tab_lane = reinterpret_cast<const uint4*>(facetab) + k;      // this lane group's [r] quadruples: + 4 t
    __syncthreads();

    for (int li = 0; li < nli; ++li) {
        const uint32_t p0_l = nx0, p1_l = nx1;
        if (li + 1 < nli) load_run(li + 1, nx0, nx1);
        const int wq = 4 * (li - HLO) - (M - 1);        // first plane of the layer's window (in owned-plane coordinates)
        // planes of this layer's window the task owns (the first and last layers of a segment reach into the neighbouring segments)
        const bool zok = i16 < RZ && wq + i16 >= 0 && wq + i16 < nq;

        // ---- the wave's bins of this layer as ONE sequence of loads of sixteen records: the record load runs two steps ahead of the matrix
        //      work, the value load (which needs the record's index) one step — a load of sixteen is 4 batches x NT matrix instructions
        //      of work for the wave, about the latency of one trip to memory ----
        struct Step { int jb; uint32_t p, r1; };        // bin of the wave, first record of the load, end of the bin's run (scalars)
        auto advance = [&](Step& st) __attribute__((always_inline)) -> bool {
            st.p += 16u;
            while (st.p >= st.r1) {
                if (++st.jb >= nbw) return false;
                st.p = (uint32_t)__builtin_amdgcn_readlane((int)p0_l, st.jb);
                st.r1 = (uint32_t)__builtin_amdgcn_readlane((int)p1_l, st.jb);
            }
            return true;
        };
        // (every load of the pipeline is UNCONDITIONAL, from an index clamped into the array: a load behind a branch makes the compiler wait for
        // all outstanding loads where the branches join — the first version waited out both round trips in every step, 0.41 of 3.0 ms)
        const T* wsrc = a.weights ? a.weights : vin;    // (no weights: a valid address, the value is not used)
        const int wstride = a.weights ? 1 : vgs;
        auto load_rec = [&](const Step& st, bool ok) __attribute__((always_inline)) -> PointRec<T, 3> {
            const uint32_t pr = st.p + 4u * (uint32_t)qd + (uint32_t)k;
#if NUFFT_DMARCH_ABL & 16
            PointRec<T, 3> fake{};
            fake.r[0] = T(org1 + 4 * (mybin % nbx)) + T(0.37) * T(1 + qd); fake.r[1] = T(org2 + 4 * ((mybin / nbx) % nby)) + T(0.21) * T(1 + k); fake.r[2] = T(1.5) + T(pr & 1u);
            fake.idx = (int)pr;
            return fake;
#endif
            const uint32_t last = ok ? st.r1 - 1u : 0u;
            return sorted[ok && pr < st.r1 ? pr : last];
        };
        struct Val { T v, w; };
        auto load_val = [&](const PointRec<T, 3>& rec) __attribute__((always_inline)) -> Val {
#if NUFFT_DMARCH_ABL & 16
            return Val{T(rec.idx & 7), T(1)};
#endif
            return Val{vin[(int64_t)rec.idx * vgs], wsrc[(int64_t)rec.idx * wstride]};
        };
        // value of the lane's slot: 0 beyond the run (its row of B is then zero), times the point's weight of the callback menu
        // (callbacks.nonuniform(v, n), src/spreading/gpu.jl:289)
        auto value_of = [&](const Step& st, const Val& x) __attribute__((always_inline)) -> T {
            const bool have = st.p + 4u * (uint32_t)qd + (uint32_t)k < st.r1;
            const T v = a.weights ? x.v * x.w : x.v;
            return have ? v : T(0);
        };
        Step s0{-1, 0u, 0u};
        bool ok0 = advance(s0);
        Step s1 = s0;
        bool ok1 = ok0 && advance(s1);
        Step s2 = s1;
        bool ok2 = ok1 && advance(s2);
        PointRec<T, 3> rec0{}, rec1{}, rec2{};
        Val val0{T(0), T(1)}, val1{T(0), T(1)};
        if (ok0) {                                       // (a layer without points in the wave's bins: nothing is loaded at all)
            rec0 = load_rec(s0, true);
            rec1 = load_rec(s1, ok1);
            val0 = load_val(rec0);
        }
        DMv4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = DMv4{0.0, 0.0, 0.0, 0.0};
        // Software pipeline over the batches of four points: the operands of batch i + 1 are built (window evaluation, padded rows into the
        // strip, face products read back into registers) BEFORE the matrix instructions of batch i are issued — their issue (NT x 64 cycles
        // of the matrix pipe) then covers the strip's LDS round trip, and the other wave of the SIMD evaluates windows meanwhile.
        // (First version: evaluate, write, read, wait, multiply, issue — per batch: VALU 37 %, matrix pipe 25 %, LDS 32 % busy, waves waiting
        // 46 % of their cycles at two waves per SIMD: 3.85 ms at rho = 1 against 2.97 ms for the stream of atomics.)
        struct Ops { double ap[NT]; double b; };        // A (face products of the lane's tile rows) and B (its plane's value-weighted dimension-3 entry)
        Ops cur{};
        bool pending = false;                            // `cur` holds a batch whose matrix instructions have not been issued
        int cur_jb = 0;                                  // ... of this bin of the wave
        auto issue = [&](const Ops& o) __attribute__((always_inline)) {
#if NUFFT_DMARCH_PRIO
            __builtin_amdgcn_s_setprio(NUFFT_DMARCH_PRIO);      // (experiment, round 6: the wave that issues matrix instructions / atomics first)
#endif
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#if NUFFT_DMARCH_ABL & 2
                acc[t][0] += o.ap[t] * o.b;
#else
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.ap[t], o.b, acc[t], 0, 0, 0);
#endif
            }
#if NUFFT_DMARCH_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        auto flush = [&](int jb) __attribute__((always_inline)) {
#if NUFFT_DMARCH_PRIO
            __builtin_amdgcn_s_setprio(NUFFT_DMARCH_PRIO);
#endif
            // the bin is complete: its footprint onto the window, 4 NT atomics (lanes: plane z = lane mod 16, four consecutive face positions)
            const int b = wave + NW * jb, bxi = b % nbx, byi = b / nbx;
            // footprint origin of the bin in window coordinates: cell 4 b - (M - 1) relative to the window's first cell; ... on this lane's plane
            double* bin_base = ring + ((4 * byi + C::YLO - (M - 1)) * RS + 4 * bxi + C::XLO - (M - 1)) + i16 * PS;
            // (the table entries of ALL tiles first: LDS instructions return in order, so a read behind four atomics waits for them —
            // tile by tile the flush was 8 round trips per bin, 0.6 of 3.0 ms at rho = 1)
            // (M = 6: 15 tiles — four at a time, the registers are the accumulators')
            constexpr int TG = NT <= 11 ? NT : 4;
            uint4 off[TG];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t % TG == 0) {
#pragma unroll
                    for (int u = 0; u < TG; ++u)
                        if (t + u < NT) off[u] = tab_lane[4 * (t + u)];
                }
                const uint32_t offs[4] = {off[t % TG].x, off[t % TG].y, off[t % TG].z, off[t % TG].w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#if NUFFT_DMARCH_ABL & 1
                    asm volatile("" ::"v"(acc[t][r]), "v"(offs[r]));
#else
                    if (zok && offs[r] != 0xffffffffu) atomicAdd(bin_base + offs[r], acc[t][r]);
#endif
                }
                acc[t] = DMv4{0.0, 0.0, 0.0, 0.0};
            }
#if NUFFT_DMARCH_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
        };
        while (ok0) {
            val1 = load_val(rec1);
            rec2 = load_rec(s2, ok2);
            {
                // ---- sixteen records: lane (k, i) holds point 4 (i mod 4) + k of them; its cell, fraction and value ----
                const uint32_t p = s0.p, r1 = s0.r1;
                T X[3];
                int spk = 0;
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    const int c = cell_of(rec0.r[d], g.Nover[d]);
                    X[d] = rec0.r[d] - T(c);
                    spk |= (c & 3) << (2 * d);          // offset of the stencil inside the bin's footprint
                }
                const T v = value_of(s0, val0);
                // operands of batch B of the sixteen (into registers), then the matrix instructions of the batch before it
                auto batch = [&](auto bc) __attribute__((always_inline)) {
                    constexpr int B = decltype(bc)::value;
                    T Xb[3];
#pragma unroll
                    for (int d = 0; d < 3; ++d) Xb[d] = quad_bcast<B>(X[d]);
                    const int sb = quad_bcast<B>(spk);
                    const T vb = quad_bcast<B>(v);
                    T wv[WE::NSLOT];
#if NUFFT_DMARCH_ABL & 4
#pragma unroll
                    for (int sl = 0; sl < WE::NSLOT; ++sl) wv[sl] = Xb[sl % 3];
#else
                    we.eval_regs(am, Xb, wv);
#endif
                    lds_order();                   // (the previous batch's operands are in registers already)
#if NUFFT_DMARCH_ABL & 32
                    asm volatile("" ::"v"(wv[0]), "v"(sb), "v"(vb));
#else
#pragma unroll
                    for (int d = 0; d < 3; ++d) srow[16 * d + i16] = 0.0;
                    lds_order();
#pragma unroll
                    for (int sl = 0; sl < WE::NSLOT; ++sl)
                        if (we.has[sl]) {
                            const int d = we.dsel[sl];
                            const int sd = (sb >> (2 * d)) & 3;
                            srow[16 * d + sd + we.jsel[sl]] = (double)(d == 2 ? wv[sl] * vb : wv[sl]);
                        }
#endif
                    lds_order();
                    Ops nx;
                    // (wide stencils: 2 NT operand registers in flight next to 4 NT accumulators do not fit 256 registers — there the matrix
                    // instructions of the batch before go out first, over the strip writes, and the products are formed as the reads return)
                    // (M = 6: 15 tiles — 120 accumulator registers; no batch is held back at all: its own matrix instructions follow its products)
                    constexpr bool READS_FIRST = NT <= 8, LAG = NT <= 11;
                    if constexpr (!READS_FIRST && LAG) {
                        if (pending) {
                            issue(cur);
                            if (cur_jb != s0.jb) flush(cur_jb);       // ... and it was the last batch of its bin
                        }
                    }
                    nx.b = srow[32 + i16];
                    if constexpr (!LAG) {
                        // no batch held back: read, multiply, issue — tile by tile (nothing but the accumulators stays live)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const double a1 = srow[fxy[t] & 0xff], a2 = srow[16 + (fxy[t] >> 8)];
#if NUFFT_DMARCH_ABL & 2
                            acc[t][0] += a1 * a2 * nx.b;
#else
                            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1 * a2, nx.b, acc[t], 0, 0, 0);
#endif
                        }
                    } else {
                    double a1[NT], a2[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
#if NUFFT_DMARCH_ABL & 8
                        a1[t] = nx.b; a2[t] = (double)(fxy[t] + 1);
#else
                        a1[t] = srow[fxy[t] & 0xff];
                        a2[t] = srow[16 + (fxy[t] >> 8)];
#endif
                        if constexpr (!READS_FIRST) nx.ap[t] = a1[t] * a2[t];
                    }
                    if constexpr (READS_FIRST) {
                        // the batch before this one: its matrix instructions go out while the reads above are in flight
                        if (pending) {
                            issue(cur);
                            if (cur_jb != s0.jb) flush(cur_jb);       // ... and it was the last batch of its bin
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) nx.ap[t] = a1[t] * a2[t];
                    }
                    }
                    if constexpr (LAG) {
                        cur = nx;
                        cur_jb = s0.jb;
                        pending = true;
                    }
                };
                batch(std::integral_constant<int, 0>{});
                if (p + 4u < r1) batch(std::integral_constant<int, 1>{});
                if (p + 8u < r1) batch(std::integral_constant<int, 2>{});
                if (p + 12u < r1) batch(std::integral_constant<int, 3>{});
            }
            if constexpr (NT > 11) {
                if (!ok1 || s1.jb != s0.jb) flush(s0.jb);             // (no batch held back: the bin ends with this load of sixteen)
            }
            s0 = s1; rec0 = rec1; val0 = val1; ok0 = ok1;
            s1 = s2; rec1 = rec2; ok1 = ok2;
            if (ok2) ok2 = advance(s2);
        }
        if (pending) {
            issue(cur);
            flush(cur_jb);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        lds_barrier();

#include "smarch_retire.inc"
        lds_barrier();
    }
}

}  // namespace nufft
