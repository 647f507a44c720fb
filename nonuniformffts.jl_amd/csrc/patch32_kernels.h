// Type-1 spreading of ComplexF32 plans on register-resident patches accumulated by the FP32 matrix pipe (gfx950, wave64).
//
// The Float32 sibling of patch_kernels.h (same decomposition: a wave owns a patch of cube columns and marches along
// dimension 3 through a segment, output-driven, no atomics, no zero fill, every cell written once) with the sums
// accumulated in Float32 — the reference accumulates in T = real(Z) (src/spreading/gpu.jl:271-283: the LDS tile has
// element type Z; :381-403 adds it to the Float32 grid), so this is the reference's arithmetic, not a precision trade.
//
//     G[(x, y), (z, c)] += sum_p  A[(x, y), p] * B[p, (z, c)],    A = w1_p[x] * w2_p[y],   B = v_p[c] * w3_p[z]
//
// on v_mfma_f32_16x16x4_f32: i = (x, y) of a 4 x 4 cube face, j = (z, component) of an OCTET of 8 planes, k = 4 points.
// One instruction forms 16 x 16 x 4 = 1024 products in 36 cycles (measured, scripts/microbench8.hip) — 2x the rate of
// v_mfma_f64_4x4x4_4b (256 in 18) — and its 256 results cost 4 accumulator registers instead of 8, so the same register
// file holds a patch of twice the area: PBX x PBY = 4 x 7 cube columns x 3 octets at M = 8 (4 x 3 x 5 cubes in Float64),
// i.e. 3.1 instead of 4.7 visits per point, each of which re-evaluates the 3 x 2M window values and sets the operands up.
//
// Layouts (measured by scripts/microbench8.hip): A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k, D[4 (l / 16) + r][l % 16]
// in register r of lane l.  With i = x + 4 y: lane l holds the cells x = 0..3 (registers), y = l / 16, z = (l % 16) / 2,
// component l % 2 of its cube column and octet — two neighbouring lanes exchange two registers and store 32 contiguous
// bytes (four complex cells); the four cube columns of a patch row complete 128-byte lines in L2.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>
#include <utility>

#include "device_common.h"
#include "nufft_mi355x.h"
#include "patch_kernels.h"
#include "tile_kernels.h"

namespace nufft {

// Accumulator registers of a patch, of the 256 + 256 a wave has at one wave per SIMD: in the accumulation registers
// (AGPRs) — not all 256: the register allocator needs slack there, or it parks accumulators in vector registers and moves
// them around the inline-assembly matrix instructions, which is a read-after-issue hazard (scripts/lint_patch32_isa.py
// fails the build on it) — and in vector registers next to the working set.
#ifndef NUFFT_PATCH32_OCC
#define NUFFT_PATCH32_OCC 1             // waves per SIMD the kernel is compiled for (2: A/B builds, half the registers and LDS each)
#endif
#ifndef NUFFT_PATCH32_AGPRS
#define NUFFT_PATCH32_AGPRS 224
#endif
#ifndef NUFFT_PATCH32_VGPRS
#define NUFFT_PATCH32_VGPRS 96
#endif

template <int M>
struct Patch32Cfg {
    static constexpr int L = 2 * M;
    static constexpr int CLO = floor_div4(1 - M);      // cubes a stencil reaches relative to the bin of its point
    static constexpr int CHI = floor_div4(3 + M);
    static constexpr int NCB = CHI - CLO + 1;
    static constexpr int NOB = NCB / 2 + 1;            // octets that cover NCB consecutive cube layers, whatever their parity
    static constexpr int PBX = 4;
    static constexpr int NCA_MAX = NUFFT_PATCH32_AGPRS / (4 * NOB), NCV_MAX = NUFFT_PATCH32_VGPRS / (4 * NOB);
    static constexpr int rows() {
        int r = (NCA_MAX + NCV_MAX) / PBX;
        return r < 1 ? 1 : (r > 8 ? 8 : r);
    }
    static constexpr int PBY = rows();
    static constexpr int NRB = PBY + NCB - 1;           // rows of bins visited per bin layer
    // zeros in front of / behind the 2M window values of a staged row: dimensions 1 and 2 are read per cube ...
    static constexpr int PADB = 4 - M - 4 * CLO;
    static constexpr int PADA = 4 * CHI + 3 - M;
    static constexpr int LW = PADB + L + PADA;
    // dimension 1 is read at (cube column of the patch, lane) - (stencil start): every patch-relative position a visited
    // point can produce lies inside the padding, so the PBX reads of a lane are plain offsets from one address
    // (pairs of columns by ds_read2_b32) — no clamping arithmetic
    static constexpr int PADXB = 4 * (PBX - CLO) - M;
    static constexpr int PADXA = 4 * PBX + 4 * CHI - M - 1;
    static constexpr int LWX = PADXB + L + PADXA;
    // ... dimension 3 per octet, whose first cube may be the one below the stencil's first cube
    static constexpr int PADBZ = PADB + 4;
    static constexpr int TMAXZ = 4 * CLO + M - 1 + 8 * NOB - 1;        // largest window index an octet read can reach
    static constexpr int PADAZ = TMAXZ - (L - 1) > 0 ? TMAXZ - (L - 1) : 0;
    static constexpr int LWZ = PADBZ + L + PADAZ;
    static constexpr int G = next_pow2(L);              // lanes per point during window evaluation
    static constexpr int PPW = kWave / G;
    static constexpr int WX = 0, WY = LWX * 4, WZ = (LWX + LW) * 4;     // byte offsets of the three rows of a staged point
    static constexpr int MT = round_up((LWX + LW + LWZ) * 4, 16);       // meta data {offx, offy, offz, rbx} + value (re, im)
    // stride between staged points: an odd multiple of 8 banks, so that the rows the four points of a K-batch read at the
    // same offsets fall on disjoint banks
    static constexpr int pstride() {
        int s = round_up(MT + 32, 32);
        while ((s / 4) % 32 != 8 && (s / 4) % 32 != 24) s += 32;
        return s;
    }
    static constexpr int PSTRIDE = pstride();
    static constexpr int NPOLY = M + 4;
    static constexpr int table_bytes() { return round_up(3 * NPOLY * L * 4, 16); }
    static constexpr int STAGE_PT = 16;                 // staged cell fractions (3 floats + pad)
    // points per chunk: one wave per SIMD = one workgroup of four waves per CU with 160 KiB
    static constexpr int chunk_points() {
        const int budget = (160 * 1024 / NUFFT_PATCH32_OCC - 512 - table_bytes()) / kPatchWaves;
        for (int ch = 64; ch > 16; ch -= 8)
            if ((ch + 1) * PSTRIDE + ch * STAGE_PT + 32 <= budget) return ch;
        return 16;
    }
    static constexpr int CH = chunk_points();
    static constexpr int WBYTES = round_up((CH + 1) * PSTRIDE, 16);     // + the all-zero point
    static constexpr int WAVE_BYTES = WBYTES + round_up(CH * STAGE_PT, 16);
    static constexpr int lds_bytes() { return table_bytes() + kPatchWaves * WAVE_BYTES; }
    // Accumulator placement, explicit (the matrix instructions are issued from inline assembly, see mfma_acc): the first
    // NCA cube columns live in the 256 accumulation registers (AGPRs), the remaining NCV in vector registers next to the
    // working set.  Left to the register allocator, the builtin form shuttled a fifth of the accumulators between the two
    // files around every instruction (v_accvgpr_write x4, MFMA, s_nop 8, v_accvgpr_read x4) and spilled to scratch.
    static constexpr int NCOL = PBX * PBY;
    static constexpr int NCA = NCA_MAX < NCOL ? NCA_MAX : NCOL;
    static constexpr int NCV = NCOL - NCA;
    static_assert(PADB >= 1 && PADA >= 1 && PADBZ >= 1 && PADXB >= 0 && PADXA >= 0, "padding");
    static_assert(NCV <= NCV_MAX, "vector-register accumulators leave room for the working set");
};

typedef float v4f __attribute__((ext_vector_type(4)));

// c += A B on v_mfma_f32_16x16x4_f32 with the accumulator pinned to one register file (AG: accumulation registers).
// Inline assembly: the compiler's hazard recogniser does not see these as matrix instructions, so every other consumer of
// an accumulator is ordered behind acc_fence() + acc_touch() (retire), the operand hazard is padded by hand (below), and
// nothing but these statements touches the accumulators inside the K-batch loops (checked on the ISA by scripts/lint_patch32_isa.py at build time).  Two matrix
// instructions on the same accumulator are a whole K-batch iteration (> 150 cycles) apart.
template <bool AG>
__device__ __forceinline__ void mfma_acc(v4f& c, float a, float b) {
#if defined(NUFFT_PATCH32_BUILTIN)      // debugging: the compiler's own matrix instruction (and hazard handling)
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    return;
#endif
    // s_nop 1: a matrix instruction must not read a vector register in the two wait states after a VALU instruction
    // wrote it (the compiler pads its own v_mfma the same way; found as stale A operands: w1 instead of w1 w2).  Behind
    // another matrix instruction the no-op is hidden by the wait for the pipe.
    if constexpr (AG) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// every matrix instruction issued so far has written its result (8 passes of 4 cycles + write-back)
__device__ __forceinline__ void acc_fence() { asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory"); }
// orders the compiler's own reads / copies of c behind the preceding acc_fence()
template <bool AG>
__device__ __forceinline__ void acc_touch(v4f& c) {
    if constexpr (AG) asm volatile("" : "+a"(c));
    else asm volatile("" : "+v"(c));
}

typedef float v2f32 __attribute__((ext_vector_type(2)));
// two dwords at addr + 4 O0 and addr + 4 O1 into a register pair
template <int O0, int O1>
__device__ __forceinline__ void lds_read2_b32(v2f32& dst, uint32_t addr) {
    static_assert(O0 >= 0 && O0 < 256 && O1 >= 0 && O1 < 256, "ds_read2 offsets");
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(O0), "n"(O1));
}
// N values at a stride of S dwords: pairs by ds_read2_b32 into p[(N + 1) / 2]
template <int N, int S, int I = 0>
__device__ __forceinline__ void lds_read_strided(v2f32 (&p)[(N + 1) / 2], uint32_t addr) {
    if constexpr (I + 1 < N) {
        lds_read2_b32<I * S, (I + 1) * S>(p[I / 2], addr);
        lds_read_strided<N, S, I + 2>(p, addr);
    } else if constexpr (I < N) {
        lds_read2_b32<I * S, I * S>(p[I / 2], addr);       // odd count: the last value twice (no copy out of a register in flight)
    }
}

template <int M, bool OTHERK>
__global__ __launch_bounds__(kPatchWaves * kWave, NUFFT_PATCH32_OCC) void spread_patch32_kernel(PatchArgs<float> a) {
    using T = float;
    using P = Patch32Cfg<M>;
    constexpr int L = P::L, CLO = P::CLO, CHI = P::CHI, NCB = P::NCB, NOB = P::NOB, PBX = P::PBX, PBY = P::PBY, NRB = P::NRB;
    constexpr int PADB = P::PADB, PADBZ = P::PADBZ, PADXB = P::PADXB, LW = P::LW, CH = P::CH, PSTRIDE = P::PSTRIDE;
    constexpr int WX = P::WX, WY = P::WY, WZ = P::WZ, MT = P::MT;
    using WE = WindowEval<T, 1, 3, M, P::G, OTHERK>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x / kWave;
    const Geom& g = a.t.g;
    const PatchGeom& pg = a.pg;

    if (a.enabled && *a.enabled == 0u) return;      // this point set goes to the LDS-tile kernel (set_points: balance.hip)
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

    unsigned char* wmem = smem + P::table_bytes() + wave * P::WAVE_BYTES;                // staged points
    unsigned char* stage = wmem + P::WBYTES;
    const uint32_t wbase = (uint32_t)(uintptr_t)wmem;
    for (int o = lane * 16; o < P::WBYTES; o += kWave * 16) *reinterpret_cast<uint4*>(wmem + o) = make_uint4(0, 0, 0, 0);

    // ---- accumulators: [ring slot = octet][cube row][cube column], 256 cells x 4 registers each ----
    // (column = y * PBX + x; columns < NCA in accumulation registers, the others in vector registers)
    constexpr int NCA = P::NCA, NCV = P::NCV;
    v4f accA[NOB][NCA], accV[NOB][NCV > 0 ? NCV : 1];
#pragma unroll
    for (int s = 0; s < NOB; ++s) {
#pragma unroll
        for (int c = 0; c < NCA; ++c) accA[s][c] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < (NCV > 0 ? NCV : 1); ++c) accV[s][c] = v4f{0.f, 0.f, 0.f, 0.f};
    }

    const int grp = lane / P::G, q = lane % P::G;                      // window evaluation roles (group mapping)
    const int mk = lane >> 4, mb = (lane >> 2) & 3, mi = lane & 3;     // matrix roles of the A operand: point, y, x
    const int bzl = (lane & 15) >> 1, bcl = lane & 1;                  // ... of the B operand: plane of the octet, component

    const PointRec<T, 3>* sorted = static_cast<const PointRec<T, 3>*>(a.t.sorted);
    const T* vs = a.vsorted[comp_id];
    T* grid = a.t.grid[comp_id];

    // ---- runs of the sorted array (as in spread_patch_kernel): lane t < 4 NRB holds bound (row t >> 2, piece, end) ----
    const int gx0 = bx0 - CHI, gx1 = bx0 + ncx - 1 - CLO;
    auto load_bounds = [&](int bz) __attribute__((always_inline)) -> uint32_t {
        const int r = lane >> 2, piece = (lane >> 1) & 1, isend = lane & 1;
        const int rb = r - CHI;
        uint32_t val = 0;
        if (lane < 4 * NRB && rb <= ncy - 1 - CLO) {
            int lo, hi;
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
    float2 pf_v;
    auto issue_prefetch = [&](const Cursor& c) __attribute__((always_inline)) {
        const uint32_t n = min((uint32_t)CH, c.pe - c.p);
        const uint32_t pp = c.p + min((uint32_t)lane, n - 1);
        pf_rec = sorted[pp];
        pf_v = reinterpret_cast<const float2*>(vs)[pp];
    };
    // lane l < CH = point l of the chunk: cell fraction staged for the window evaluation, meta data {-4 sx, byte offsets of
    // the first reads of dimensions 2 and 3, bin along dimension 1} + value next to the point's window rows
    auto commit_prefetch = [&](int bz) __attribute__((always_inline)) {
        if (lane < CH) {
            int cell[3];
            float4 fr;
            cell[0] = cell_of(pf_rec.r[0], g.Nover[0]); fr.x = pf_rec.r[0] - T(cell[0]);
            cell[1] = cell_of(pf_rec.r[1], g.Nover[1]); fr.y = pf_rec.r[1] - T(cell[1]);
            cell[2] = cell_of(pf_rec.r[2], g.Nover[2]); fr.z = pf_rec.r[2] - T(cell[2]);
            fr.w = 0.f;
            *reinterpret_cast<float4*>(stage + lane * P::STAGE_PT) = fr;
            int sx = cell[0] - (M - 1) - X0;                           // stencil start relative to the patch, unwrapped
            if (sx > g.Nover[0] / 2) sx -= g.Nover[0];
            if (sx < -(g.Nover[0] / 2)) sx += g.Nover[0];
            const int par = (bz + CLO) & 1;                            // first cube of the ring is the odd one of slot 0's octet
            int4 m;
            m.x = (PADXB - sx) * 4;                                    // first read of dimension 1: cube column 0, lane x = 0
            m.y = (PADB + M - 1 - (cell[1] & 3) + 4 * CLO) * 4;        // cube offset CLO, lane row 0
            m.z = (PADBZ + M - 1 - (cell[2] & 3) + 4 * CLO - 4 * par) * 4;   // octet slot 0, plane 0
            m.w = (sx + (M - 1)) >> 2;                                 // bin of the point relative to the patch
            unsigned char* pw = wmem + lane * PSTRIDE + MT;
            *reinterpret_cast<int4*>(pw) = m;
            *reinterpret_cast<float2*>(pw + 16) = pf_v;
        }
    };

    // ---- bin layer bz is finished: cube layer bz + CLO is complete; when it is the odd layer of its octet the octet
    //      leaves (slot 0 of the ring) and the ring shifts ----
    auto retire = [&](int bz) __attribute__((always_inline)) {
        const int cz = bz + CLO;
        if ((cz & 1) == 0) return;
        acc_fence();
#pragma unroll
        for (int s = 0; s < NOB; ++s) {
#pragma unroll
            for (int c = 0; c < NCA; ++c) acc_touch<true>(accA[s][c]);
#pragma unroll
            for (int c = 0; c < NCV; ++c) acc_touch<false>(accV[s][c]);
        }
        if (cz >= z0 && cz < z1) {
            const int64_t gz = (int64_t)(cz - 1) * 4 + bzl;           // plane of this lane
            const int yl = lane >> 4;
#pragma unroll
            for (int y = 0; y < PBY; ++y) {
                if (y < ncy) {
                    const int64_t rowbase = ((gz * g.Nover[1] + (int64_t)(by0 + y) * 4 + yl) * g.Nover[0] + X0) * 2;
#pragma unroll
                    for (int x = 0; x < PBX; ++x) {
                        if (x < ncx) {
                            const int col = y * PBX + x;
                            const v4f v = col < NCA ? accA[0][col < NCA ? col : 0] : accV[0][col >= NCA ? col - NCA : 0];
                            // lane pair (component 0, 1): exchange two registers, then each stores two complex cells
                            const float s0 = bcl ? v[0] : v[2], s1 = bcl ? v[1] : v[3];
                            const float t0 = dpp_move<0xB1>(s0), t1 = dpp_move<0xB1>(s1);
                            const float4 out = bcl ? make_float4(t0, v[2], t1, v[3]) : make_float4(v[0], t0, v[1], t1);
                            *reinterpret_cast<float4*>(grid + rowbase + (4 * x + 2 * bcl) * 2) = out;
                        }
                        __builtin_amdgcn_sched_barrier(0);      // one accumulator at a time through the vector registers
                    }
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCA; ++c) {
#pragma unroll
            for (int s = 0; s + 1 < NOB; ++s) accA[s][c] = accA[s + 1][c];
            accA[NOB - 1][c] = v4f{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_sched_barrier(0);                  // (the scheduler would otherwise gather all copies' temporaries)
        }
#pragma unroll
        for (int c = 0; c < NCV; ++c) {
#pragma unroll
            for (int s = 0; s + 1 < NOB; ++s) accV[s][c] = accV[s + 1][c];
            accV[NOB - 1][c] = v4f{0.f, 0.f, 0.f, 0.f};
        }
    };

    // ---- window evaluation of the n <= CH staged points of a chunk (group mapping), two passes of PPW points at a time:
    //      at one wave per SIMD nothing else hides the latency of a pass (LDS read -> Horner chain -> LDS writes), and the
    //      polynomial runs both passes' chains as the two halves of v_pk_fma_f32 ----
    typedef float v2f __attribute__((ext_vector_type(2)));
    const bool poly_eval = a.t.evalmode != NUFFT_EVAL_DIRECT;
    auto eval_chunk = [&](int n) __attribute__((always_inline)) {
        WE we;              // (hoisting this lane's coefficients out of the chunk loop: the allocator then shuttles accumulators, lint fails)
        we.init(a.t, q, ctab);
        wave_lds_fence();
        auto store = [&](int pt, const T (&v)[WE::NSLOT]) __attribute__((always_inline)) {
            unsigned char* pw = wmem + pt * PSTRIDE;
            if (pt < n) {
#pragma unroll
                for (int sl = 0; sl < WE::NSLOT; ++sl)
                    if (we.has[sl]) {
                        const int off = we.dsel[sl] == 2 ? WZ + (PADBZ + we.jsel[sl]) * 4
                                      : (we.dsel[sl] == 1 ? WY + (PADB + we.jsel[sl]) * 4 : WX + (PADXB + we.jsel[sl]) * 4);
                        *reinterpret_cast<float*>(pw + off) = v[sl];
                    }
            }
        };
#pragma unroll 1
        for (int t0 = 0; t0 < n; t0 += 2 * P::PPW) {
            const int ptA = t0 + grp, ptB = ptA + P::PPW;
            const float4 frA = *reinterpret_cast<const float4*>(stage + min(ptA, n - 1) * P::STAGE_PT);
            const float4 frB = *reinterpret_cast<const float4*>(stage + min(ptB, n - 1) * P::STAGE_PT);
            T vA[WE::NSLOT], vB[WE::NSLOT];
            if (!OTHERK && poly_eval) {
                // Horner (src/Kernels/piecewise_polynomial.jl:84-92) for both points at once: same operations, same order
                const v2f x0 = {2.f * frA.x - 1.f, 2.f * frB.x - 1.f}, x1 = {2.f * frA.y - 1.f, 2.f * frB.y - 1.f},
                          x2 = {2.f * frA.z - 1.f, 2.f * frB.z - 1.f};
                // (the NSLOT independent chains advance together: a dependent v_pk_fma_f32 costs an extra wait state)
                v2f xx[WE::NSLOT], val[WE::NSLOT];
#pragma unroll
                for (int sl = 0; sl < WE::NSLOT; ++sl) {
                    xx[sl] = we.dsel[sl] == 0 ? x0 : (we.dsel[sl] == 1 ? x1 : x2);
                    val[sl] = v2f{we.cs[sl][WE::NP - 1], we.cs[sl][WE::NP - 1]};
                }
#pragma unroll
                for (int c = WE::NP - 2; c >= 0; --c)
#pragma unroll
                    for (int sl = 0; sl < WE::NSLOT; ++sl) val[sl] = __builtin_elementwise_fma(xx[sl], val[sl], v2f{we.cs[sl][c], we.cs[sl][c]});
#pragma unroll
                for (int sl = 0; sl < WE::NSLOT; ++sl) { vA[sl] = val[sl][0]; vB[sl] = val[sl][1]; }
            } else {
                const T XA[3] = {frA.x, frA.y, frA.z}, XB[3] = {frB.x, frB.y, frB.z};
                we.eval_regs(a.t, XA, vA);
                we.eval_regs(a.t, XB, vB);
            }
            store(ptA, vA);
            store(ptB, vB);
        }
        wave_lds_fence();
    };

#if defined(NUFFT_PATCH_PROFILE)
    unsigned long long nmfma = 0, nbatchcol = 0, nbatch = 0;
#endif
#ifndef NUFFT_PATCH32_PIPE
#define NUFFT_PATCH32_PIPE 0            // 1: the travelling-work loop below (round 6 experiment: measured slower, see its header); 0: round 5's K-batch loop
#endif
#if NUFFT_PATCH32_PIPE
    // ---- EXPERIMENT (round 6, review item 4a; NUFFT_PATCH32_PIPE=1 builds it; parity-green, lint-clean, SLOWER — kept for the record).
    //      K-batches of four points with the non-matrix work of a batch moved into the gaps between its matrix instructions:
    //        column 1               | addresses of batch i + 1's operand reads and of batch i + 2's meta data, a slice per gap
    //        READS                  | the one issue site (registers an inline-assembly read has been issued into must not meet at a join)
    //        columns 0 and 3
    //        column 2               | its first row covers the reads; then the wait and batch i + 1's operands, in place: columns 0 and 1 of a
    //                               | row behind the row's last matrix instruction, columns 2 and 3 a gap later (column 2 read them at issue)
    //      The premise — an instruction behind a v_mfma issues while it runs — does not hold for v_mfma_f32_16x16x4_f32: it forms its 1024 products
    //      in 32 cycles = 64 FLOP per cycle and SIMD, exactly the rate of v_pk_fma_f32: the FP32 matrix instruction runs on the vector ALUs
    //      (as v_mfma_f64_16x16x4 does on the FP64 ones), so vector work between two of them adds its full time.
    //      Measured, C3 spread stage (profiles/round6_c3_patch32_phases.md; round 5's loop 75.0 ms, batches phase 1.15e11 wave cycles):
    //        one triangle per cube ROW, the work between the triangles, operands by v_pk_mul_f32      81.3 ms
    //        the same with v_mul_f32 (a packed Float32 instruction beside matrix instructions is dearer)   80.8 ms
    //        one triangle per column, the work in its gaps (this code)                                 82.2 ms, batches phase 1.29e11
    //      — every form pays for what it adds (scalar multiplies instead of packed ones, the second triangle, pins) and hides nothing.
    auto batches = [&](auto RBc, int n) __attribute__((always_inline)) {
        constexpr int RB = decltype(RBc)::value - CHI;                 // relative bin row: cube rows RB + CLO .. RB + CHI
        constexpr int NV = [] { int c = 0; for (int o = 0; o < NCB; ++o) c += (RB + CLO + o >= 0 && RB + CLO + o < PBY) ? 1 : 0; return c; }();
        typedef int v4i __attribute__((ext_vector_type(4)));
        v4i m;
        float vsel;                                                    // component bcl of the point's value
        v2f32 w3p[(NOB + 1) / 2], w2p[(NCB + 1) / 2], w1p[PBX / 2];      // operand values in register pairs (ds_read2_b32)
        const uint32_t vsel_off = (uint32_t)bcl * 4u;
        // what the gaps of column 1 prepare for the reads (slices of at most four vector instructions: the shadow of one matrix instruction
        // hides about five; every slice is pinned where it is written, or the optimiser sinks it across the matrix blocks to its use)
        uint32_t pbn = 0u, az = 0u, ay = 0u, ax = 0u, am = 0u, nmask = 0u;
        int pidx_n = 0, pidx_m = 0;
        float vnext = 0.f;
        constexpr int NSL = 5;
        auto pre_slice = [&](int S, int b0) __attribute__((always_inline)) {
            if (S == 0) {
                pidx_n = b0 + 4 + mk < n ? b0 + 4 + mk : CH;
                pidx_m = b0 + 8 + mk < n ? b0 + 8 + mk : CH;
                asm volatile("" : "+v"(pidx_n), "+v"(pidx_m));
            } else if (S == 1) {
                pbn = wbase + (uint32_t)(pidx_n * PSTRIDE);
                az = pbn + WZ + (uint32_t)m.z + (uint32_t)bzl * 4;    // octets: 8 planes apart
                asm volatile("" : "+v"(pbn), "+v"(az));
            } else if (S == 2) {
                ay = pbn + WY + (uint32_t)m.y + (uint32_t)mb * 4;     // cube rows: 4 cells apart
                ax = pbn + WX + (uint32_t)m.x + (uint32_t)mi * 4;     // cube columns of the patch
                vnext = vsel;
                asm volatile("" : "+v"(ay), "+v"(ax), "+v"(vnext));
            } else if (S == 3) {
                const int nvalid = max(1, min(4, n - (b0 + 4)));
                const int lo = max(__builtin_amdgcn_readlane(m.w, 0) + CLO, 0);
                const int hi = min(__builtin_amdgcn_readlane(m.w, 16 * (nvalid - 1)) + CHI, PBX - 1);
                nmask = hi >= lo ? (2u << hi) - (1u << lo) : 0u;
                asm volatile("" : "+s"(nmask));
            } else {
                am = wbase + (uint32_t)(pidx_m * PSTRIDE + MT);
                asm volatile("" : "+v"(am));
            }
        };
        auto reads = [&]() __attribute__((always_inline)) {
            lds_read_strided<NOB, 8>(w3p, az);
            lds_read_strided<NCB, 4>(w2p, ay);
            lds_read_strided<PBX, 4>(w1p, ax);
            const uint32_t amv = am + vsel_off;
            asm volatile("ds_read_b128 %0, %1" : "=v"(m) : "v"(am));
            asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(vsel) : "v"(amv));
        };
        auto wait_all = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(m), "+v"(vsel));
#pragma unroll
            for (int s = 0; s < (NOB + 1) / 2; ++s) asm volatile("" : "+v"(w3p[s]));
#pragma unroll
            for (int o = 0; o < (NCB + 1) / 2; ++o) asm volatile("" : "+v"(w2p[o]));
#pragma unroll
            for (int cx = 0; cx < PBX / 2; ++cx) asm volatile("" : "+v"(w1p[cx]));
        };
        // operands of a batch: A = w1 w2 per cube column and row, B = v w3 per octet.  One v_mul_f32 each, NOT v_pk_mul_f32: a packed Float32
        // instruction next to matrix instructions costs ~22 cycles more than two plain ones (MI355X_MICROARCH.md, "price of one filler") — the first
        // version of this loop built the operands in pairs and ran 8 % slower than round 5's (C3 spread 81.3 against 75.0 ms).
        float Af[NCB][PBX], Bf[NOB];
        auto build_A = [&](int o, int pr) __attribute__((always_inline)) {
            // (inline assembly: pinned where it is written, and the vectoriser cannot pack it with a neighbour)
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(Af[o][2 * pr]) : "v"(w1p[pr][0]), "v"(w2p[o / 2][o % 2]));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(Af[o][2 * pr + 1]) : "v"(w1p[pr][1]), "v"(w2p[o / 2][o % 2]));
        };
        auto build_B = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int s = 0; s < NOB; ++s) {
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(Bf[s]) : "v"(w3p[s / 2][s % 2]), "v"(vnext));
            }
        };
        auto row_valid = [](int o) __attribute__((always_inline)) -> bool { return RB + CLO + o >= 0 && RB + CLO + o < PBY; };
        auto mfma_one = [&](int cx, int o, int s) __attribute__((always_inline)) {
            constexpr int dummy = 0;
            const int col = (RB + CLO + o) * PBX + cx;
            if (col < NCA) mfma_acc<true>(accA[s][col < NCA ? col : dummy], Af[o][cx], Bf[s]);
            else mfma_acc<false>(accV[s][col >= NCA ? col - NCA : dummy], Af[o][cx], Bf[s]);
        };
        // rows [first, last) (ordinals among the valid rows) of a cube column, nothing in the gaps
        auto column_rows = [&](int cx, int first, int last) __attribute__((always_inline)) {
            int ord = 0;
#pragma unroll
            for (int o = 0; o < NCB; ++o) {
                if (!row_valid(o)) continue;
                if (ord >= first && ord < last) {
#pragma unroll
                    for (int s = 0; s < NOB; ++s) mfma_one(cx, o, s);
                }
                ++ord;
            }
        };
#if defined(NUFFT_PATCH_PROFILE)
#define NUFFT_P32_COUNT(rows) do { nmfma += (unsigned long long)((rows) * NOB); nbatchcol += 1; } while (0)
#else
#define NUFFT_P32_COUNT(rows) do { } while (0)
#endif
        // ---- prologue: meta data of batch 0; operand reads of batch 0 and meta data of batch 1; operands of batch 0 ----
        {
            const int pidx = mk < n ? mk : CH;
            am = wbase + (uint32_t)(pidx * PSTRIDE + MT);
            const uint32_t amv = am + vsel_off;
            asm volatile("ds_read_b128 %0, %1" : "=v"(m) : "v"(am));
            asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(vsel) : "v"(amv));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(m), "+v"(vsel));
        }
#pragma unroll
        for (int S = 0; S < NSL; ++S) pre_slice(S, -4);
        reads();
        wait_all();
#pragma unroll
        for (int o = 0; o < NCB; ++o)
            if (row_valid(o)) { build_A(o, 0); build_A(o, 1); }
        build_B();
        uint32_t mask = nmask;
#pragma unroll 1
        for (int b0 = 0; b0 < n; b0 += 4) {
#if defined(NUFFT_PATCH_PROFILE)
            nbatch += 1;
#endif
            // (a column's matrix instructions sit in ONE triangle — `if (column in the mask) { ... }` — with the travelling work in its gaps, at
            // most five instructions per gap: a lone wave issues an instruction of any kind every ~4 cycles and a matrix instruction runs 32.  The batch
            // that skips the column does the same work in a second triangle — not an else arm: around a diamond the register allocator copied
            // every accumulator.  The conditions are opaque, or the compiler threads the two triangles back into one diamond.  Measured on the
            // way (C3 spread, round 5's loop 75.0 ms): one triangle per ROW with the work between the triangles 80.8 ms — three scalar instructions
            // per triangle and the work of a row in one gap overflow the gaps.)
            auto has_col = [&](uint32_t bit) __attribute__((always_inline)) -> bool {
                uint32_t mm = mask;
                asm volatile("" : "+s"(mm));
                return (mm & bit) != 0u;
            };
            // ---- column 1: the addresses of the next reads in its gaps ----
            if (has_col(2u)) {
                NUFFT_P32_COUNT(NV);
                constexpr int K1 = NV * NOB;
                int gap = 0;
#pragma unroll
                for (int o = 0; o < NCB; ++o) {
                    if (!row_valid(o)) continue;
#pragma unroll
                    for (int s = 0; s < NOB; ++s) {
                        mfma_one(1, o, s);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int S = 0; S < NSL; ++S)
                            if ((K1 >= NSL ? S * (K1 / NSL) : S * K1 / NSL) == gap) pre_slice(S, b0);
                        __builtin_amdgcn_sched_barrier(0);
                        ++gap;
                    }
                }
            }
            if (!has_col(2u)) {
#pragma unroll
                for (int S = 0; S < NSL; ++S) pre_slice(S, b0);
            }
            reads();
            if (mask & 1u) { NUFFT_P32_COUNT(NV); column_rows(0, 0, NV); }
            if (mask & 8u) { NUFFT_P32_COUNT(NV); column_rows(3, 0, NV); }
            // ---- column 2: the reads land behind its first row; then the operands of the next batch in its gaps — columns 0 and 1 of a row
            //      behind the row's last matrix instruction, columns 2 and 3 a gap later (column 2 read them at issue) ----
            if (has_col(4u)) {
                NUFFT_P32_COUNT(NV);
                int prev = -1;
#pragma unroll
                for (int o = 0; o < NCB; ++o) {
                    if (!row_valid(o)) continue;
#pragma unroll
                    for (int s = 0; s < NOB; ++s) {
                        mfma_one(2, o, s);
                        __builtin_amdgcn_sched_barrier(0);
                        if (s == 0 && prev >= 0) build_A(prev, 1);
                        if (s == NOB - 1) {
                            if (prev < 0) wait_all();
                            build_A(o, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    prev = o;
                }
                if (prev >= 0) build_A(prev, 1);
                build_B();
            }
            if (!has_col(4u)) {
                wait_all();
#pragma unroll
                for (int o = 0; o < NCB; ++o)
                    if (row_valid(o)) { build_A(o, 0); build_A(o, 1); }
                build_B();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (no path without a wait, whatever the opaque conditions: the build's lint follows the control flow)
            mask = nmask;
        }
    };
#else
    // ---- K-batches of four points (k = lane >> 4; the all-zero point pads the last one), software-pipelined as in
    //      spread_patch_kernel: the LDS reads of batch i + 1 (operands) and i + 2 (meta data) fly while batch i's MFMAs issue
    auto batches = [&](auto RBc, int n) __attribute__((always_inline)) {
        constexpr int RB = decltype(RBc)::value - CHI;                 // relative bin row: cube rows RB + CLO .. RB + CHI
        typedef int v4i __attribute__((ext_vector_type(4)));
        v4i m;
        float vsel;                                                    // component bcl of the point's value
        v2f32 w3p[(NOB + 1) / 2], w2p[(NCB + 1) / 2], w1p[PBX / 2];      // operand values in register pairs (ds_read2_b32)
        uint32_t cxmask = 0u;
        const uint32_t vsel_off = (uint32_t)bcl * 4u;
        auto issue_meta = [&](int b0) __attribute__((always_inline)) {
            const int pidx = b0 + mk < n ? b0 + mk : CH;
            const uint32_t ad = wbase + (uint32_t)(pidx * PSTRIDE + MT);
            const uint32_t adv = ad + vsel_off;
            asm volatile("ds_read_b128 %0, %1" : "=v"(m) : "v"(ad));
            asm volatile("ds_read_b32 %0, %1 offset:16" : "=v"(vsel) : "v"(adv));
        };
        float vcur = 0.f;
        auto issue_ops = [&](int b0) __attribute__((always_inline)) {
            const int pidx = b0 + mk < n ? b0 + mk : CH;
            const uint32_t pb = wbase + (uint32_t)(pidx * PSTRIDE);
            const int nvalid = max(1, min(4, n - b0));
            {
                const int lo = max(__builtin_amdgcn_readlane(m.w, 0) + CLO, 0);
                const int hi = min(__builtin_amdgcn_readlane(m.w, 16 * (nvalid - 1)) + CHI, PBX - 1);
                cxmask = hi >= lo ? (2u << hi) - (1u << lo) : 0u;
            }
            vcur = vsel;
            lds_read_strided<NOB, 8>(w3p, pb + WZ + (uint32_t)m.z + (uint32_t)bzl * 4);       // octets: 8 planes apart
            lds_read_strided<NCB, 4>(w2p, pb + WY + (uint32_t)m.y + (uint32_t)mb * 4);        // cube rows: 4 cells apart
            lds_read_strided<PBX, 4>(w1p, pb + WX + (uint32_t)m.x + (uint32_t)mi * 4);        // cube columns of the patch
        };
        auto wait_all = [&]() __attribute__((always_inline)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(m), "+v"(vsel));
#pragma unroll
            for (int s = 0; s < (NOB + 1) / 2; ++s) asm volatile("" : "+v"(w3p[s]));
#pragma unroll
            for (int o = 0; o < (NCB + 1) / 2; ++o) asm volatile("" : "+v"(w2p[o]));
#pragma unroll
            for (int cx = 0; cx < PBX / 2; ++cx) asm volatile("" : "+v"(w1p[cx]));
        };
        issue_meta(0);
        wait_all();
        issue_ops(0);
        issue_meta(4);
        wait_all();
#pragma unroll 1
        for (int b0 = 0; b0 < n; b0 += 4) {
#if defined(NUFFT_PATCH_PROFILE)
            nbatch += 1;
#endif
            // operands of this batch: A = w1 w2 per cube column and row (pairs of columns: v_pk_mul_f32), B = v w3 per octet
            v2f32 Ap[NCB][PBX / 2], Bp[(NOB + 1) / 2];
#pragma unroll
            for (int s = 0; s < (NOB + 1) / 2; ++s) Bp[s] = w3p[s] * vcur;
#pragma unroll
            for (int o = 0; o < NCB; ++o)
#pragma unroll
                for (int cx = 0; cx < PBX / 2; ++cx)
                    Ap[o][cx] = (RB + CLO + o >= 0 && RB + CLO + o < PBY) ? w1p[cx] * w2p[o / 2][o % 2] : v2f32{0.f, 0.f};
            const uint32_t mask = cxmask;
            // The operand reads of the next batch (and the meta data of the one after) are issued behind the matrix
            // instructions of cube column 1 — four fifths of the matrix work belongs to batches that touch it — so that
            // their address arithmetic, the LDS issue and the scalar mask set-up run while the matrix pipe works: with one
            // wave per SIMD nothing else fills it, and everything in front of the first matrix instruction is serial time.
            // ONE issue site: registers an inline-assembly read has been issued into must not meet at a control-flow join,
            // where the compiler would copy them before the data has arrived.
            auto column = [&](auto CXc) __attribute__((always_inline)) {
                constexpr int cx = decltype(CXc)::value;
                if (mask & (1u << cx)) {
#if defined(NUFFT_PATCH_PROFILE)
                    {
                        int rows = 0;
#pragma unroll
                        for (int o = 0; o < NCB; ++o) rows += (RB + CLO + o >= 0 && RB + CLO + o < PBY) ? 1 : 0;
                        nmfma += (unsigned long long)(rows * NOB);
                        nbatchcol += 1;
                    }
#endif
#pragma unroll
                    for (int o = 0; o < NCB; ++o) {
                        const int cy = RB + CLO + o;
                        if (cy >= 0 && cy < PBY) {
#pragma unroll
                            for (int s = 0; s < NOB; ++s) {
                                constexpr int dummy = 0;
                                const int col = cy * PBX + cx;
                                if (col < NCA) mfma_acc<true>(accA[s][col < NCA ? col : dummy], Ap[o][cx / 2][cx % 2], Bp[s / 2][s % 2]);
                                else mfma_acc<false>(accV[s][col >= NCA ? col - NCA : dummy], Ap[o][cx / 2][cx % 2], Bp[s / 2][s % 2]);
                            }
                        }
                    }
                }
            };
            column(std::integral_constant<int, 1>{});
            issue_ops(b0 + 4);
            issue_meta(b0 + 8);
            column(std::integral_constant<int, 2>{});
            column(std::integral_constant<int, 0>{});
            column(std::integral_constant<int, 3>{});
            wait_all();
        }
    };

#endif

#if defined(NUFFT_PATCH_PROFILE)
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    tph[5] = 0;
#define NUFFT_PH32(i) do { const unsigned long long tn = __builtin_readcyclecounter(); tph[i] += tn - tlast; tlast = tn; } while (0)
#else
#define NUFFT_PH32(i) do { } while (0)
#endif
    // ---- main loop over the chunks of the segment ----
    Cursor nxt{bz_first, -1, 0u, 0u};
    bool has = advance(nxt);
    if (has) issue_prefetch(nxt);
    int bz_done = bz_first;
    while (has) {
        const Cursor cur = nxt;
        NUFFT_PH32(0);
        while (bz_done < cur.bz) { retire(bz_done); ++bz_done; }
        NUFFT_PH32(1);
        wave_lds_fence();
        commit_prefetch(cur.bz);
        NUFFT_PH32(2);
        has = advance(nxt);
        if (has) issue_prefetch(nxt);
        NUFFT_PH32(0);
        const int n = (int)min((uint32_t)CH, cur.pe - cur.p);
        eval_chunk(n);
        NUFFT_PH32(3);
        dispatch_row<0, NRB>(cur.u >> 1, [&](auto Rc) __attribute__((always_inline)) { batches(Rc, n); });
        NUFFT_PH32(4);
    }
    while (bz_done <= bz_last) { retire(bz_done); ++bz_done; }
    NUFFT_PH32(1);
#if defined(NUFFT_PATCH_PROFILE)
    if (lane == 0 && a.prof) {
        for (int i = 0; i < 5; ++i) atomicAdd(a.prof + i, tph[i]);
        atomicAdd(a.prof + 5, nmfma);
        atomicAdd(a.prof + 6, nbatch);
        atomicAdd(a.prof + 7, nbatchcol);
    }
#endif
}

}  // namespace nufft
