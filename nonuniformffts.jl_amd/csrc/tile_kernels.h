// Type-1 spreading and type-2 interpolation on LDS tiles (gfx950, wave64).
//
// Replaces the reference's shared-memory kernels
//   spread_from_points_shmem_kernel!    src/spreading/gpu.jl:237-377 (+ :381-434)
//   interpolate_to_points_shmem_kernel! src/interpolation/gpu.jl:211-328 (+ :331-395)
// with an MI355X-first design (see DESIGN.md §4.1 for the measurements behind each choice):
//
//   * Points arrive sorted by fine bins (binsort.hip) as aligned records {r_1..r_D, idx}; a tile is a
//     box of bins, so its points are a few contiguous runs of the sorted array, cut into ~64 work items
//     that the waves of the workgroup pull from a shared LDS counter.  Heavy tiles of a non-uniform point
//     set are shared by several workgroups (slices, balance.hip).
//   * Lane mapping: G = nextpow2(ncomp * 2M) consecutive lanes own one point (one lane per
//     (component, j1) of the stencil's first dimension); a wave works on 64 / G points at once.  The
//     D*2M window values of a point are evaluated once by its G lanes (direct form, or the piecewise
//     polynomial with the lane's coefficients held in registers).
//   * Spreading is OUTPUT-DRIVEN: only the tile interior lives in LDS (Float64 accumulation with
//     native ds_add_f64); the workgroup visits every point whose stencil touches the tile, clips the
//     stencil to the tile, and finally stores the tile with plain coalesced stores.  There are no
//     global atomics (memory-side float atomics run at 1.2 TB/s on MI355X, 3-5x below plain stores) and
//     the grid needs no zero fill: every cell is written exactly once.  The window values go through a
//     wave-private LDS strip; accumulation walks the chunk's points one at a time with the 64 lanes on
//     the (component, j1, j2) face of the stencil and a loop over j3 (dimension-3 values by DPP row
//     broadcast).
//   * Interpolation loads the padded tile (interior + 2M-1 halo) once, visits each point exactly once,
//     gathers with the group mapping (window values exchanged by DPP broadcasts for real data) and
//     reduces over the G lanes with DPP / permlane-swap butterflies.
//   * Default configurations use compile-time tile shapes (fixed_spread_tile / fixed_interp_tile): LDS
//     strides become immediates and the hot LDS instructions are issued from inline assembly.
#pragma once

#include <hip/hip_runtime.h>

#include <utility>

#include "device_common.h"
#include "nufft_mi355x.h"

#ifndef NUFFT_W3_READLANE
#define NUFFT_W3_READLANE 1
#endif
#ifndef NUFFT_SPREAD_ASM_STRIP
#define NUFFT_SPREAD_ASM_STRIP 1    // compile-time-tile spreading: the strip reads of two points issued together (inline asm)
#endif
#ifndef NUFFT_INTERP_ASM_READS
#define NUFFT_INTERP_ASM_READS 1    // compile-time-tile interpolation: hand-scheduled LDS reads with immediate offsets
#endif
#ifndef NUFFT_INTERP_REGW
#define NUFFT_INTERP_REGW 1         // interpolation: window values exchanged by DPP broadcasts instead of an LDS strip
#endif

namespace nufft {

template <typename T>
struct TileArgs {
    Geom g;
    const void* sorted;
    const uint32_t* offsets;              // [nbins + 1]
    const T* coefs;                       // [D][npoly][2M], scaled by the window normalisation
    T beta[3];
    T bop[3];                             // β/π times the power-of-two window normalisation
    T* grid[kMaxCompPerLaunch];           // component grids (as arrays of reals)
    const T* vin[kMaxCompPerLaunch];      // spread: values (as reals; complex = interleaved)
    T* vout[kMaxCompPerLaunch];           // interp
    T prefactor;
    const T* weights;                     // optional real weight per point (nonuniform callback menu), or null
    const uint2* desc;                    // slot -> (tile, slice << 16 | slices of the tile), balance.hip
    const uint32_t* desc_total;           // slots in use (the launch grid may be larger)
    int xcd_chunk;                        // slots per XCD chunk (0: one contiguous range per XCD)
    const uint32_t* march_flag;           // interpolation: device flag of set_points (balance.hip), 1 = the z-marching kernel
                                          // (march_kernels.h) serves this point set and interp_tile_kernel returns; null: no ring
    int evalmode;
    int kernel;                           // NUFFT_KERNEL_*; beta / bop per kernel: see WindowEval
};

// The window fields of TileArgs with the evaluation mode fixed at compile time (kernels instantiated per mode: the other
// mode's code and registers disappear).
template <typename T, int EVALMODE>
struct EvalArgs {
    static constexpr int evalmode = EVALMODE;
    int kernel;
    const T* coefs;
    T beta[3], bop[3];
    __device__ __forceinline__ explicit EvalArgs(const TileArgs<T>& a) : kernel(a.kernel), coefs(a.coefs) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { beta[d] = a.beta[d]; bop[d] = a.bop[d]; }
    }
};

template <int NC, int M>
struct Grp {
    static constexpr int L = 2 * M;
    static constexpr int W1 = NC * L;                   // reals of the stencil along dimension 1
    static constexpr int G = lanes_per_point(NC, M);    // lanes per point
    static constexpr int PPW = kWave / G;               // points per wave at once
};

// Window evaluation of one wave: the G lanes of a group evaluate the D*2M values of their point and
// exchange them through the wave's LDS strip.  Coefficients of the piecewise polynomial stay in
// registers (the lane's (dimension, j) role per slot never changes).
//
// OTHERK = false (the hot instantiations): BackwardsKaiserBessel Direct (sinh form) and every
// polynomial evaluation (FastApproximation of both Kaiser-Bessel kernels), no per-point weights.
// OTHERK = true is the general variant: it adds the remaining kernel x mode combinations of the
// reference and the per-point weights of the callback menu (the extra code costs the hot kernels
// 4-12 % when it is merely present, measured) —
//   KaiserBessel Direct   I0(beta sqrt(1 - y^2))                 src/Kernels/kaiser_bessel.jl:197-210
//   Gaussian Direct       exp(-((M-1-j+X) dx)^2 / tau)           src/Kernels/gaussian.jl:141-153
//   Gaussian Fast         fast Gaussian gridding a cs[m] b^(+-m) src/Kernels/gaussian.jl:125-139,155-192
//   BSpline (both modes)  order-2M recursion                     src/Kernels/bspline.jl:99-119,140-193
// Per-dimension parameters in TileArgs: BKB beta, (beta/pi) 2^k; KB beta, 2^k; Gaussian dx, tau.
template <typename T>
__device__ __forceinline__ T dev_bessel_i0(T x) {
    // power series, all terms positive (stands in for Bessels.besseli0 / the branch-free version of
    // ext/NonuniformFFTsAMDGPUExt.jl:32-44)
    const T q = T(0.25) * x * x;
    const T eps = sizeof(T) == 8 ? T(1e-17) : T(1e-9);
    T term = T(1), sum = T(1);
    for (int k = 1; k < 400; ++k) {
        term *= q / (T(k) * T(k));
        sum += term;
        if (term < eps * sum) break;
    }
    return sum;
}

template <typename T, int M>
__device__ __forceinline__ T dev_bspline_value(T x, int jsel) {
    constexpr int K = 2 * M;
    T bs[K];
    bs[0] = T(1);
#pragma unroll
    for (int q = 2; q <= K; ++q) {
        const T alpha = T(1) / T(q - 1);
        T ds[K - 1];
        T xx = x;
#pragma unroll
        for (int j = 0; j < q - 1; ++j) { ds[j] = alpha * xx; xx += T(1); }
        bs[q - 1] = (T(1) - ds[q - 2]) * bs[q - 2];
#pragma unroll
        for (int j = q - 2; j >= 1; --j) bs[j] = (T(1) - ds[j - 1]) * bs[j - 1] + ds[j] * bs[j];
        bs[0] = ds[0] * bs[0];
    }
    T val = bs[0];
#pragma unroll
    for (int j = 1; j < K; ++j) val = jsel == j ? bs[j] : val;
    return val;
}

template <typename T, int NC, int D, int M, int GS = Grp<NC, M>::G, bool OTHERK = false>
struct WindowEval {
    static constexpr int L = 2 * M;
    static constexpr int NV = D * L;
    static constexpr int NSLOT = (NV + GS - 1) / GS;
    static constexpr int NP = M + 4;
    T cs[NSLOT][NP];
    int dsel[NSLOT], jsel[NSLOT];
    bool has[NSLOT];
    T beta_s[NSLOT], bop_s[NSLOT];

    // A: TileArgs<T> or any struct with its window fields (evalmode, kernel, beta, bop, coefs) — see EvalArgs
    template <typename A>
    __device__ __forceinline__ void init(const A& a, int q) { init(a, q, a.coefs); }
    // cf: the coefficient table [D][NP][2M] (a.coefs or a copy of it, e.g. in LDS)
    template <typename A, typename CP>
    __device__ __forceinline__ void init(const A& a, int q, CP cf) {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const int k = q + s * GS;
            has[s] = k < NV;
            const int kk = has[s] ? k : 0;
            dsel[s] = kk / L;
            jsel[s] = kk % L;
            beta_s[s] = a.beta[dsel[s]];
            bop_s[s] = a.bop[dsel[s]];
            const bool poly = a.evalmode != NUFFT_EVAL_DIRECT &&
                              (!OTHERK || a.kernel == NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL || a.kernel == NUFFT_KERNEL_KAISER_BESSEL);
            if (OTHERK && a.kernel == NUFFT_KERNEL_GAUSSIAN) {
#pragma unroll
                for (int c = 0; c < NP; ++c) cs[s][c] = T(0);
                const int m = jsel[s] - (M - 1);              // cs[|m|] = exp(-(m dx)^2 / tau), gaussian.jl:81-84
                const T xm = T(m < 0 ? -m : m) * beta_s[s];
                cs[s][0] = exp(-(xm * xm) / bop_s[s]);
            } else if (poly) {
#pragma unroll
                for (int c = 0; c < NP; ++c) cs[s][c] = cf[(dsel[s] * NP + c) * L + jsel[s]];
            } else {
#pragma unroll
                for (int c = 0; c < NP; ++c) cs[s][c] = T(0);
            }
        }
    }

    // X[d]: cell fraction of this lane's point; strip: this group's NV slots in LDS.
    // PAD: every dimension's row is [PAD zeros | 2M values | PAD zeros] (the zeros are written once by the caller)
    template <int PAD = 0, typename A = TileArgs<T>>
    __device__ __forceinline__ void eval_to_strip(const A& a, const T (&X)[3], T* strip, int q) const {
        T v[NSLOT];
        eval_regs(a, X, v);
#pragma unroll
        for (int s = 0; s < NSLOT; ++s)
            if (has[s]) {
                if constexpr (PAD == 0) strip[q + s * GS] = v[s];
                else strip[dsel[s] * (L + 2 * PAD) + PAD + jsel[s]] = v[s];
            }
    }

    // the same values left in registers: v[s] is window value k = q + s * GS of the lane's point
    template <typename A>
    __device__ __forceinline__ void eval_regs(const A& a, const T (&X)[3], T (&v)[NSLOT]) const {
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
            const T x = dsel[s] == 0 ? X[0] : (dsel[s] == 1 ? X[1] : X[2]);
            T val;
            const bool poly = a.evalmode != NUFFT_EVAL_DIRECT &&
                              (!OTHERK || a.kernel == NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL || a.kernel == NUFFT_KERNEL_KAISER_BESSEL);
            if (OTHERK && !poly && a.kernel != NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL) {
                const int j = jsel[s];
                if (a.kernel == NUFFT_KERNEL_KAISER_BESSEL) {
                    const T y = (T(M - 1 - j) + x) / T(M);
                    const T z = T(1) - y * y;
                    val = dev_bessel_i0<T>(beta_s[s] * sqrt(z > T(0) ? z : T(0))) * bop_s[s];
                } else if (a.kernel == NUFFT_KERNEL_GAUSSIAN) {
                    const T dx = beta_s[s], tau = bop_s[s];
                    if (a.evalmode == NUFFT_EVAL_DIRECT) {
                        const T ys = (T(M - 1 - j) + x) * dx;
                        val = exp(-(ys * ys) / tau);
                    } else {
                        const T Xp = x * dx;
                        const T av = exp(-(Xp * Xp) / tau);
                        const T bv = exp(T(2) * Xp * dx / tau);
                        const int m = j - (M - 1);
                        const int am = m < 0 ? -m : m;
                        T bpow = T(1);
                        for (int i = 0; i < (am < M - 1 ? am : M - 1); ++i) bpow *= bv;
                        const T ac = av * cs[s][0];
                        val = m == 0 ? av : (m < 0 ? ac / bpow : (m < M ? ac * bpow : ac * bpow * bv));
                    }
                } else {
                    val = dev_bspline_value<T, M>(T(1) - x, j);
                }
            } else if (!poly) {
                val = bkb_direct<T, M>(x, jsel[s], beta_s[s], bop_s[s]);
            } else {
                const T xx = T(2) * x - T(1);
                val = cs[s][NP - 1];
#pragma unroll
                for (int c = NP - 2; c >= 0; --c) val = fma(xx, val, cs[s][c]);
            }
            v[s] = val;
        }
    }
};

template <typename T>
__device__ __forceinline__ void lds_atomic_add(T* p, T v) {
#if defined(NUFFT_ABL_NO_ATOMIC)
    {                                         // ablation build: keep operands alive, no LDS traffic
        const unsigned lo = (unsigned)(unsigned long long)p;
        const float f = (float)v;
        asm volatile("" ::"v"(lo), "v"(f));
    }
#else
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}

// Row walker for tile <-> global traffic: a wave instruction covers RPW rows of the tile
// (row = fixed l2, l3; w_row reals), lanes running along dimension 1 so that global addresses are
// contiguous within a row.
struct RowWalker {
    int lanes_per_row, rpw, sub, lane_in_row;
    int rows2, rows_total, row, l2, l3, step, step2, step3;
    __device__ RowWalker(int rows2_, int rows3_, int w_row, int wave, int nwaves, int lane) {
        int lpr = kWave;
        while (lpr / 2 >= w_row && lpr > 1) lpr >>= 1;
        lanes_per_row = lpr;
        rpw = kWave / lpr;
        sub = lane / lpr;
        lane_in_row = lane % lpr;
        rows2 = rows2_;
        rows_total = rows2_ * rows3_;
        row = wave * rpw + sub;
        l2 = row % rows2;
        l3 = row / rows2;
        step = nwaves * rpw;
        step2 = step % rows2;
        step3 = step / rows2;
    }
    __device__ __forceinline__ bool valid() const { return row < rows_total; }
    __device__ __forceinline__ void next() {
        row += step;
        l2 += step2;
        l3 += step3;
        if (l2 >= rows2) { l2 -= rows2; l3 += 1; }
    }
};

// L Float64 LDS atomics with immediate offsets J * PLANE_BYTES from one address (values w * w3[J]).
template <int OFF>
__device__ __forceinline__ void lds_add_imm(uint32_t addr, double v) {
    static_assert(OFF >= 0 && OFF < 65536, "LDS immediate offset out of range");
    asm volatile("ds_add_f64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int L, int PLANE_BYTES, typename T, int... J>
__device__ __forceinline__ void lds_add_planes(double* p, T w, const T (&w3)[L], std::integer_sequence<int, J...>) {
    const uint32_t addr = (uint32_t)(uintptr_t)p;
    (lds_add_imm<J * PLANE_BYTES>(addr, (double)(w * w3[J])), ...);
}

// Hand-scheduled LDS reads for the interpolation gather of the compile-time-tile variant: one base address,
// immediate offsets (no address arithmetic) and explicit s_waitcnt, written as inline assembly so that the
// compiler can neither pair them into ds_read2_b64 (whose two addresses share banks at an unaligned row
// stride) nor serialise them.  The caller keeps at most two planes (2 * 2M reads) in flight.
template <typename T, int OFF>
__device__ __forceinline__ void lds_read_imm(T& dst, uint32_t addr) {
    static_assert(OFF >= 0 && OFF < 65536, "LDS immediate offset out of range");
    if constexpr (sizeof(T) == 16) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
    else if constexpr (sizeof(T) == 8) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
    else asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <typename T, int R, int FIRST_OFF, int ROW_BYTES, int... J>
__device__ __forceinline__ void lds_read_rows(T (&b)[R], uint32_t addr, std::integer_sequence<int, J...>) {
    (lds_read_imm<T, FIRST_OFF + J * ROW_BYTES>(b[J], addr), ...);
}
// wait until at most PENDING LDS operations are outstanding; the registers are operands so that their
// consumers are ordered after the wait
template <int PENDING, typename T, int R>
__device__ __forceinline__ void lds_wait_rows(T (&b)[R]) {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PENDING));
#pragma unroll
    for (int j = 0; j < R; ++j) asm volatile("" : "+v"(b[j]));
}

// Row groups G, G + 1, ... of R rows each (group g = rows (g % (L / R)) * R ... of plane g / (L / R)):
// `cur` was requested by the caller; request group G + 1 into `nxt`, wait for `cur`, consume it, swap.
template <typename T, int L, int R, int PB, int RB, int G, typename F>
__device__ __forceinline__ void interp_row_groups(T (&cur)[R], T (&nxt)[R], uint32_t baddr, F&& consume) {
    constexpr int GPP = L / R, NG = L * GPP;
    if constexpr (G + 1 < NG) {
        lds_read_rows<T, R, ((G + 1) / GPP) * PB + ((G + 1) % GPP) * R * RB, RB>(nxt, baddr, std::make_integer_sequence<int, R>{});
        lds_wait_rows<R>(cur);              // the R reads of group G + 1 may still be in flight
        consume(cur, G);
        interp_row_groups<T, L, R, PB, RB, G + 1>(nxt, cur, baddr, consume);
    } else {
        lds_wait_rows<0>(cur);
        consume(cur, G);
    }
}

// f(std::integral_constant<int, G0>) for G0 = 0, 2, 4, ... < N
template <int G0, int N, typename F>
__device__ __forceinline__ void for_each_pair(F&& f) {
    if constexpr (G0 < N) {
        f(std::integral_constant<int, G0>{});
        for_each_pair<G0 + 2, N>(f);
    }
}

// Splits the `nruns` runs of the item table (uint2 = [first, last) of the sorted array) in place into
// work items: every run is cut into c * nslices pieces (c such that a slice has about kItemTarget
// items; piece lengths a multiple of `ppw` points: full chunks), of which this workgroup — slice `slice`
// of the `nslices` that share the tile — keeps pieces slice, slice + nslices, ...  Returns the new item
// count (nruns * c <= max_items).  Must be called by the whole workgroup after the table has been
// written (it contains the barriers it needs).
__device__ __forceinline__ int split_work_items(uint2* items, int nruns, int max_items, int ppw, int tid, int nthreads,
                                                int slice, int nslices) {
    __syncthreads();
    if (nruns <= 0) return 0;
    int c = nruns >= kItemTarget ? 1 : (kItemTarget + nruns - 1) / nruns;
    if (nruns * c > max_items) c = max_items / nruns;
    if (c < 1) c = 1;
    if (c == 1 && nslices == 1) return nruns;
    const int total = nruns * c;
    // In-place expansion from the top: a round of `nthreads` items reads its runs (slots it / c <= it),
    // then writes slots that no later round reads.
    for (int hi = total; hi > 0; hi -= nthreads) {
        const int it = hi - 1 - tid;
        uint2 run = make_uint2(0u, 0u);
        if (it >= 0) run = items[it / c];
        __syncthreads();
        if (it >= 0) {
            const uint32_t len = run.y - run.x;
            const uint32_t pieces = (uint32_t)c * (uint32_t)nslices;
            uint32_t piece = (len + pieces - 1) / pieces;
            piece = (piece + (uint32_t)ppw - 1) / (uint32_t)ppw * (uint32_t)ppw;
            const uint64_t first = (uint64_t)run.x + (uint64_t)((it % c) * nslices + slice) * piece;
            const uint32_t q0 = first < run.y ? (uint32_t)first : run.y;
            const uint32_t q1 = (uint64_t)q0 + piece < run.y ? q0 + piece : run.y;
            items[it] = make_uint2(q0, q1);
        }
        __syncthreads();
    }
    return total;
}

__device__ __forceinline__ int wrap_index(int gidx, int N) {
    if (gidx < 0) gidx += N;
    if (gidx >= N) gidx -= N;
    if (gidx >= N) gidx -= N;
    return gidx;
}

__device__ __forceinline__ void tile_coords(int tile_id, const TileShape& ts, int (&t)[3]) {
    int rem = tile_id;
    t[0] = rem % ts.nt[0]; rem /= ts.nt[0];
    t[1] = rem % ts.nt[1]; rem /= ts.nt[1];
    t[2] = rem;
}

// ---------------------------------------------------------------------------------------------
// Cube accumulation of one chunk of PPW points (real data, D = 3, stencils of three cubes per dimension)
// ---------------------------------------------------------------------------------------------
template <int I, int N, typename F>
__device__ __forceinline__ void static_for_cubes(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_cubes<I + 1, N>(f);
    }
}

__device__ __forceinline__ double bperm_t(double x, int src_lane) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b & 0xffffffffLL));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double bperm_t(float x, int src_lane) {
    return (double)__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, x)));
}
__device__ __forceinline__ double lds_ld_f64(const double* p) { return *p; }
__device__ __forceinline__ double lds_ld_f64(const float* p) { return (double)*p; }

// The chunk's points sit in the lane groups g = 0 .. PPW-1 (G lanes each; every lane of a group holds the group's
// stencil start s[d] in tile coordinates and the value), their windows in the wave's strip (rows of 2M + 2 with a zero
// at both ends).  Four points at a time (k = lane >> 4) against the cubes of the tile they can reach: the bins of the
// chunk's points share dimensions 2 and 3 (a chunk never leaves its run of the sorted array) and ascend along
// dimension 1, so the cube range is three cubes in y and z and [first bin - 1, last bin + 1] in x.  Lane roles of the
// matrix instruction: operands (k, b, i / j), result (x = lane >> 4, y = (lane >> 2) & 3, z = lane & 3).
template <typename T, int M, int G, int PPW, int N1, int N2, int N3, int RS, int PS>
__device__ __forceinline__ void spread_cubes(double* tile, const T* strip_wave, const int (&s)[3], T value, unsigned long long okmask,
                                             const int (&neff)[3], int lane) {
    constexpr int L = 2 * M, SROW = L + 2;
    constexpr int NCX = N1 / 4;
    static_assert(N1 % 4 == 0 && N2 % 4 == 0 && N3 % 4 == 0 && PPW % 4 == 0, "cubes need 4-aligned tiles and K-batches of four points");
    const int mk = lane >> 4, mb = (lane >> 2) & 3, mi = lane & 3;
    // result roles of the matrix instruction: z = mi, y = mb, x = mk
#pragma unroll
    for (int bt = 0; bt < PPW / 4; ++bt) {
        const unsigned okb = (unsigned)((okmask >> (bt * 4 * G)) & ((G * 4 >= 64) ? ~0ull : ((1ull << (4 * G)) - 1ull)));
        if (okb == 0u) continue;                          // none of the four points touches the tile
        // first / last point of the batch that touches the tile (group granularity: bit g * G of okb)
        const int gfirst = (__builtin_ctz(okb) / G) + bt * 4, glast = ((31 - __builtin_clz(okb)) / G) + bt * 4;
        const int s1f = __builtin_amdgcn_readlane(s[0], gfirst * G), s1l = __builtin_amdgcn_readlane(s[0], glast * G);
        const int s2f = __builtin_amdgcn_readlane(s[1], gfirst * G), s3f = __builtin_amdgcn_readlane(s[2], gfirst * G);
        // cubes (tile coordinates): bin of a point = (s + M - 1) >> 2, its stencil reaches bins - 1 .. + 1
        const int cxlo = max(((s1f + M - 1) >> 2) - 1, 0), cxhi = min(((s1l + M - 1) >> 2) + 1, (neff[0] >> 2) - 1);
        // (first cube of the y / z ranges, not below -1: a stencil that starts two cubes below the tile only reaches
        // cube 0, which the shifted range still covers — and the atomics' base address stays inside the LDS)
        const int cy0 = max(((s2f + M - 1) >> 2) - 1, -1), cz0 = max(((s3f + M - 1) >> 2) - 1, -1);
#if defined(NUFFT_CUBES_DEBUG)
        if (lane == 0) printf("cubes bt %d okb %08x gfirst %d glast %d s1f %d s1l %d s2f %d s3f %d cx [%d,%d] cy0 %d cz0 %d neff %d %d %d\n", bt, okb, gfirst, glast, s1f, s1l, s2f, s3f, cxlo, cxhi, cy0, cz0, neff[0], neff[1], neff[2]);
#endif
        // the lane's point: group bt * 4 + k
        const int src = (bt * 4 + mk) * G;
        const int p1 = __builtin_amdgcn_ds_bpermute(src << 2, s[0]);
        const int p2 = __builtin_amdgcn_ds_bpermute(src << 2, s[1]);
        const int p3 = __builtin_amdgcn_ds_bpermute(src << 2, s[2]);
        const bool pok = (okmask >> src) & 1ull;
        const double vsrc = bperm_t(value, src);         // unconditionally: ds_bpermute returns 0 from source lanes that are masked off
        const double v = pok ? vsrc : 0.0;
        const T* rows = strip_wave + (bt * 4 + mk) * (3 * SROW);
        // window values: row[clamp(4 c + lane coordinate - s, -1, L) + 1]
        double w1[NCX], w2[3], bz[3];
#pragma unroll
        for (int cx = 0; cx < NCX; ++cx) w1[cx] = lds_ld_f64(rows + 1 + max(-1, min(4 * cx + mi - p1, L)));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            w2[r] = lds_ld_f64(rows + SROW + 1 + max(-1, min(4 * (cy0 + r) + mb - p2, L)));
            bz[r] = lds_ld_f64(rows + 2 * SROW + 1 + max(-1, min(4 * (cz0 + r) + mi - p3, L))) * v;
        }
        // cube (cx, cy0 + ry, cz0 + rz) = constant offset from the address of cube (0, cy0, cz0); cy0, cz0 >= -1: the
        // caller keeps the tile at least 4 (PS + RS) doubles above the start of the LDS so that this base stays inside
        // it.  (Compiler-issued atomics: the hazard between an MFMA and a reader of its result is the compiler's to
        // handle, which it does not do for inline assembly.)
        double* cube0 = tile + (mi * PS + mb * RS + mk) + (4 * cz0 * PS + 4 * cy0 * RS);
        const unsigned ncy = (unsigned)(neff[1] >> 2), ncz = (unsigned)(neff[2] >> 2);
        static_for_cubes<0, NCX * 3>([&](auto Ic) __attribute__((always_inline)) {
            constexpr int I = decltype(Ic)::value, cx = I / 3, ry = I % 3;
            if (cx >= cxlo && cx <= cxhi && (unsigned)(cy0 + ry) < ncy) {
                const double A = w1[cx] * w2[ry];
                double d[3];
#pragma unroll
                for (int rz = 0; rz < 3; ++rz) d[rz] = __builtin_amdgcn_mfma_f64_4x4x4f64(A, bz[rz], 0.0, 0, 0, 0);
#pragma unroll
                for (int rz = 0; rz < 3; ++rz)
                    if ((unsigned)(cz0 + rz) < ncz) lds_atomic_add(cube0 + (4 * rz * PS + 4 * ry * RS + 4 * cx), d[rz]);
            }
        });
    }
}

// ---------------------------------------------------------------------------------------------
// Spreading (output-driven)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float readlane_t(float x, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l));
}
__device__ __forceinline__ double readlane_t(double x, int l) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), l);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Windows are evaluated with the group mapping (G lanes per point, 64 / G points per chunk); the
// accumulation then walks the chunk's points one at a time with the FACE mapping: the 64 lanes own the
// (component, j1, j2) elements of the stencil face (NPASS passes when the face has more than 64
// elements) and loop over j3.  Rows of the LDS tile are strided so that the rows one wave instruction
// touches fall on disjoint banks (8.8 cycles per ds_add_f64 wave instruction instead of 12-16 for
// randomly placed segments, scripts/microbench.hip), and stencil planes outside the tile are skipped
// by a scalar branch.  WRAP = some axis is spanned by a single tile (small grids): stencil indices
// then wrap around that axis instead of being clipped; the hot instantiation (WRAP = false) carries
// none of that code.
// FIXEDT (only with !WRAP): the tile is the compile-time one of fixed_spread_tile(); the LDS atomics of a point
// whose planes all lie inside the tile are then issued with immediate offsets from one address.
// CUBES (only with FIXEDT, real data, D = 3, M <= 4, oversampled sizes that are multiples of 4): four points at a time
// are accumulated cube by cube — one v_mfma_f64_4x4x4_4b per 4 x 4 x 4 cube of cells forms the sum of the four points'
// contributions (A = w1 w2, B = v w3, see patch_kernels.h for the operand layout), one ds_add_f64 adds it to the tile —
// instead of one ds_add_f64 per point and stencil plane: 9 instead of 12.5 LDS atomics per point, and the products
// leave the vector ALUs.
template <typename T, bool CPLX, int D, int M, bool WRAP, bool OTHERK = false, bool FIXEDT = false, bool CUBES = false>
__global__ __launch_bounds__(1024) void spread_tile_kernel(TileArgs<T> a) {
    constexpr int NC = CPLX ? 2 : 1;
    constexpr int L = 2 * M;
    using GP = Grp<NC, M>;
    using A = double;
    constexpr FixedTileDims FS = fixed_spread_tile((int)sizeof(T), NC, D, M);
    static_assert(!FIXEDT || (FS.n[0] > 0 && !WRAP), "no compile-time tile for this instantiation");
    constexpr int FACE = GP::W1 * (D >= 2 ? L : 1);
    constexpr int NPASS = (FACE + kWave - 1) / kWave;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const int nthreads = blockDim.x;
    const int nwaves = nthreads / kWave;
    const Geom& g = a.g;
    const TileShape& ts = g.sp;

    // slot -> (tile, slice): heavy tiles are shared by several workgroups (balance.hip)
    const uint32_t nslots = *a.desc_total;
    if (blockIdx.x >= nslots) return;
    const uint2 de = a.desc[xcd_remap_chunked(blockIdx.x, (int)nslots, a.xcd_chunk)];
    const int tile_id = (int)de.x, slice = (int)(de.y >> 16), nslices = (int)(de.y & 0xffffu);
    const int comp_id = blockIdx.y;
    int t[3];
    tile_coords(tile_id, ts, t);
    int org[3], neff[3];
    bool wrapd[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int nd = FIXEDT ? FS.n[d] : ts.n[d];
        org[d] = t[d] * nd;
        neff[d] = min(nd, g.Nover[d] - org[d]);
        wrapd[d] = WRAP && ts.nt[d] == 1;   // a single tile spans the axis: wrap instead of clip
    }
    const int RS = FIXEDT ? FS.row_stride : ts.row_stride;
    const int PS = FIXEDT ? FS.plane_stride : ts.plane_stride;

    static_assert(!CUBES || (FIXEDT && !CPLX && D == 3 && M <= 4), "cube accumulation: compile-time tile, real data, 3-D, M <= 4");
    constexpr int SPAD = CUBES ? 1 : 0;                 // zeros around every dimension's window values in the strip
    constexpr int SROW = L + 2 * SPAD;                  // strip row of one dimension
    const LdsLayout lay = lds_layout(ts.elems, (int)sizeof(A), (int)sizeof(T), D, M, NC, nwaves, ts.max_items, spread_strip_pad(D, NC));
    // [tile | items | strips]; the cube variant puts the tile last: its immediate-offset atomics start from the address
    // of a cube one cube row / layer below the tile origin, which must not fall below the start of the LDS
    const int tile_off = CUBES ? lay.items_bytes + nwaves * lay.strip_bytes_per_wave : 0;
    const int rest_off = CUBES ? 0 : lay.tile_bytes;
    A* tile = reinterpret_cast<A*>(smem + tile_off);
    uint2* items = reinterpret_cast<uint2*>(smem + rest_off);
    int* next_item = reinterpret_cast<int*>(items + ts.max_items);
    T* strip_wave = reinterpret_cast<T*>(smem + rest_off + lay.items_bytes + wave * lay.strip_bytes_per_wave);

    for (int i = tid; i < ts.elems; i += nthreads) tile[i] = A(0);
    if constexpr (CUBES) {
        if (tile_off < 4 * (FS.plane_stride + FS.row_stride) * 8) __builtin_trap();                              // the padding zeros of the strips (never overwritten)
        for (int i = lane; i < GP::PPW * D * 2; i += kWave) strip_wave[(i >> 1) * SROW + (i & 1) * (L + 1)] = T(0);
    }

    // evaluation roles
    const int grp = lane / GP::G, q = lane % GP::G;
    T* strip = strip_wave + grp * (D * SROW);
    WindowEval<T, NC, D, M, GP::G, OTHERK> we;
    we.init(a, q);
    // accumulation roles
    int j1f[NPASS], j2f[NPASS], cmpf[NPASS];
    bool actf[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int e = lane + ps * kWave;
        actf[ps] = e < FACE;
        const int e1 = e % GP::W1;
        j2f[ps] = (e / GP::W1) % L;
        cmpf[ps] = e1 % NC;
        j1f[ps] = e1 / NC;
    }

    // Work items: the bins whose points can touch this tile form, per (bin2, bin3) row, one or two
    // contiguous runs of the sorted array.  All runs are looked up at once (one round of loads for
    // the whole workgroup) into an LDS table, and waves then pull items from a shared counter.
    BinSegs seg[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d < D) seg[d] = bin_segments(org[d] - M, org[d] + neff[d] + M - 1, g.Nover[d], g.blog[d], g.nb[d]);
        else { seg[d].n = 1; seg[d].lo[0] = 0; seg[d].len[0] = 1; seg[d].lo[1] = 0; seg[d].len[1] = 0; }
    }
    const int R2 = seg[1].total(), R3 = seg[2].total();
    const int nruns = R2 * R3 * seg[0].n;
    // (cannot exceed the table: plan creation replays this arithmetic for every tile position, exact_tile_runs in
    // plan_math.cpp, and refuses the plan otherwise — a device trap would take the caller's whole HIP context down)
#if defined(NUFFT_DEBUG_TRAPS)
    if (nruns > ts.max_items) __builtin_trap();
#endif
    for (int item = tid; item < nruns; item += nthreads) {
        const int sg = item % seg[0].n;
        const int r2 = (item / seg[0].n) % R2;
        const int r3 = item / (seg[0].n * R2);
        const int bin0 = (seg[2].bin(r3) * g.nb[1] + seg[1].bin(r2)) * g.nb[0] + (sg ? seg[0].lo[1] : seg[0].lo[0]);
        items[item] = make_uint2(a.offsets[bin0], a.offsets[bin0 + (sg ? seg[0].len[1] : seg[0].len[0])]);
    }
    if (tid == 0) *next_item = 0;
    const int nitems = split_work_items(items, nruns, ts.max_items, GP::PPW, tid, nthreads, slice, nslices);

    const PointRec<T, D>* sorted = static_cast<const PointRec<T, D>*>(a.sorted);
    const T* vin = a.vin[comp_id];

    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(next_item, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= nitems) break;
        const uint2 pr = items[item];
        const uint32_t p0 = pr.x, p1 = pr.y;
        if (p0 >= p1) continue;
        // software pipeline: the record of the next chunk is requested before the current chunk is
        // processed, its value right after
        PointRec<T, D> rec = sorted[min(p0 + (uint32_t)grp, p1 - 1)];
        T vcur = T(0);
        if (q < NC) {
            vcur = vin[(int64_t)rec.idx * NC + q];
            if constexpr (OTHERK) { if (a.weights) vcur *= a.weights[rec.idx]; }   // callbacks.nonuniform(v, n), src/spreading/gpu.jl:289
        }
        for (uint32_t pc = p0; pc < p1; pc += GP::PPW) {
            const uint32_t p = pc + grp;
            const bool have = p < p1;
            const uint32_t npc = pc + GP::PPW;
            const bool more = npc < p1;
            PointRec<T, D> recn = rec;
            if (more) recn = sorted[min(npc + (uint32_t)grp, p1 - 1)];
            int s[3] = {0, 0, 0};
            T X[3] = {T(0), T(0), T(0)};
            bool ok = have;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int c = cell_of(rec.r[d], g.Nover[d]);
                X[d] = rec.r[d] - T(c);
                int sd = c - (M - 1) - org[d];             // local index of the first stencil node
                if (!wrapd[d]) {
                    if (sd > neff[d] - 1) sd -= g.Nover[d];    // periodic image next to this tile
                    if (sd < -(L - 1)) sd += g.Nover[d];
                    ok = ok && (sd >= -(L - 1)) && (sd <= neff[d] - 1);
                }
                s[d] = sd;
            }
            const unsigned long long okmask = __ballot(ok);
            const T vmine = vcur;
            if (okmask != 0ull) {                             // else nothing of this chunk touches the tile
            wave_lds_fence();
#if !defined(NUFFT_ABL_NO_EVAL)
            we.template eval_to_strip<SPAD>(a, X, strip, q);
#endif
            wave_lds_fence();

            if constexpr (CUBES) {
                spread_cubes<T, M, GP::G, GP::PPW, FS.n[0], FS.n[1], FS.n[2], FS.row_stride, FS.plane_stride>(tile, strip_wave, s, vmine, okmask, neff, lane);
            } else {
#if !defined(NUFFT_ABL_NO_VISIT)
            // one point of the chunk: w1v / w2v are the lane's window values of dimensions 1 and 2 (per pass),
            // w3a the lane's share of the dimension-3 values (value l & 15 in lane l of every 16-lane row)
            auto do_point = [&](int gi, const T (&w1v)[NPASS], const T (&w2v)[NPASS], T w3a) {
                const int src = gi * GP::G;                    // first lane of the point's group
                if (!((okmask >> src) & 1ull)) return;
                const int S1 = __builtin_amdgcn_readlane(s[0], src);
                const int S2 = D >= 2 ? __builtin_amdgcn_readlane(s[1], src) : 0;
                const int S3 = D >= 3 ? __builtin_amdgcn_readlane(s[2], src) : 0;
                const T Vre = readlane_t(vmine, src);
                const T Vim = CPLX ? readlane_t(vmine, src + (CPLX ? 1 : 0)) : T(0);
                const T* sp = strip_wave + gi * (D * L);
                // the 2M window values of dimension 3 by DPP row broadcasts: one VALU instruction per value
                T w3[L];
                if constexpr (D >= 3) {
                    T w3b = T(0);
                    if constexpr (L > 16) w3b = sp[2 * L + min(16 + (lane & 15), L - 1)];
                    if constexpr (L <= 16) {
                        row_bcast_all(w3a, w3, 0, std::make_integer_sequence<int, L>{});
                    } else {
#pragma unroll
                        for (int j = 0; j < L; ++j) w3[j] = j < 16 ? row_bcast(w3a, j) : row_bcast(w3b, j - 16);
                    }
                }
                // valid planes as a bit mask (bit j3 set: plane S3 + j3 lies inside the tile).  One scalar
                // unit serves the whole CU, so the per-plane control is kept to a bit test + branch.
                unsigned planes = (1u << L) - 1u;
                int first3 = 0;
                if constexpr (D >= 3) {
                    if (!wrapd[2]) {
                        const int lo3 = max(0, -S3);
                        const int hi3 = min(L, neff[2] - S3);
                        planes = ((1u << hi3) - 1u) & ~((1u << lo3) - 1u);
                        first3 = lo3;
                    }
                }
#pragma unroll
                for (int ps = 0; ps < NPASS; ++ps) {
                    int l1 = S1 + j1f[ps];
                    if (WRAP && wrapd[0]) { if (l1 < 0) l1 += g.Nover[0]; if (l1 >= g.Nover[0]) l1 -= g.Nover[0]; }
                    bool lane_ok = actf[ps] && (unsigned)l1 < (unsigned)neff[0];
                    T w = w1v[ps] * (CPLX ? (cmpf[ps] ? Vim : Vre) : Vre);
                    A* addr = tile + l1 * NC + cmpf[ps];
                    if constexpr (D >= 2) {
                        int l2 = S2 + j2f[ps];
                        if (WRAP && wrapd[1]) { if (l2 < 0) l2 += g.Nover[1]; if (l2 >= g.Nover[1]) l2 -= g.Nover[1]; }
                        lane_ok = lane_ok && (unsigned)l2 < (unsigned)neff[1];
                        w *= w2v[ps];
                        addr += l2 * RS;
                    }
                    if constexpr (D <= 2) {
                        if (lane_ok) lds_atomic_add(addr, (A)w);
                    } else {
                        if (lane_ok) {
                            if (WRAP && wrapd[2]) {
#pragma unroll
                                for (int j3 = 0; j3 < L; ++j3) {
                                    int l3 = S3 + j3;
                                    if (l3 < 0) l3 += g.Nover[2];
                                    if (l3 >= g.Nover[2]) l3 -= g.Nover[2];
                                    lds_atomic_add(addr + l3 * PS, (A)(w * w3[j3]));
                                }
                            } else {
                                A* pl = addr + (S3 + first3) * PS;
                                if (planes == (1u << L) - 1u) {      // all planes inside: no per-plane control
                                    if constexpr (FIXEDT && (L - 1) * FS.plane_stride * 8 < 65536) {
                                        lds_add_planes<L, FS.plane_stride * 8>(pl, w, w3, std::make_integer_sequence<int, L>{});
                                    } else {
#pragma unroll
                                        for (int j3 = 0; j3 < L; ++j3) {
                                            lds_atomic_add(pl, (A)(w * w3[j3]));
                                            pl += PS;
                                        }
                                    }
                                } else {
#pragma unroll
                                    for (int j3 = 0; j3 < L; ++j3) {
                                        if (planes & (1u << j3)) {
                                            lds_atomic_add(pl, (A)(w * w3[j3]));
                                            pl += PS;
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            };
            if constexpr (FIXEDT && NUFFT_SPREAD_ASM_STRIP && NPASS == 1 && D == 3 && L <= 16 && GP::PPW % 2 == 0) {
                // The three strip reads of a point are issued together for two points at a time, from inline
                // assembly (the compiler otherwise sinks two of them behind the lane-mask branch, which costs a
                // second LDS round trip per point), followed by one wait.
                const uint32_t sb1 = (uint32_t)(uintptr_t)(strip_wave + j1f[0]);
                const uint32_t sb2 = (uint32_t)(uintptr_t)(strip_wave + L + j2f[0]);
                const uint32_t sb3 = (uint32_t)(uintptr_t)(strip_wave + 2 * L + min(lane & 15, L - 1));
                constexpr int PB = D * L * (int)sizeof(T);     // bytes between the strips of consecutive points
                for_each_pair<0, GP::PPW>([&](auto G0c) {
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
                for (int gi = 0; gi < GP::PPW; ++gi) {
                    const T* sp = strip_wave + gi * (D * L);
                    T w1v[NPASS], w2v[NPASS];
#pragma unroll
                    for (int ps = 0; ps < NPASS; ++ps) {
                        w1v[ps] = sp[j1f[ps]];
                        w2v[ps] = D >= 2 ? sp[L + j2f[ps]] : T(1);
                    }
                    const T w3a = D >= 3 ? sp[2 * L + min(lane & 15, L - 1)] : T(0);
                    do_point(gi, w1v, w2v, w3a);
                }
            }
#else
            asm volatile("" ::"v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(vmine));
#endif
            }   // !CUBES
            }   // okmask != 0
            if (more && q < NC) {
                vcur = vin[(int64_t)recn.idx * NC + q];
                if constexpr (OTHERK) { if (a.weights) vcur *= a.weights[recn.idx]; }
            }
            rec = recn;
        }
    }
    // the immediate-offset atomics are inline assembly: the compiler does not know that they are in flight
    if constexpr (FIXEDT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();

    // store the finished interior: every grid cell is written exactly once (no zero fill needed)
    T* grid = a.grid[comp_id];
    const int w_row = NC * neff[0];
    RowWalker rw(D >= 2 ? neff[1] : 1, D >= 3 ? neff[2] : 1, w_row, wave, nwaves, lane);
    for (; rw.valid(); rw.next()) {
        int64_t rowbase = 0;
        if constexpr (D >= 2) rowbase = org[1] + rw.l2;
        if constexpr (D >= 3) rowbase += (int64_t)(org[2] + rw.l3) * g.Nover[1];
        rowbase = (rowbase * g.Nover[0] + org[0]) * NC;
        const A* src = tile + rw.l2 * RS + rw.l3 * PS;
        if (nslices == 1) {
            for (int e = rw.lane_in_row; e < w_row; e += rw.lanes_per_row) grid[rowbase + e] = (T)src[e];
        } else {
            // one of several slices of this tile: add the partial tile (interior zeroed by zero_split_tiles_kernel)
            for (int e = rw.lane_in_row; e < w_row; e += rw.lanes_per_row) {
                const T v = (T)src[e];
                if (v != T(0)) (void)__hip_atomic_fetch_add(&grid[rowbase + e], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Interpolation
// ---------------------------------------------------------------------------------------------
// FIXED: the tile shape is the compile-time one of fixed_interp_tile() (the host launches this variant
// only when the plan's tile equals it), which turns the LDS strides into immediates.
template <typename T, bool CPLX, int D, int M, bool FIXED, bool OTHERK = false>
__global__ __launch_bounds__(1024) void interp_tile_kernel(TileArgs<T> a) {
    constexpr int NC = CPLX ? 2 : 1;
    constexpr int L = 2 * M;
    using GP = Grp<NC, M>;
    constexpr FixedTileDims FD = fixed_interp_tile((int)sizeof(T), NC, D, M);
    static_assert(!FIXED || FD.n[0] > 0, "no compile-time tile for this instantiation");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const int nthreads = blockDim.x;
    const int nwaves = nthreads / kWave;
    const Geom& g = a.g;
    const TileShape& ts = g.ip;

    // (one workgroup per slot.  A bounded grid whose workgroups stride over the slots would make the launch that finds the
    // ring at work cost 8192 workgroups instead of one per tile — 0.76 ms at C3's 7e5 tiles — but the loop around the body
    // cost 15-60 spilled registers in half of the instantiations, including every 1-D / 2-D one: not kept.)
    const uint32_t nslots = *a.desc_total;
    if (blockIdx.x >= nslots) return;
    if (a.march_flag && *a.march_flag != 0u) return;                // this point set goes to interp_march_kernel
    const uint2 de = a.desc[xcd_remap_chunked(blockIdx.x, (int)nslots, a.xcd_chunk)];
    const int tile_id = (int)de.x, slice = (int)(de.y >> 16), nslices = (int)(de.y & 0xffffu);
    const int comp_id = blockIdx.y;
    int t[3];
    tile_coords(tile_id, ts, t);
    int org[3], neff[3], P[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int nd = FIXED ? FD.n[d] : ts.n[d];
        org[d] = t[d] * nd;
        neff[d] = min(nd, g.Nover[d] - org[d]);
        P[d] = d < D ? nd + L - 1 : 1;
    }
    const int RS = FIXED ? FD.row_stride : ts.row_stride;
    const int PS = FIXED ? FD.row_stride * (FD.n[1] + L - 1) : ts.plane_stride;

    // points of this tile: a box of bins, one contiguous run of the sorted array per (bin2, bin3)
    int blo[3], bcnt[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        blo[d] = d < D ? org[d] >> g.blog[d] : 0;
        bcnt[d] = d < D ? ((org[d] + neff[d] - 1) >> g.blog[d]) - blo[d] + 1 : 1;
    }
    const int nruns = bcnt[1] * bcnt[2];
#if defined(NUFFT_DEBUG_TRAPS)
    if (nruns > ts.max_items) __builtin_trap();     // checked at plan creation (exact_tile_runs, plan_math.cpp)
#endif

    const LdsLayout lay = lds_layout(ts.elems, (int)sizeof(T), (int)sizeof(T), D, M, NC, nwaves, ts.max_items);
    T* tile = reinterpret_cast<T*>(smem);
    uint2* items = reinterpret_cast<uint2*>(smem + lay.tile_bytes);
    int* next_item = reinterpret_cast<int*>(items + ts.max_items);
    T* strip_wave = reinterpret_cast<T*>(smem + lay.tile_bytes + lay.items_bytes + wave * lay.strip_bytes_per_wave);

    // Work items: one contiguous run of the sorted array per (bin2, bin3) row of the tile, looked up
    // once into an LDS table.  The tile load is skipped when the tile holds no points (the flag lives
    // in the dynamic LDS region: HIP's __syncthreads_or would add static LDS to the 160 KiB request).
    {
        if (tid == 0) *next_item = 0;
        __syncthreads();
        int any = 0;
        for (int item = tid; item < nruns; item += nthreads) {
            const int bin0 = ((blo[2] + item / bcnt[1]) * g.nb[1] + blo[1] + item % bcnt[1]) * g.nb[0] + blo[0];
            const uint2 pr = make_uint2(a.offsets[bin0], a.offsets[bin0 + bcnt[0]]);
            items[item] = pr;
            any |= pr.x != pr.y;
        }
        if (any) *next_item = 1;
        __syncthreads();
        const int f = *next_item;
        __syncthreads();
        if (!f) return;
        if (tid == 0) *next_item = 0;
    }
    const int nitems = split_work_items(items, nruns, ts.max_items, GP::PPW, tid, nthreads, slice, nslices);

    // load the padded tile with periodic wrap (gridvalues_to_local_memory!, src/interpolation/gpu.jl:331-355)
    const T* grid = a.grid[comp_id];
#if !defined(NUFFT_ABL_INTERP_NOLOAD)
    {
        // U rows per wave are in flight at once: all global loads of a batch are issued before the first
        // LDS store waits for them (one row at a time leaves the load latency fully exposed — with one
        // workgroup per CU nothing else hides it).
#ifndef NUFFT_INTERP_LOAD_ROUNDS
#define NUFFT_INTERP_LOAD_ROUNDS 1
#endif
        // compile-time tile: as many rows per wave as it takes to load the tile in NUFFT_INTERP_LOAD_ROUNDS rounds
        // (a mostly empty last round costs a full memory round trip)
        constexpr int ROWS_PER_ROUND1 = 16 * (NC * (FD.n[0] + L - 1) <= 32 ? 2 : 1);   // 16 waves, U = 1
        constexpr int ROWS_FIXED = (FD.n[1] + (D >= 2 ? L - 1 : 0)) * (FD.n[2] + (D >= 3 ? L - 1 : 0));
        constexpr int U_FIXED = (ROWS_FIXED + ROWS_PER_ROUND1 * NUFFT_INTERP_LOAD_ROUNDS - 1) / (ROWS_PER_ROUND1 * NUFFT_INTERP_LOAD_ROUNDS);
        constexpr int U = (FIXED && U_FIXED >= 4 && U_FIXED <= 18) ? U_FIXED : 8;
        const int w_row = NC * P[0];
        RowWalker rw(P[1], P[2], w_row, wave, nwaves, lane);
        const int o1 = org[0] - (M - 1), o2 = org[1] - (M - 1), o3 = org[2] - (M - 1);
        while (rw.valid()) {
            const T* src[U];
            T* dst[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = rw.valid();
                int64_t rowbase = 0;
                if constexpr (D >= 2) rowbase = (int64_t)wrap_index(o2 + rw.l2, g.Nover[1]);
                if constexpr (D >= 3) rowbase += (int64_t)wrap_index(o3 + min(rw.l3, P[2] - 1), g.Nover[2]) * g.Nover[1];
                src[u] = grid + rowbase * ((int64_t)g.Nover[0] * NC);
                dst[u] = tile + rw.l2 * RS + rw.l3 * PS;
                rw.next();
            }
            auto copy = [&](int e) {
                const int l1 = e / NC, c = e % NC;
                const int goff = wrap_index(o1 + l1, g.Nover[0]) * NC + c;
                T v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = ok[u] ? src[u][goff] : T(0);
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (ok[u]) dst[u][e] = v[u];
            };
            if (w_row <= kWave) {            // one lane per real of the row: straight-line code
                if (rw.lane_in_row < w_row) copy(rw.lane_in_row);
            } else {
                for (int e = rw.lane_in_row; e < w_row; e += rw.lanes_per_row) copy(e);
            }
        }
    }
#endif
#if defined(NUFFT_ABL_INTERP_NOPTS)
    if (a.evalmode >= 0) return;
#endif

    const int grp = lane / GP::G, q = lane % GP::G;
    const bool lane_active = q < GP::W1;
    const int comp = q % NC, j1 = (q / NC) % L;
    T* strip = strip_wave + grp * (D * L);
    WindowEval<T, NC, D, M, GP::G, OTHERK> we;
    we.init(a, q);
    __syncthreads();

    const PointRec<T, D>* sorted = static_cast<const PointRec<T, D>*>(a.sorted);
    T* vout = a.vout[comp_id];
    // The point loop wants run-time strides even in the FIXED variant: with immediates the compiler pairs
    // the reads of rows j2, j2 + 1 into ds_read2_b64, whose two addresses share banks when the row
    // stride is not bank-aligned (measured 30 % slower on f64 / M = 4).
    int RSp = RS, PSp = PS;
    asm volatile("" : "+s"(RSp), "+s"(PSp));

    for (;;) {
        int item = 0;
        if (lane == 0) item = atomicAdd(next_item, 1);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= nitems) break;
        const uint2 pr = items[item];
        const uint32_t p0 = pr.x, p1 = pr.y;
        if (p0 >= p1) continue;
        PointRec<T, D> rec = sorted[min(p0 + (uint32_t)grp, p1 - 1)];
        for (uint32_t pc = p0; pc < p1; pc += GP::PPW) {
            const uint32_t p = pc + grp;
            const bool have = p < p1;
            const uint32_t npc = pc + GP::PPW;
            PointRec<T, D> recn = rec;
            if (npc < p1) recn = sorted[min(npc + (uint32_t)grp, p1 - 1)];     // prefetch the next chunk
            int s[3] = {0, 0, 0};
            T X[3] = {T(0), T(0), T(0)};
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int c = cell_of(rec.r[d], g.Nover[d]);
                X[d] = rec.r[d] - T(c);
                s[d] = c - org[d];                 // first stencil node in padded-tile coordinates
            }
            // Real data with 8 or 16 lanes per point: the window values stay in the registers of the lanes that
            // evaluated them (value k = d * 2M + j in slot k / G of lane k % G of the group) and reach the
            // other lanes of the group by DPP row broadcasts — no LDS strip, no LDS waits around it.
            constexpr bool REGW = NUFFT_INTERP_REGW && !CPLX && D >= 2 && (GP::G == 8 || GP::G == 16);
            T wv[WindowEval<T, NC, D, M, GP::G, OTHERK>::NSLOT];
            T w1;
            if constexpr (REGW) {
                we.eval_regs(a, X, wv);
                w1 = wv[0];                                   // NC = 1: lane q < 2M owns w1[q]
            } else {
                wave_lds_fence();
#if !defined(NUFFT_ABL_NO_EVAL)
                we.eval_to_strip(a, X, strip, q);
#endif
                wave_lds_fence();
                w1 = strip[j1];
            }
            auto wfetch = [&](int d, int j) -> T {           // w_d[j] of this lane's point (d, j constants after unrolling)
                const int kk = d * L + j;
                const T x = wv[kk / GP::G];
                if constexpr (GP::G == 16) return row_bcast(x, kk % GP::G);
                const T lo = row_bcast(x, kk % GP::G), hi = row_bcast(x, kk % GP::G + 8);
                return (lane & 8) ? hi : lo;
            };
#if defined(NUFFT_ABL_INTERP_SAMEADDR)
            for (int d = 0; d < D; ++d) s[d] = __builtin_amdgcn_readfirstlane(s[d]);
#endif
            const T* base = tile + (s[0] + j1) * NC + comp + s[1] * RSp + s[2] * PSp;
            T acc = T(0);
#if defined(NUFFT_ABL_INTERP_NOREAD)
            if (have && lane_active) acc = base[0] * w1;
#else
            if constexpr (REGW) {
                // every lane runs the loop (the broadcasts need the whole wave); inactive lanes read valid
                // addresses of the tile (their record repeats the last point) and are masked at the end
                if constexpr (D == 2) {
#pragma unroll
                    for (int j2 = 0; j2 < L; ++j2) acc = fma(base[j2 * RSp], wfetch(1, j2), acc);
                } else if constexpr (FIXED && NUFFT_INTERP_ASM_READS && L <= 15 &&
                                     (L - 1) * FD.row_stride * (FD.n[1] + L) * (int)sizeof(T) < 65536) {
                    T w2[L];
#pragma unroll
                    for (int j = 0; j < L; ++j) w2[j] = wfetch(1, j);
                    constexpr int RB = FD.row_stride * (int)sizeof(T);
                    constexpr int PB = FD.row_stride * (FD.n[1] + L - 1) * (int)sizeof(T);
                    const uint32_t baddr = (uint32_t)(uintptr_t)base;       // LDS byte address of this lane's corner
                    // software pipeline over groups of R rows, two groups in flight
                    constexpr int R = 2;                     // 2M is even; larger groups spill registers (measured)
                    constexpr int GPP = L / R;
                    T pa[R], pb[R];
                    lds_read_rows<T, R, 0, RB>(pa, baddr, std::make_integer_sequence<int, R>{});
                    T t2 = T(0);
                    auto group_fma = [&](T (&rows)[R], int gidx) {
                        const int j3 = gidx / GPP, r0 = (gidx % GPP) * R;
#pragma unroll
                        for (int r = 0; r < R; ++r) t2 = fma(rows[r], w2[r0 + r], t2);
                        if (gidx % GPP == GPP - 1) {
                            acc = fma(t2, wfetch(2, j3), acc);
                            t2 = T(0);
                        }
                    };
                    interp_row_groups<T, L, R, PB, RB, 0>(pa, pb, baddr, group_fma);
                } else {
                    T w2[L];
#pragma unroll
                    for (int j = 0; j < L; ++j) w2[j] = wfetch(1, j);
#pragma unroll
                    for (int j3 = 0; j3 < L; ++j3) {
                        const T* plane = base + j3 * PSp;
                        T t2 = T(0);
#pragma unroll
                        for (int j2 = 0; j2 < L; ++j2) t2 = fma(plane[j2 * RSp], w2[j2], t2);
                        acc = fma(t2, wfetch(2, j3), acc);
                    }
                }
                acc = (have && lane_active) ? acc * w1 : T(0);
            } else if (have && lane_active) {
                if constexpr (D == 1) {
                    acc = base[0];
                } else if constexpr (D == 2) {
#pragma unroll
                    for (int j2 = 0; j2 < L; ++j2) acc = fma(base[j2 * RSp], strip[L + j2], acc);
                } else {
                    T w2[L];
#pragma unroll
                    for (int j = 0; j < L; ++j) w2[j] = strip[L + j];
#pragma unroll
                    for (int j3 = 0; j3 < L; ++j3) {
                        const T* plane = base + j3 * PSp;
                        T t2 = T(0);
#pragma unroll
                        for (int j2 = 0; j2 < L; ++j2) t2 = fma(plane[j2 * RSp], w2[j2], t2);
                        acc = fma(t2, strip[2 * L + j3], acc);
                    }
                }
                acc *= w1;
            }
#endif
            acc = group_sum<T, GP::G, CPLX>(acc);
            if (have && q < NC) {
                T res = acc * a.prefactor;
                if constexpr (OTHERK) { if (a.weights) res *= a.weights[rec.idx]; }   // callbacks.nonuniform(v, n), src/interpolation/gpu.jl:254
                vout[(int64_t)rec.idx * NC + q] = res;
            }
            rec = recn;
        }
    }
}

}  // namespace nufft
