// Device-side helpers shared by the HIP kernels (gfx950 / CDNA4, wave64).
#pragma once

#include <hip/hip_runtime.h>

#include "nufft_mi355x.h"
#include <cstdint>
#include <utility>

namespace nufft {

constexpr int kWave = 64;
constexpr int kMaxCompPerLaunch = 8;    // components (ntransforms) handled by one launch

template <typename T> struct TwoPi;
template <> struct TwoPi<float>  { static constexpr float  value = 6.28318530717958647692f; };
template <> struct TwoPi<double> { static constexpr double value = 6.28318530717958647692; };

// One bin-sorted point: coordinates in grid units r_d = (x_d / 2π) Ñ_d ∈ [0, Ñ_d] of the folded
// point, plus its index in the caller's arrays.  Power-of-two sized so that the scatter pass of
// the bin sort writes one aligned record per point.
template <typename T, int D>
struct alignas((D * sizeof(T) + 4 > 8) ? 16 : 8) PointRec {
    T r[D];
    int32_t idx;
};
static_assert(sizeof(PointRec<double, 3>) == 32, "record layout");
static_assert(sizeof(PointRec<float, 3>) == 16, "record layout");
static_assert(sizeof(PointRec<double, 1>) == 16, "record layout");
static_assert(sizeof(PointRec<float, 1>) == 8, "record layout");

// Geometry passed by value to every kernel.
//
// Points are bin-sorted by *fine bins* of b_d cells (b_d a power of two, 4 by default); the bin index
// runs with dimension 1 fastest.  Spreading and interpolation then tile the grid independently, with
// tile edges that are multiples of the bin size:
//   * spreading tile: the INTERIOR only lives in LDS ("output-driven": the workgroup visits every point
//     whose stencil touches its tile, clips the stencil, and stores the finished tile with plain
//     coalesced stores — no global atomics, no zero fill of the grid);
//   * interpolation tile: interior + (2M-1) halo in LDS, every point visited exactly once.
struct TileShape {
    int n[3];          // interior cells per dimension
    int nt[3];         // tiles per dimension
    int row_stride;    // LDS row stride in reals
    int plane_stride;  // row_stride * rows per plane
    int elems;         // LDS reals of the tile
    int ntiles;
    int max_items;     // upper bound of the work items (runs of the sorted array) of one tile
};

struct Geom {
    int Nover[3];      // oversampled grid
    int blog[3];       // log2 of the bin size
    int nb[3];         // bins per dimension
    int nbins;
    TileShape sp;      // spreading tile (interior only)
    TileShape ip;      // interpolation tile (n = interior; LDS holds n + 2M - 1 per dimension)
};

// to_unit_cell_gpu, reference src/blocking/blocking.jl:26-33.
template <typename T>
__device__ __forceinline__ T fold_to_unit_cell(T x) {
    const T L = TwoPi<T>::value;
    T r = fmod(x, L);
    r = (r == T(0)) ? T(0) : r;      // -0.0 -> +0.0
    return (r < T(0)) ? (L + r) : r;
}

// _transform_point_convention, reference src/abstractNFFTs.jl:147-155: AbstractNFFTs locations
// x ∈ [-1/2, 1/2) and opposite sign of the exponent -> this package's x ∈ [0, 2π).  Applied before the
// fold (point_transform_fold = to_unit_cell ∘ point_transform, src/set_points.jl:46-50).
template <typename T>
__device__ __forceinline__ T nfft_point_convention(T x) {
    const T L = TwoPi<T>::value;
    T t = L * x;
    t = -t;
    return t < T(0) ? t + L : t;
}

template <typename T>
__device__ __forceinline__ T transform_and_fold(T x, int point_transform) {
    if (point_transform == NUFFT_POINT_TRANSFORM_NFFT) x = nfft_point_convention(x);
    return fold_to_unit_cell(x);
}

// point_to_cell, reference src/Kernels/Kernels.jl:121-126: r = (x / L) * N in this order.
template <typename T>
__device__ __forceinline__ T to_grid_units(T x_folded, int N) {
    return (x_folded / TwoPi<T>::value) * T(N);
}

// Cell index (0-based) from grid units; clamps the (measure-zero) r == N case produced by
// rounding of L + r in the fold, which the reference leaves out of bounds.
template <typename T>
__device__ __forceinline__ int cell_of(T r, int N) {
    int i = (int)r;
    return i >= N ? N - 1 : i;
}

// XCD-aware tile order: blocks b and b + 8 share an XCD (and its L2), so give every XCD a
// contiguous chunk of the tile list.  Bijective for any number of tiles.
__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = b & 7, k = b >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// Contiguous runs of bins that cover the cell interval [lo, hi) of a periodic axis of N cells.
struct BinSegs {
    int n;          // number of runs (1 or 2)
    int lo[2];      // first bin of each run
    int len[2];     // bins in each run
    __device__ __forceinline__ int total() const { return len[0] + len[1]; }
    __device__ __forceinline__ int bin(int r) const { return r < len[0] ? lo[0] + r : lo[1] + (r - len[0]); }
};

__device__ __forceinline__ BinSegs bin_segments(int lo, int hi, int N, int blog, int nb) {
    BinSegs s;
    s.n = 1;
    s.lo[0] = 0; s.len[0] = nb; s.lo[1] = 0; s.len[1] = 0;
    if (hi - lo >= N) return s;                       // whole axis
    if (lo >= 0 && hi <= N) {                         // no wrap
        s.lo[0] = lo >> blog;
        s.len[0] = ((hi - 1) >> blog) - s.lo[0] + 1;
        return s;
    }
    // wraps once: [lo', N) and [0, hi')
    const int lo2 = lo < 0 ? lo + N : lo;
    const int hi2 = lo < 0 ? hi : hi - N;
    const int a_first = lo2 >> blog;                  // run A: a_first .. nb-1
    const int b_last = (hi2 - 1) >> blog;             // run B: 0 .. b_last
    if (b_last + 1 >= a_first) return s;              // runs touch or overlap: whole axis
    s.n = 2;
    s.lo[0] = a_first; s.len[0] = nb - a_first;
    s.lo[1] = 0;       s.len[1] = b_last + 1;
    return s;
}

// Same idea at a finer grain: chunks of `ch` consecutive slots stay together (neighbouring tiles share halo
// lines in one XCD's L2) but consecutive chunks go to different XCDs, so that a dense region of a
// non-uniform point set is shared by all XCDs instead of landing in one XCD's contiguous range.
// Bijective on [0, nblocks); ch = 0 selects xcd_remap.
__device__ __forceinline__ int xcd_remap_chunked(int b, int nblocks, int ch) {
    if (ch <= 0) return xcd_remap(b, nblocks);
    const int group = 8 * ch;
    const int full = nblocks / group * group;
    if (b >= full) return b;
    const int xcd = b & 7, k = b >> 3;
    return ((k / ch) * 8 + xcd) * ch + k % ch;
}

// Workgroup barrier for LDS data only.  __syncthreads() is a workgroup-scope fence + barrier: the compiler puts s_waitcnt vmcnt(0) in front of
// s_barrier, i.e. every wave waits until its outstanding GLOBAL stores have been acknowledged (on gfx9 stores count in vmcnt) — in the
// z-marching kernels that is the latency of the retire pass's stores (or of the scattered value stores) once per bin layer, for data no
// wave of the kernel ever reads back.  This barrier waits for the wave's LDS traffic only; loads that are still in flight are waited
// for where their registers are used (the compiler keeps counting them across the asm).  NUFFT_LDS_BARRIER=0: __syncthreads() (A/B builds).
#ifndef NUFFT_LDS_BARRIER
#define NUFFT_LDS_BARRIER 1
#endif
__device__ __forceinline__ void lds_barrier() {
#if NUFFT_LDS_BARRIER
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
    __syncthreads();
#endif
}

// Order LDS traffic of one wave without a workgroup barrier: LDS instructions of a wave complete
// in issue order; this only stops the compiler from moving accesses across the point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr __host__ __device__ int next_pow2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

constexpr __host__ __device__ int round_up(int x, int a) { return (x + a - 1) / a * a; }

// LDS layout shared by host (size computation) and kernels:  [tile | per-wave window strips].
// The spreading tile always accumulates in Float64 (ds_add_f32 is ~22x slower than ds_add_f64 on
// gfx950: 193 vs 8.5 cycles per wave instruction, scripts/microbench.hip); the interpolation tile holds
// the grid's own precision.
constexpr __host__ __device__ int lanes_per_point(int ncomp, int M) { return next_pow2(ncomp * 2 * M); }

//   [tile | work-item table (p0, p1 pairs) + fetch counter | per-wave window strips]
struct LdsLayout {
    int tile_bytes, items_bytes, strip_bytes_per_wave, total;
};

// Work items per tile the kernels aim for: with fewer runs than this (1-D tiles have one or two, a 3-D
// interpolation tile one per wave) the runs are split so that the 16 waves of a workgroup share the points.
constexpr int kItemTarget = 64;

// Spreading strips of real 3-D plans carry one zero in front of and behind the 2M window values of every
// dimension (the cube accumulation of spread_tile_kernel clamps its window index into them).
constexpr __host__ __device__ int spread_strip_pad(int D, int ncomp) { return (D == 3 && ncomp == 1) ? 1 : 0; }

constexpr __host__ __device__ LdsLayout lds_layout(int tile_elems, int tile_elem_bytes, int real_bytes, int D, int M,
                                                   int ncomp, int nwaves, int max_items, int strip_pad = 0) {
    LdsLayout l{};
    l.tile_bytes = round_up(tile_elems * tile_elem_bytes, 16);
    l.items_bytes = round_up(max_items * 8 + 16, 16);
    const int ppw = kWave / lanes_per_point(ncomp, M);           // points a wave works on at once
    l.strip_bytes_per_wave = round_up(ppw * D * (2 * M + 2 * strip_pad) * real_bytes, 16);
    l.total = l.tile_bytes + l.items_bytes + nwaves * l.strip_bytes_per_wave;
    return l;
}

// Upper bound of the runs of the sorted array (work items before splitting) a tile looks up, from the same
// arithmetic as the kernels.  Spreading (output-driven): the bins that cover the cells [org - M, org + n + M - 1)
// of dimensions 2 and 3 — n / b + ceil(M / b) + floor((M - 2) / b) + 1 bin rows for a tile edge n that is a multiple
// of the bin edge b (at most all nb bins of the axis) — times the two runs of a tile that wraps around dimension 1.
// Interpolation: the tile's own bins.  nb = nullptr: grid unknown (compile-time tiles).
constexpr __host__ __device__ int tile_bin_rows_bound(bool spreading, int n, int b, int M) {
    if (!spreading) return (n + b - 1) / b + 1;
    return (n + b - 1) / b + (M + b - 1) / b + (M >= 2 ? (M - 2) / b : 0) + 1;
}
constexpr __host__ __device__ long tile_items_bound(bool spreading, int D, int M, int b, const int (&n)[3], const int* nb) {
    long items = spreading ? 2 : 1;
    for (int d = 1; d < D; ++d) {
        long rows = tile_bin_rows_bound(spreading, n[d], b, M);
        if (nb && rows > nb[d]) rows = nb[d];
        items *= rows;
    }
    return items + kItemTarget;      // room for splitting long runs (split_work_items)
}
constexpr int kMaxTileItems = 4096;

// Compile-time interpolation tile for large grids (every edge shorter than the axis), 1024 threads and
// the full 160 KiB of LDS: with constant row/plane strides all 2M x 2M LDS reads of a point use immediate
// offsets from one base address.  Same cost model as the run-time search in plan_math.cpp (halo
// amplification of the tile load); n[0] == 0: nothing fits with 4-cell bins.
// LDS row stride (in reals) of a tile whose wave instructions touch `stencil_inner` contiguous reals in
// each of several consecutive rows: the rows land on disjoint banks when the stride is congruent to the
// stencil width modulo the 128-byte half of the bank period.
constexpr __host__ __device__ int padded_row_stride(int inner_elems, int stencil_inner, int real_bytes) {
    const int period = 128 / real_bytes;
    if (stencil_inner >= period) return inner_elems;
    int s = inner_elems;
    while (s % period != stencil_inner % period) ++s;
    return s;
}

struct FixedTileDims { int n[3]; int row_stride; int plane_stride; };

// Plane stride (in Float64 reals) of the spreading tile.  Real 3-D tiles are padded so that the stride is 2 (mod 32):
// the cube accumulation of spread_tile_kernel adds 4 x 4 x 4 cubes (lane = 16 x + 4 y + z) with one ds_add_f64, and the
// LDS serves 16 consecutive lanes at a time — one x, all (y, z): their doubles sit at z * plane + y * row, which with row
// strides of 8 or 24 (mod 32) and a plane stride of 2 (mod 32) are 16 different bank pairs.  With the natural stride
// rows * row_stride (0 mod 32 at C2) the four z planes collide (measured: kernel 10.0 ms; stride 4 mod 32: 6.2 ms).
constexpr __host__ __device__ int spread_plane_stride(int row_stride, int rows, int D, int ncomp) {
    int ps = row_stride * rows;
    if (D == 3 && ncomp == 1)
        while (ps % 32 != 2) ++ps;
    return ps;
}
constexpr __host__ __device__ FixedTileDims fixed_interp_tile(int elem_bytes, int ncomp, int D, int M) {
    const int L = 2 * M, halo = L - 1, b = 4, nwaves = 16;
    const int cap = D == 1 ? 8192 : 96;
    FixedTileDims best{{0, D >= 2 ? 0 : 1, D >= 3 ? 0 : 1}, 0, 0};
    double best_cost = 1e300;
    for (int n3 = (D >= 3 ? b : 1); n3 <= (D >= 3 ? cap : 1); n3 += b)
        for (int n2 = (D >= 2 ? b : 1); n2 <= (D >= 2 ? cap : 1); n2 += b)
            for (int n1 = b; n1 <= cap; n1 += b) {
                const int rs = ncomp * (n1 + halo);      // rows unpadded: LDS capacity beats bank alignment here
                const long elems = (long)rs * (D >= 2 ? n2 + halo : 1) * (D >= 3 ? n3 + halo : 1);
                const int nn[3] = {n1, n2, n3};
                const long items = tile_items_bound(false, D, M, b, nn, nullptr);
                if (elems * elem_bytes > 163840 || items > kMaxTileItems ||
                    lds_layout((int)elems, elem_bytes, elem_bytes, D, M, ncomp, nwaves, (int)items).total > 163840 - 256) break;
                double cost = (double)(n1 + halo) / n1;
                if (D >= 2) cost *= (double)(n2 + halo) / n2;
                if (D >= 3) cost *= (double)(n3 + halo) / n3;
                cost -= 1e-6 * n1;
                if (cost < best_cost) {
                    best_cost = cost; best.n[0] = n1; best.n[1] = n2; best.n[2] = n3; best.row_stride = rs;
                    best.plane_stride = rs * (D >= 2 ? n2 + halo : 1);
                }
            }
    return best;
}

// Compile-time spreading tile (interior only, Float64 accumulation) under the same conditions and with the
// same cost model as the run-time search (point visits per point); row stride padded for the LDS banks.
constexpr __host__ __device__ FixedTileDims fixed_spread_tile(int real_bytes, int ncomp, int D, int M) {
    const int L = 2 * M, halo = L - 1, b = 4, nwaves = 16;
    const int cap = D == 1 ? 8192 : 96;
    FixedTileDims best{{0, D >= 2 ? 0 : 1, D >= 3 ? 0 : 1}, 0, 0};
    double best_cost = 1e300;
    for (int n3 = (D >= 3 ? b : 1); n3 <= (D >= 3 ? cap : 1); n3 += b)
        for (int n2 = (D >= 2 ? b : 1); n2 <= (D >= 2 ? cap : 1); n2 += b)
            for (int n1 = b; n1 <= cap; n1 += b) {
                const int rs = D >= 2 ? padded_row_stride(ncomp * n1, ncomp * L, 8) : ncomp * n1;
                const int ps = spread_plane_stride(rs, D >= 2 ? n2 : 1, D, ncomp);
                const long elems = (long)ps * (D >= 3 ? n3 : 1);
                const int nn[3] = {n1, n2, n3};
                const long items = tile_items_bound(true, D, M, b, nn, nullptr);
                if (elems * 8 > 163840 || items > kMaxTileItems ||
                    lds_layout((int)elems, 8, real_bytes, D, M, ncomp, nwaves, (int)items, spread_strip_pad(D, ncomp)).total > 163840 - 256) break;
                double cost = (double)(n1 + halo) / n1;
                if (D >= 2) cost *= (double)(n2 + halo) / n2;
                if (D >= 3) cost *= (double)(n3 + halo) / n3;
                cost -= 1e-6 * n1;
                if (cost < best_cost) { best_cost = cost; best.n[0] = n1; best.n[1] = n2; best.n[2] = n3; best.row_stride = rs; best.plane_stride = ps; }
            }
    return best;
}

// ---------------------------------------------------------------------------------------------
// Window evaluation (backwards Kaiser-Bessel)
// ---------------------------------------------------------------------------------------------

// Direct: reference src/Kernels/kaiser_bessel_backwards.jl:158-175.
// j is 0-based (reference j = 1..2M), X ∈ [0, 1].
//
// Double precision uses its own square root, exponential and division (each good to an ulp or two) instead of the
// library's: the correctly rounded / special-case-proof versions cost 20 + 30 + 12 instructions per window value,
// these 9 + 19 + 6 — Direct() is the reference's ROC default, so this is the headline path.

// p <- p * h + c as ONE v_fma_f64 with the coefficient in a scalar register pair (the compiler otherwise keeps the
// coefficients in vector registers and emits v_mov_b64 + v_fmac_f64 per Horner step)
__device__ __forceinline__ double fma_sc(double p, double h, double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(h), "s"(c));
    return r;
}

// exp(x) for 0 <= x < 700 without the library's special-case handling: x = k ln2 + r, |r| <= ln2 / 2,
// exp(r) = (exp(r / 2))^2 with a degree-11 Taylor polynomial of exp(r / 2) (|r / 2| <= 0.174: truncation
// 2e-18), scaled by 2^k.  Relative error ~2e-16.  Float: the library exp.
__device__ __forceinline__ double exp_pos(double x) {
    const double k = rint(x * 1.4426950408889634074);              // log2(e)
    double r = fma(k, -6.93147180369123816490e-01, x);              // ln2 high part
    r = fma(k, -1.90821492927058770002e-10, r);                     // ln2 low part
    const double h = 0.5 * r;
    double p = 1.0 / 39916800.0;
    p = fma_sc(p, h, 1.0 / 3628800.0);
    p = fma_sc(p, h, 1.0 / 362880.0);
    p = fma_sc(p, h, 1.0 / 40320.0);
    p = fma_sc(p, h, 1.0 / 5040.0);
    p = fma_sc(p, h, 1.0 / 720.0);
    p = fma_sc(p, h, 1.0 / 120.0);
    p = fma_sc(p, h, 1.0 / 24.0);
    p = fma_sc(p, h, 1.0 / 6.0);
    p = fma(p, h, 0.5);
    p = fma(p, h, 1.0);
    p = fma(p, h, 1.0);
    return ldexp(p * p, (int)k);
}
// Float32: the same range reduction, then the hardware exp2 on the small remainder (|r| <= 0.35: one ulp)
__device__ __forceinline__ float exp_pos(float x) {
    const float k = rintf(x * 1.44269504088896340736f);
    float r = fmaf(k, -6.93145751953125e-1f, x);                    // ln2 high part (exact product for |k| < 2^12)
    r = fmaf(k, -1.42860682030941723212e-6f, r);                    // ln2 low part
    return ldexpf(__builtin_amdgcn_exp2f(r * 1.44269504088896340736f), (int)k);
}

// sqrt(z) for 0 <= z <= 1: v_rsq_f64 (26 bits) + two coupled Newton steps (Goldschmidt); z = 0 is kept away from the
// infinite reciprocal root by a floor far below anything the window needs (sqrt(1e-280) = 1e-140)
__device__ __forceinline__ double sqrt_unit(double z) {
    z = fmax(z, 1e-280);
    const double r = __builtin_amdgcn_rsq(z);
    double g = z * r, h = 0.5 * r;
    double e = fma(-h, g, 0.5);
    g = fma(g, e, g);
    h = fma(h, e, h);
    e = fma(-g, g, z);
    return fma(e, h, g);
}
__device__ __forceinline__ float sqrt_unit(float z) { return __builtin_amdgcn_sqrtf(z > 0.f ? z : 0.f); }   // v_sqrt_f32: one ulp

// n / d for normal, positive d: v_rcp_f64 + two Newton steps on the reciprocal + one correction of the quotient
__device__ __forceinline__ double div_pos(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}
__device__ __forceinline__ float div_pos(float n, float d) { return n * __builtin_amdgcn_rcpf(d); }   // v_rcp_f32: one ulp

// sinh(x) / x for x >= 0 without the library sinh (ocml's double sinh costs ~2x the whole rest of the
// evaluation): one exp and one division above 0.5, the even Taylor series below (no cancellation).
template <typename T>
__device__ __forceinline__ T sinh_over_x(T x) {
    if (x < T(0.5)) {
        const T z = x * x;
        T p = T(1.0 / 1307674368000.0);                       // 1/15!
        p = fma(p, z, T(1.0 / 6227020800.0));                 // 1/13!
        p = fma(p, z, T(1.0 / 39916800.0));                   // 1/11!
        p = fma(p, z, T(1.0 / 362880.0));                     // 1/9!
        p = fma(p, z, T(1.0 / 5040.0));                       // 1/7!
        p = fma(p, z, T(1.0 / 120.0));                        // 1/5!
        p = fma(p, z, T(1.0 / 6.0));                          // 1/3!
        return fma(p, z, T(1));
    }
    const T e = exp_pos(x);
    if constexpr (sizeof(T) == 4) {
        // Float32: e * e overflows for x > 44 (beta = 46.9 at M = 10, sigma = 2)
        return div_pos(T(0.5) * (e - div_pos(T(1), e)), x);
    }
    return div_pos(T(0.5) * fma(e, e, T(-1)), e * x);         // (e - 1/e) / (2x) with a single division
}

template <typename T, int M>
__device__ __forceinline__ T bkb_direct(T X, int j, T beta, T beta_over_pi) {
    const T y = (T(M - 1 - j) + X) / T(M);
    const T z = T(1) - y * y;
    const T s = sqrt_unit(z > T(0) ? z : T(0));
    return sinh_over_x(beta * s) * beta_over_pi;
}

// FastApproximation: Horner evaluation of the degree-(M+3) piecewise polynomial,
// reference src/Kernels/piecewise_polynomial.jl:76-92.  cs points to [npoly][2M] for one dimension.
template <typename T, int M>
__device__ __forceinline__ T bkb_poly(T X, int j, const T* cs) {
    constexpr int NP = M + 4;
    constexpr int L = 2 * M;
    const T x = T(2) * X - T(1);
    T y = cs[(NP - 1) * L + j];
#pragma unroll
    for (int k = NP - 2; k >= 0; --k) y = fma(x, y, cs[k * L + j]);
    return y;
}

// ---------------------------------------------------------------------------------------------
// Wave-level sum over lanes whose index differs in the bits of MASKS (butterfly with DPP and
// gfx950 permlane swaps; no LDS traffic).  Every participating lane ends with the full sum.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_move(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    const long long r = ((long long)hi << 32) | (unsigned int)lo;
    return __builtin_bit_cast(double, r);
}

// Broadcast of lane J (0..15) of every 16-lane row to the whole row: DPP row_newbcast, one VALU instruction
// also for 64-bit data (v_mov_b64_dpp; the double-precision ALU accepts only this DPP control).
template <int J>
__device__ __forceinline__ float row_bcast_c(float x) {
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xF, 0xF, false));
}
template <int J>
__device__ __forceinline__ double row_bcast_c(double x) {
    const long v = __builtin_bit_cast(long, x);
    return __builtin_bit_cast(double, (long)__builtin_amdgcn_update_dpp(v, v, 0x150 + J, 0xF, 0xF, false));
}
// The same as inline assembly with a separate destination: the builtin ties the destination to its `old`
// operand, which costs a register copy per broadcast.  A DPP instruction must not read a VGPR written by
// a VALU instruction in the two preceding wait states; the compiler's hazard recogniser does not look into
// inline assembly, hence the s_nop (two wait states of the issuing wave only).
template <int J>
__device__ __forceinline__ float row_bcast_asm(float x) {
    float r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(J));
    return r;
}
template <int J>
__device__ __forceinline__ double row_bcast_asm(double x) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(J));
    return r;
}
// all values 0 .. N-1 (N <= 16) of a row into w[], N instructions
template <typename T, int N, int... J>
__device__ __forceinline__ void row_bcast_all(T x, T (&w)[N], int base, std::integer_sequence<int, J...>) {
    ((w[base + J] = row_bcast_asm<J>(x)), ...);
}
// the common case N = 8 as one block: a single hazard no-op, nothing can be scheduled in between
// (early-clobber outputs: none of them may alias the source)
__device__ __forceinline__ void row_bcast_all(double x, double (&w)[8], int, std::integer_sequence<int, 0, 1, 2, 3, 4, 5, 6, 7>) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b64_dpp %0, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %4, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %5, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %6, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %7, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]), "=&v"(w[4]), "=&v"(w[5]), "=&v"(w[6]), "=&v"(w[7])
        : "v"(x));
}

// four values BASE .. BASE + 3 of a row as one block (a single hazard no-op)
template <int BASE>
__device__ __forceinline__ void row_bcast4(double x, double (&w)[4]) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b64_dpp %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %1, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %2, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b64_dpp %3, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
        : "v"(x), "n"(BASE), "n"(BASE + 1), "n"(BASE + 2), "n"(BASE + 3));
}
template <int BASE>
__device__ __forceinline__ void row_bcast4(float x, float (&w)[4]) {
    asm volatile(
        "s_nop 1\n\t"
        "v_mov_b32_dpp %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %1, %4 row_newbcast:%6 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %2, %4 row_newbcast:%7 row_mask:0xf bank_mask:0xf\n\t"
        "v_mov_b32_dpp %3, %4 row_newbcast:%8 row_mask:0xf bank_mask:0xf"
        : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3])
        : "v"(x), "n"(BASE), "n"(BASE + 1), "n"(BASE + 2), "n"(BASE + 3));
}

// Groups of 8 lanes: lane J of its own group to every lane — lanes 0-7 of a row take lane J, lanes 8-15 lane J + 8 of the row: two DPP moves
// that each write one half of the row (bank_mask), instead of two full broadcasts and a select on (lane & 8).
template <int J>
__device__ __forceinline__ float half_bcast_asm(float x) {
    float r;
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3\n\tv_mov_b32_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xc"
                 : "=&v"(r) : "v"(x), "n"(J), "n"(J + 8));
    return r;
}
template <int J>
__device__ __forceinline__ double half_bcast_asm(double x) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0x3\n\tv_mov_b64_dpp %0, %1 row_newbcast:%3 row_mask:0xf bank_mask:0xc"
                 : "=&v"(r) : "v"(x), "n"(J), "n"(J + 8));
    return r;
}
// j (0..7) must be a compile-time constant after unrolling
template <typename T>
__device__ __forceinline__ T half_bcast(T x, int j) {
    switch (j & 7) {
        case 0: return half_bcast_asm<0>(x);   case 1: return half_bcast_asm<1>(x);
        case 2: return half_bcast_asm<2>(x);   case 3: return half_bcast_asm<3>(x);
        case 4: return half_bcast_asm<4>(x);   case 5: return half_bcast_asm<5>(x);
        case 6: return half_bcast_asm<6>(x);   default: return half_bcast_asm<7>(x);
    }
}

// j must be a compile-time constant after unrolling (the switch folds away)
template <typename T>
__device__ __forceinline__ T row_bcast(T x, int j) {
    switch (j & 15) {
        case 0: return row_bcast_asm<0>(x);   case 1: return row_bcast_asm<1>(x);
        case 2: return row_bcast_asm<2>(x);   case 3: return row_bcast_asm<3>(x);
        case 4: return row_bcast_asm<4>(x);   case 5: return row_bcast_asm<5>(x);
        case 6: return row_bcast_asm<6>(x);   case 7: return row_bcast_asm<7>(x);
        case 8: return row_bcast_asm<8>(x);   case 9: return row_bcast_asm<9>(x);
        case 10: return row_bcast_asm<10>(x); case 11: return row_bcast_asm<11>(x);
        case 12: return row_bcast_asm<12>(x); case 13: return row_bcast_asm<13>(x);
        case 14: return row_bcast_asm<14>(x); default: return row_bcast_asm<15>(x);
    }
}

__device__ __forceinline__ float swap16_add(float x) {
    const int v = __builtin_bit_cast(int, x);
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ float swap32_add(float x) {
    const int v = __builtin_bit_cast(int, x);
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return __builtin_bit_cast(float, (int)r[0]) + __builtin_bit_cast(float, (int)r[1]);
}
__device__ __forceinline__ double swap16_add(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const long long a0 = ((long long)(int)rh[0] << 32) | (unsigned int)rl[0];
    const long long a1 = ((long long)(int)rh[1] << 32) | (unsigned int)rl[1];
    return __builtin_bit_cast(double, a0) + __builtin_bit_cast(double, a1);
}
__device__ __forceinline__ double swap32_add(double x) {
    const long long b = __builtin_bit_cast(long long, x);
    const int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const long long a0 = ((long long)(int)rh[0] << 32) | (unsigned int)rl[0];
    const long long a1 = ((long long)(int)rh[1] << 32) | (unsigned int)rl[1];
    return __builtin_bit_cast(double, a0) + __builtin_bit_cast(double, a1);
}

// Sum over the lanes of a group of G consecutive lanes (G a power of two ≤ 64).  With SKIP1 the
// lanes of even and odd index are summed separately (interleaved re/im lanes): every exchange
// then moves data by an even lane distance.  DPP controls: quad_perm [1,0,3,2] = 0xB1 (xor 1),
// [2,3,0,1] = 0x4E (xor 2), [3,2,1,0] = 0x1B, row_half_mirror = 0x141, row_ror:4 / :8 =
// 0x124 / 0x128 (rotations inside a row of 16; two of them visit all four quads).
template <typename T, int G, bool SKIP1>
__device__ __forceinline__ T group_sum(T x) {
    if constexpr (G >= 2 && !SKIP1) x += dpp_move<0xB1>(x);
    if constexpr (G >= 4) x += dpp_move<0x4E>(x);
    if constexpr (G == 8) {
        if constexpr (SKIP1) x += dpp_move<0x141>(dpp_move<0x1B>(x));   // exact xor 4
        else x += dpp_move<0x141>(x);
    }
    if constexpr (G >= 16) {
        x += dpp_move<0x124>(x);
        x += dpp_move<0x128>(x);
    }
    if constexpr (G >= 32) x = swap16_add(x);   // other row of the pair
    if constexpr (G >= 64) x = swap32_add(x);   // other half-wave
    return x;
}

}  // namespace nufft
