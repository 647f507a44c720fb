// Type-1 spreading and type-2 interpolation on LDS tiles (gfx950, wave64).
//
// Replaces the reference's shared-memory kernels
//   spread_from_points_shmem_kernel!    src/spreading/gpu.jl:237-377 (+ :381-434)
//   interpolate_to_points_shmem_kernel! src/interpolation/gpu.jl:211-328 (+ :331-395)
// with an MI355X-first design:
//   * one workgroup per tile of the oversampled grid; the padded tile (interior + 2M-1 halo)
//     lives in LDS (up to 160 KiB on gfx950, non-cubic tiles, bank-aware row stride);
//   * points arrive bin-sorted as aligned records {r_1..r_D, idx} (binsort.hip);
//   * each wave stages 16 points at a time: 4 lanes per point evaluate the D·2M window values
//     (direct sinh form or the piecewise polynomial) into a wave-private LDS strip;
//   * spreading: the wave then walks its staged points; for every point the lanes own one
//     (component, j1, j2) element of the stencil face and loop over j3, accumulating with
//     LDS float atomics (ds_add_f64 / ds_add_f32, no return) — no workgroup barrier per point
//     (the reference needs one, src/spreading/gpu.jl:354-360).  The finished tile is flushed to
//     HBM with global float atomics in row-contiguous wave instructions, skipping exact zeros;
//   * interpolation: the padded tile is loaded once (coalesced rows, periodic wrap), then every
//     point is gathered by a whole wave (conflict-free row reads) and reduced with DPP /
//     permlane-swap butterflies; results leave through the staging strip in one scattered store.
#pragma once

#include <hip/hip_runtime.h>

#include "device_common.h"
#include "nufft_mi355x.h"

namespace nufft {

template <typename T>
struct TileArgs {
    Geom g;
    const void* sorted;
    const uint32_t* offsets;
    const T* coefs;                       // [D][npoly][2M]
    T beta[3];
    T bop[3];                             // β/π times the power-of-two window normalisation
    T* grid[kMaxCompPerLaunch];           // component grids (as arrays of reals)
    const T* vin[kMaxCompPerLaunch];      // spread: values (as reals; complex = interleaved)
    T* vout[kMaxCompPerLaunch];           // interp
    T prefactor;
    int evalmode;
};

// Per-lane description of the stencil face element(s) a lane owns.
template <int NC, int D, int M>
struct Face {
    static constexpr int L = 2 * M;
    static constexpr int W1 = NC * L;                         // inner extent in reals
    static constexpr int FACE = W1 * (D >= 2 ? L : 1);
    static constexpr int G = FACE >= kWave ? kWave : next_pow2(FACE);   // lanes per point
    static constexpr int PPW = kWave / G;                     // points processed at once
    static constexpr int NPASS = (FACE + G - 1) / G;
};

// Wave-private staging strip.
template <typename T, int NC, int D, int M>
struct Stage {
    static constexpr int NV = D * 2 * M;
    T* wv;     // [kCH][NV]   window values
    T* vv;     // [kCH][NC]   input values (spread) / results (interp)
    int* ss;   // [kCH][D]    local stencil start
    __device__ Stage(unsigned char* base) {
        wv = reinterpret_cast<T*>(base);
        vv = wv + kCH * NV;
        ss = reinterpret_cast<int*>(vv + kCH * NC);
    }
};

// Loads the chunk's records, evaluates the windows and fills the staging strip.
// Returns the original index of this lane's point (valid for lanes with part == 0 and pt < npts).
template <typename T, int NC, int D, int M>
__device__ __forceinline__ int stage_chunk(const TileArgs<T>& a, const PointRec<T, D>* __restrict__ sorted,
                                           uint32_t first, int npts, const int (&origin)[3], const T* coefs_lds,
                                           Stage<T, NC, D, M>& st, int lane) {
    constexpr int L = 2 * M;
    constexpr int NV = D * L;
    const int pt = lane % kCH;
    const int part = lane / kCH;
    int idx = -1;
    if (pt < npts) {
        const PointRec<T, D> rec = sorted[first + pt];
        idx = rec.idx;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = cell_of(rec.r[d], a.g.Nover[d]);
            const T X = rec.r[d] - T(i);
            if (part == 0) st.ss[pt * D + d] = i - origin[d];
            if (a.evalmode == NUFFT_EVAL_DIRECT) {
                const T beta = a.beta[d];
                const T bop = a.bop[d];
                for (int j = part; j < L; j += kParts) st.wv[pt * NV + d * L + j] = bkb_direct<T, M>(X, j, beta, bop);
            } else {
                const T* cs = coefs_lds + d * (M + 4) * L;
                for (int j = part; j < L; j += kParts) st.wv[pt * NV + d * L + j] = bkb_poly<T, M>(X, j, cs);
            }
        }
    }
    return idx;
}

template <typename T>
__device__ __forceinline__ void lds_atomic_add(T* p, T v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <typename T>
__device__ __forceinline__ void global_atomic_add(T* p, T v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Row walker for tile <-> global traffic: a wave instruction covers RPW rows of the padded tile
// (row = fixed l2, l3; W_row = NC * P[0] reals), lanes running along dimension 1 so that global
// addresses are contiguous within a row.
struct RowWalker {
    int lanes_per_row, rpw, sub, lane_in_row;
    int rows_total, row, l2, l3, step, step2, step3;
    __device__ RowWalker(const Geom& g, int w_row, int wave, int nwaves, int lane) {
        int lpr = kWave;
        while (lpr / 2 >= w_row && lpr > 1) lpr >>= 1;
        lanes_per_row = lpr;
        rpw = kWave / lpr;
        sub = lane / lpr;
        lane_in_row = lane % lpr;
        rows_total = g.P[1] * g.P[2];
        row = wave * rpw + sub;
        l2 = row % g.P[1];
        l3 = row / g.P[1];
        step = nwaves * rpw;
        step2 = step % g.P[1];
        step3 = step / g.P[1];
    }
    __device__ __forceinline__ bool valid() const { return row < rows_total; }
    __device__ __forceinline__ void next(const Geom& g) {
        row += step;
        l2 += step2;
        l3 += step3;
        if (l2 >= g.P[1]) { l2 -= g.P[1]; l3 += 1; }
    }
};

__device__ __forceinline__ int wrap_index(int gidx, int N) {
    if (gidx < 0) gidx += N;
    if (gidx >= N) gidx -= N;
    if (gidx >= N) gidx -= N;
    return gidx;
}

// ---------------------------------------------------------------------------------------------
// Spreading
// ---------------------------------------------------------------------------------------------
template <typename T, bool CPLX, int D, int M>
__global__ __launch_bounds__(1024) void spread_tile_kernel(TileArgs<T> a) {
    constexpr int NC = CPLX ? 2 : 1;
    constexpr int L = 2 * M;
    using F = Face<NC, D, M>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const int nthreads = blockDim.x;
    const int nwaves = nthreads / kWave;
    const Geom& g = a.g;

    const int tile_id = xcd_remap(blockIdx.x, g.ntiles);
    const uint32_t pa = a.offsets[tile_id];
    const uint32_t pb = a.offsets[tile_id + 1];
    if (pa == pb) return;   // nothing to spread: skip zeroing and flush (reference src/spreading/gpu.jl:364)

    const int comp_id = blockIdx.y;
    int t[3], origin[3];
    {
        int rem = tile_id;
        t[0] = rem % g.nt[0]; rem /= g.nt[0];
        t[1] = rem % g.nt[1]; rem /= g.nt[1];
        t[2] = rem;
#pragma unroll
        for (int d = 0; d < 3; ++d) origin[d] = t[d] * g.n[d];
    }

    using A = double;   // LDS accumulation type (see lds_layout)
    const LdsLayout lay = lds_layout(g.tile_elems, (int)sizeof(A), (int)sizeof(T), D, M, NC, nwaves);
    A* tile = reinterpret_cast<A*>(smem);
    T* coefs_lds = reinterpret_cast<T*>(smem + lay.tile_bytes);
    Stage<T, NC, D, M> st(smem + lay.tile_bytes + lay.coef_bytes + wave * lay.stage_bytes_per_wave);

    // zero the tile, copy the polynomial coefficients
    for (int i = tid; i < g.tile_elems; i += nthreads) tile[i] = A(0);
    if (a.evalmode != NUFFT_EVAL_DIRECT)
        for (int i = tid; i < D * (M + 4) * L; i += nthreads) coefs_lds[i] = a.coefs[i];
    __syncthreads();

    // lane roles on the stencil face
    const int gq = lane / F::G, q = lane % F::G;
    int lane_off[F::NPASS], j1v[F::NPASS], j2v[F::NPASS], cmp[F::NPASS];
    bool act[F::NPASS];
#pragma unroll
    for (int ps = 0; ps < F::NPASS; ++ps) {
        const int e = q + ps * F::G;
        act[ps] = e < F::FACE;
        const int e1 = e % F::W1;
        const int j2 = e / F::W1;
        cmp[ps] = e1 % NC;
        j1v[ps] = e1 / NC;
        j2v[ps] = j2;
        lane_off[ps] = j2 * g.row_stride + e1;
    }

    const PointRec<T, D>* sorted = static_cast<const PointRec<T, D>*>(a.sorted);
    const T* vin = a.vin[comp_id];
    const int npts_tile = (int)(pb - pa);
    const int nchunks = (npts_tile + kCH - 1) / kCH;

    for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
        const uint32_t first = pa + (uint32_t)chunk * kCH;
        const int npts = min(kCH, (int)(pb - first));
        wave_lds_fence();   // previous chunk's reads are done before the strip is overwritten
        const int idx = stage_chunk<T, NC, D, M>(a, sorted, first, npts, origin, coefs_lds, st, lane);
        if (lane < kCH && lane < npts) {
#pragma unroll
            for (int c = 0; c < NC; ++c) st.vv[lane * NC + c] = vin[(int64_t)idx * NC + c];
        }
        wave_lds_fence();

        for (int p0 = 0; p0 < npts; p0 += F::PPW) {
            const int pt = p0 + gq;
            if (pt < npts) {
                const T* wv = st.wv + pt * (D * L);
                int base = st.ss[pt * D + 0] * NC;
                if constexpr (D >= 2) base += st.ss[pt * D + 1] * g.row_stride;
                if constexpr (D >= 3) base += st.ss[pt * D + 2] * g.plane_stride;
#pragma unroll
                for (int ps = 0; ps < F::NPASS; ++ps) {
                    if (act[ps]) {
                        T w = st.vv[pt * NC + cmp[ps]] * wv[j1v[ps]];
                        if constexpr (D >= 2) w *= wv[L + j2v[ps]];
                        A* dst = tile + base + lane_off[ps];
                        if constexpr (D >= 3) {
#pragma unroll
                            for (int j3 = 0; j3 < L; ++j3) lds_atomic_add(dst + j3 * g.plane_stride, (A)(w * wv[2 * L + j3]));
                        } else {
                            lds_atomic_add(dst, (A)w);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();

    // flush: tile -> global grid with float atomics (add_from_local_to_global_memory!, :406-434)
    T* grid = a.grid[comp_id];
    const int w_row = NC * g.P[0];
    RowWalker rw(g, w_row, wave, nwaves, lane);
    const int o1 = origin[0] - (M - 1), o2 = origin[1] - (M - 1), o3 = origin[2] - (M - 1);
    for (; rw.valid(); rw.next(g)) {
        int64_t rowbase = 0;
        if constexpr (D >= 2) rowbase = (int64_t)wrap_index(o2 + rw.l2, g.Nover[1]);
        if constexpr (D >= 3) rowbase += (int64_t)wrap_index(o3 + rw.l3, g.Nover[2]) * g.Nover[1];
        rowbase *= (int64_t)g.Nover[0] * NC;
        const A* src = tile + rw.l2 * g.row_stride + rw.l3 * g.plane_stride;
        for (int e = rw.lane_in_row; e < w_row; e += rw.lanes_per_row) {
            const T v = (T)src[e];
            if (v != T(0)) {
                const int l1 = e / NC, c = e % NC;
                const int g1 = wrap_index(o1 + l1, g.Nover[0]);
                global_atomic_add(grid + rowbase + (int64_t)g1 * NC + c, v);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Interpolation
// ---------------------------------------------------------------------------------------------
template <typename T, bool CPLX, int D, int M>
__global__ __launch_bounds__(1024) void interp_tile_kernel(TileArgs<T> a) {
    constexpr int NC = CPLX ? 2 : 1;
    constexpr int L = 2 * M;
    using F = Face<NC, D, M>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;
    const int nthreads = blockDim.x;
    const int nwaves = nthreads / kWave;
    const Geom& g = a.g;

    const int tile_id = xcd_remap(blockIdx.x, g.ntiles);
    const uint32_t pa = a.offsets[tile_id];
    const uint32_t pb = a.offsets[tile_id + 1];
    if (pa == pb) return;

    const int comp_id = blockIdx.y;
    int t[3], origin[3];
    {
        int rem = tile_id;
        t[0] = rem % g.nt[0]; rem /= g.nt[0];
        t[1] = rem % g.nt[1]; rem /= g.nt[1];
        t[2] = rem;
#pragma unroll
        for (int d = 0; d < 3; ++d) origin[d] = t[d] * g.n[d];
    }

    const LdsLayout lay = lds_layout(g.tile_elems, (int)sizeof(T), (int)sizeof(T), D, M, NC, nwaves);
    T* tile = reinterpret_cast<T*>(smem);
    T* coefs_lds = reinterpret_cast<T*>(smem + lay.tile_bytes);
    Stage<T, NC, D, M> st(smem + lay.tile_bytes + lay.coef_bytes + wave * lay.stage_bytes_per_wave);

    // load the padded tile with periodic wrap (gridvalues_to_local_memory!, src/interpolation/gpu.jl:331-355)
    const T* grid = a.grid[comp_id];
    {
        const int w_row = NC * g.P[0];
        RowWalker rw(g, w_row, wave, nwaves, lane);
        const int o1 = origin[0] - (M - 1), o2 = origin[1] - (M - 1), o3 = origin[2] - (M - 1);
        for (; rw.valid(); rw.next(g)) {
            int64_t rowbase = 0;
            if constexpr (D >= 2) rowbase = (int64_t)wrap_index(o2 + rw.l2, g.Nover[1]);
            if constexpr (D >= 3) rowbase += (int64_t)wrap_index(o3 + rw.l3, g.Nover[2]) * g.Nover[1];
            rowbase *= (int64_t)g.Nover[0] * NC;
            T* dst = tile + rw.l2 * g.row_stride + rw.l3 * g.plane_stride;
            for (int e = rw.lane_in_row; e < w_row; e += rw.lanes_per_row) {
                const int l1 = e / NC, c = e % NC;
                const int g1 = wrap_index(o1 + l1, g.Nover[0]);
                dst[e] = grid[rowbase + (int64_t)g1 * NC + c];
            }
        }
    }
    if (a.evalmode != NUFFT_EVAL_DIRECT)
        for (int i = tid; i < D * (M + 4) * L; i += nthreads) coefs_lds[i] = a.coefs[i];
    __syncthreads();

    const int gq = lane / F::G, q = lane % F::G;
    int lane_off[F::NPASS], j1v[F::NPASS], j2v[F::NPASS];
    bool act[F::NPASS];
#pragma unroll
    for (int ps = 0; ps < F::NPASS; ++ps) {
        const int e = q + ps * F::G;
        act[ps] = e < F::FACE;
        const int e1 = e % F::W1;
        const int j2 = e / F::W1;
        j1v[ps] = e1 / NC;
        j2v[ps] = j2;
        lane_off[ps] = j2 * g.row_stride + e1;
    }

    const PointRec<T, D>* sorted = static_cast<const PointRec<T, D>*>(a.sorted);
    T* vout = a.vout[comp_id];
    const int npts_tile = (int)(pb - pa);
    const int nchunks = (npts_tile + kCH - 1) / kCH;

    for (int chunk = wave; chunk < nchunks; chunk += nwaves) {
        const uint32_t first = pa + (uint32_t)chunk * kCH;
        const int npts = min(kCH, (int)(pb - first));
        wave_lds_fence();
        const int idx = stage_chunk<T, NC, D, M>(a, sorted, first, npts, origin, coefs_lds, st, lane);
        wave_lds_fence();

        for (int p0 = 0; p0 < npts; p0 += F::PPW) {
            const int pt = p0 + gq;
            const bool pvalid = pt < npts;
            const int ptc = pvalid ? pt : 0;
            const T* wv = st.wv + ptc * (D * L);
            int base = st.ss[ptc * D + 0] * NC;
            if constexpr (D >= 2) base += st.ss[ptc * D + 1] * g.row_stride;
            if constexpr (D >= 3) base += st.ss[ptc * D + 2] * g.plane_stride;
            T acc = T(0);
#pragma unroll
            for (int ps = 0; ps < F::NPASS; ++ps) {
                if (act[ps] && pvalid) {
                    const T* src = tile + base + lane_off[ps];
                    T s;
                    if constexpr (D >= 3) {
                        s = T(0);
#pragma unroll
                        for (int j3 = 0; j3 < L; ++j3) s = fma(src[j3 * g.plane_stride], wv[2 * L + j3], s);
                    } else {
                        s = src[0];
                    }
                    T w = wv[j1v[ps]];
                    if constexpr (D >= 2) w *= wv[L + j2v[ps]];
                    acc = fma(s, w, acc);
                }
            }
            acc = group_sum<T, F::G, CPLX>(acc);
            if (pvalid && q < NC) st.vv[pt * NC + q] = acc * a.prefactor;
        }
        wave_lds_fence();
        if (lane < kCH && lane < npts) {
#pragma unroll
            for (int c = 0; c < NC; ++c) vout[(int64_t)idx * NC + c] = st.vv[lane * NC + c];
        }
    }
}

}  // namespace nufft
