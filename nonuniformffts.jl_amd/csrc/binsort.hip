// set_points!: bin-sort of the non-uniform points by fine bins (4^D cells by default).
//
// Replaces set_points_impl!(::GPU, ...) of the reference (src/blocking/gpu.jl:73-142):
//   K2 assign_blocks_kernel!  (:162-180)  -> bin_count_kernel   (histogram + rank)
//   K3 AK.accumulate!         (:112-115)  -> hipcub exclusive scan over nbins + 1 counters
//   K4 sortperm_kernel!       (:182-198)  \
//   K5 permute_kernel!        (:200-212)  -> bin_scatter_kernel (fused: writes one aligned record
//                                            {r_1..r_D, original index} per point in tile order)
// Differences by design: 32-bit counters and indices; the sorted copy stores the coordinates in
// grid units r = (x / 2π) Ñ, computed once, so that binning, spreading and interpolation derive
// cell and tile from the *same* number (the consistency hazard noted at src/blocking/gpu.jl:151-155).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "device_common.h"
#include "kernels.h"

namespace nufft {

template <typename T, int D>
struct BinArgs {
    const T* x[3];
    int64_t np;
    Geom g;
    int point_transform;       // NUFFT_POINT_TRANSFORM_*
};

// Linear index of the fine bin of point p (dimension 1 fastest); the analogue of block_index,
// src/blocking/gpu.jl:145-160, with power-of-two bins so that the division is a shift.
template <typename T, int D>
__device__ __forceinline__ uint32_t tile_of_point(const BinArgs<T, D>& a, int64_t p, T (&r)[D]) {
    uint32_t bin = 0, mul = 1;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const T xf = transform_and_fold(a.x[d][p], a.point_transform);
        r[d] = to_grid_units(xf, a.g.Nover[d]);
        const int i = cell_of(r[d], a.g.Nover[d]);
        bin += mul * (uint32_t)(i >> a.g.blog[d]);
        mul *= (uint32_t)a.g.nb[d];
    }
    return bin;
}

// Histogram + rank.  A workgroup first aggregates the bins of a chunk of points in an LDS hash table (key =
// bin, value = count; the slot's running count is the point's rank inside the chunk), then issues ONE global
// atomic per distinct bin of the chunk, and finally combines the returned base with the local ranks.  For
// well-spread points almost every bin of a chunk is distinct and this costs the same number of global atomics
// as one atomic per point; for clustered points (all 1e7 points in a few bins: 14.5 ms of same-address
// atomics before) it removes the contention.
constexpr int kCountThreads = 256;
constexpr int kCountPPT = 4;                               // points per thread and chunk
constexpr int kCountSlots = 2048;                          // hash slots (load factor <= 0.5)
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;

template <typename T, int D>
__global__ __launch_bounds__(kCountThreads) void bin_count_kernel(BinArgs<T, D> a, uint32_t* __restrict__ counts,
                                                                 uint2* __restrict__ binrank, const uint32_t* skip_a, const uint32_t* skip_b) {
    __shared__ uint32_t keys[kCountSlots], cnt[kCountSlots], base[kCountSlots];
    if (skip_a && *skip_a != 0u && *skip_b != 0u) return;      // this point set is column-layer sorted (CoarseSort)
    constexpr int kChunkPts = kCountThreads * kCountPPT;
    const int tid = threadIdx.x;
    const int64_t nchunks = (a.np + kChunkPts - 1) / kChunkPts;
    for (int64_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        for (int s = tid; s < kCountSlots; s += kCountThreads) { keys[s] = kEmptyKey; cnt[s] = 0; }
        __syncthreads();
        uint32_t bin[kCountPPT], lrank[kCountPPT];
        int slot[kCountPPT];
#pragma unroll
        for (int k = 0; k < kCountPPT; ++k) {
            const int64_t p = chunk * kChunkPts + (int64_t)k * kCountThreads + tid;
            slot[k] = -1;
            if (p < a.np) {
                T r[D];
                bin[k] = tile_of_point<T, D>(a, p, r);
                int h = (int)((bin[k] * 2654435761u) >> 21);              // 11 bits
                for (;;) {
                    const uint32_t old = atomicCAS(&keys[h], kEmptyKey, bin[k]);
                    if (old == kEmptyKey || old == bin[k]) break;
                    h = (h + 1) & (kCountSlots - 1);
                }
                slot[k] = h;
                lrank[k] = atomicAdd(&cnt[h], 1u);
            }
        }
        __syncthreads();
        for (int s = tid; s < kCountSlots; s += kCountThreads)
            if (keys[s] != kEmptyKey) base[s] = atomicAdd(&counts[keys[s]], cnt[s]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kCountPPT; ++k) {
            const int64_t p = chunk * kChunkPts + (int64_t)k * kCountThreads + tid;
            if (slot[k] >= 0) binrank[p] = make_uint2(bin[k], base[slot[k]] + lrank[k]);
        }
        __syncthreads();
    }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void bin_scatter_kernel(BinArgs<T, D> a, const uint32_t* __restrict__ offsets,
                                                         const uint2* __restrict__ binrank,
                                                         PointRec<T, D>* __restrict__ sorted, uint32_t* __restrict__ counts, int ncounts,
                                                         const uint32_t* skip_a, const uint32_t* skip_b) {
    if (skip_a && *skip_a != 0u && *skip_b != 0u) return;      // this point set is column-layer sorted (CoarseSort)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // the histogram has been scanned into `offsets`: clear it for the next set_points (saves that call's zero-fill launch)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncounts; i += stride) counts[i] = 0u;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < a.np; p += stride) {
        const uint2 br = binrank[p];
        PointRec<T, D> rec;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const T xf = transform_and_fold(a.x[d][p], a.point_transform);
            rec.r[d] = to_grid_units(xf, a.g.Nover[d]);
        }
        rec.idx = (int32_t)p;
        sorted[offsets[br.x] + br.y] = rec;
    }
}

// Zero fill of the histogram as a kernel rather than hipMemsetAsync: a captured hipGraph that holds a memset node
// faults on replay once other work has run in between (ROCm 7.2, scripts/graph_probe.py), kernel nodes do not.
__global__ void zero_fill_kernel(uint4* __restrict__ dst, int64_t n16, uint32_t* __restrict__ tail, int ntail) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = first; i < n16; i += stride) dst[i] = make_uint4(0u, 0u, 0u, 0u);
    if (first < ntail) tail[first] = 0u;
}

// dst: 16-byte aligned, bytes: multiple of 4
hipError_t launch_zero_fill(void* dst, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return hipSuccess;
    const int64_t n16 = (int64_t)(bytes / 16);
    const int ntail = (int)((bytes % 16) / 4);
    int64_t blocks = (n16 + 1023) / 1024;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, static_cast<uint4*>(dst), n16,
                       reinterpret_cast<uint32_t*>(static_cast<unsigned char*>(dst) + n16 * 16), ntail);
    return hipGetLastError();
}

template <typename T, int D>
static hipError_t run_binsort(const SortArgs& s, hipStream_t stream) {
    BinArgs<T, D> a;
    for (int d = 0; d < 3; ++d) a.x[d] = d < D ? static_cast<const T*>(s.coords[d]) : nullptr;
    a.np = s.np;
    a.g = s.g;
    a.point_transform = s.point_transform;
    hipError_t e = hipSuccess;
    if (!s.counts_clean) {      // normally the previous call's scatter pass has left the histogram zeroed
        e = launch_zero_fill(s.counts, sizeof(uint32_t) * (size_t)(s.g.nbins + 1), stream);
        if (e != hipSuccess) return e;
    }
    if (s.np > 0) {
        int64_t blocks = (s.np + kCountThreads * kCountPPT - 1) / (kCountThreads * kCountPPT);
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL((bin_count_kernel<T, D>), dim3((unsigned)blocks), dim3(kCountThreads), 0, stream, a, s.counts,
                           static_cast<uint2*>(s.binrank), (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    }
    size_t tmp = s.scan_tmp_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(s.scan_tmp, tmp, s.counts, s.offsets, s.g.nbins + 1, stream);
    if (e != hipSuccess) return e;
    if (s.np > 0) {
        const int threads = 256;
        int64_t blocks = (s.np + threads - 1) / threads;
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL((bin_scatter_kernel<T, D>), dim3((unsigned)blocks), dim3(threads), 0, stream, a, s.offsets,
                           static_cast<const uint2*>(s.binrank), static_cast<PointRec<T, D>*>(s.sorted), s.counts, s.g.nbins + 1,
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr);
    }
    return hipGetLastError();
}

// ---- column-layer sort (CoarseSort, kernels.h) -------------------------------------------------------------------------------
struct CoarseGeom {
    int cbx, cby, ncx, ncy, nkeys;
};
constexpr int kCoarseThreads = 1024;

// key of a point from its coordinates (the passes below load a batch ahead of the one they work on)
template <typename T>
__device__ __forceinline__ uint32_t coarse_key_of(const BinArgs<T, 3>& a, const CoarseGeom& c, const T (&x)[3], T (&r)[3]) {
    int b[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const T xf = transform_and_fold(x[d], a.point_transform);
        r[d] = to_grid_units(xf, a.g.Nover[d]);
        b[d] = cell_of(r[d], a.g.Nover[d]) >> a.g.blog[d];
    }
    return (uint32_t)((b[2] * c.ncy + b[1] / c.cby) * c.ncx + b[0] / c.cbx);
}
// coordinates of points p0 + u THREADS + tid (u < U) of a slice that ends at hi
template <typename T, int U>
__device__ __forceinline__ void coarse_load(const BinArgs<T, 3>& a, int64_t p0, int64_t hi, int tid, T (&x)[U][3]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t p = p0 + u * kCoarseThreads + tid;
        if (p < hi) {
#pragma unroll
            for (int d = 0; d < 3; ++d) x[u][d] = a.x[d][p];
        }
    }
}
// first fine bin of a column layer: where its total sits in the fake fine histogram
__device__ __forceinline__ int64_t coarse_rep_bin(const Geom& g, const CoarseGeom& c, int k) {
    const int cx = k % c.ncx, t = k / c.ncx, cy = t % c.ncy, bz = t / c.ncy;
    return ((int64_t)bz * g.nb[1] + cy * c.cby) * g.nb[0] + cx * c.cbx;
}
// slice of the point set that workgroup w counts and later scatters
__device__ __forceinline__ void coarse_slice(int64_t np, int groups, int w, int64_t& lo, int64_t& hi) {
    int64_t per = (np + groups - 1) / groups;
    per = (per + kCoarseThreads - 1) / kCoarseThreads * kCoarseThreads;
    lo = per * w < np ? per * w : np;
    hi = lo + per < np ? lo + per : np;
}

// pass 1: histogram of the slice in LDS (one 32-bit counter per key), written out as one coalesced row of the table
template <typename T>
__global__ __launch_bounds__(kCoarseThreads) void coarse_count_kernel(BinArgs<T, 3> a, CoarseGeom c, uint32_t* __restrict__ table) {
    extern __shared__ uint32_t hist[];
    const int tid = threadIdx.x, w = blockIdx.x;
    for (int k = tid; k < c.nkeys; k += kCoarseThreads) hist[k] = 0u;
    __syncthreads();
    int64_t lo, hi;
    coarse_slice(a.np, (int)gridDim.x, w, lo, hi);
    // two batches of 4 points per thread in registers: the loads of one are in flight while the other is counted (0.71 -> see DESIGN 4.10)
    constexpr int U = 4;
    constexpr int64_t S = (int64_t)U * kCoarseThreads;
    T xa[U][3], xb[U][3];
    auto consume = [&](int64_t p0, const T (&x)[U][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            T r[3];
            if (p0 + u * kCoarseThreads + tid < hi) atomicAdd(&hist[coarse_key_of<T>(a, c, x[u], r)], 1u);
        }
    };
    int64_t p0 = lo;
    if (p0 < hi) coarse_load<T, U>(a, p0, hi, tid, xa);
    while (p0 < hi) {
        if (p0 + S < hi) coarse_load<T, U>(a, p0 + S, hi, tid, xb);
        consume(p0, xa);
        p0 += S;
        if (p0 >= hi) break;
        if (p0 + S < hi) coarse_load<T, U>(a, p0 + S, hi, tid, xa);
        consume(p0, xb);
        p0 += S;
    }
    __syncthreads();
    uint32_t* row = table + (size_t)w * c.nkeys;
    for (int k = tid; k < c.nkeys; k += kCoarseThreads) row[k] = hist[k];
}

// per key: exclusive prefix over the slices (in place), and the key's total into the fake fine histogram
__global__ __launch_bounds__(64) void coarse_prefix_kernel(Geom g, CoarseGeom c, int groups, uint32_t* __restrict__ table, uint32_t* __restrict__ counts,
                                                            uint32_t* __restrict__ maxout) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    uint32_t run = 0u;
    if (k < c.nkeys) {
    int w = 0;
    for (; w + 8 <= groups; w += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = table[(size_t)(w + u) * c.nkeys + k];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            table[(size_t)(w + u) * c.nkeys + k] = run;
            run += v[u];
        }
    }
    for (; w < groups; ++w) {
        const uint32_t v = table[(size_t)w * c.nkeys + k];
        table[(size_t)w * c.nkeys + k] = run;
        run += v;
    }
    counts[coarse_rep_bin(g, c, k)] = run;
    }
    if (maxout) {                                       // (slab sort: the fullest slab decides whether level 2 can hold it)
        uint32_t m = run;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
        if (threadIdx.x == 0) atomicMax(maxout, m);
    }
}
// slab sort: fm[4] = 1 while the fullest slab stays within `limit` records (kSlabOverfill capacities of a level-2 workgroup: fuller slabs are
// sorted by two passes over global memory, which one workgroup should not do for a large part of the point set); fm[0] (the running
// maximum) cleared for the next set_points
__global__ void slab_flag_kernel(uint32_t* fm, uint32_t limit) {
    fm[4] = fm[0] <= limit ? 1u : 0u;
    fm[0] = 0u;
}

// the fine sort takes over (a ring handed the point set to the tile kernels): the fake histogram is cleared first
__global__ __launch_bounds__(256) void coarse_clear_kernel(Geom g, CoarseGeom c, uint32_t* __restrict__ counts, const uint32_t* fa, const uint32_t* fb) {
    if (*fa != 0u && *fb != 0u) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < c.nkeys) counts[coarse_rep_bin(g, c, k)] = 0u;
}

// pass 2: cursors of the slice in LDS (start of the column layer + what the earlier slices put there), one record store per point
template <typename T>
__global__ __launch_bounds__(kCoarseThreads) void coarse_scatter_kernel(BinArgs<T, 3> a, CoarseGeom c, const uint32_t* __restrict__ table,
                                                                       const uint32_t* __restrict__ offsets, PointRec<T, 3>* __restrict__ sorted,
                                                                       uint32_t* __restrict__ counts, const uint32_t* fa, const uint32_t* fb,
                                                                       const uint32_t* ra, const uint32_t* rb, uint32_t* feedback, uint32_t seq) {
    extern __shared__ uint32_t cursor[];
    if (feedback && blockIdx.x == 0 && threadIdx.x == 0) {
        // what the rings decided for this point set, for the host's choice of sort at the NEXT set_points (host-mapped memory, no synchronisation)
        __atomic_store_n(&feedback[0], *ra, __ATOMIC_RELAXED);
        __atomic_store_n(&feedback[1], *rb, __ATOMIC_RELAXED);
        __threadfence_system();
        __atomic_store_n(&feedback[2], seq, __ATOMIC_RELAXED);
    }
    if (*fa == 0u || *fb == 0u) return;                 // fine sort
    const int tid = threadIdx.x, w = blockIdx.x;
    const uint32_t* row = table + (size_t)w * c.nkeys;
    for (int k = tid; k < c.nkeys; k += kCoarseThreads) {
        const int64_t rep = coarse_rep_bin(a.g, c, k);
        cursor[k] = offsets[rep] + row[k];
    }
    __syncthreads();
    int64_t lo, hi;
    coarse_slice(a.np, (int)gridDim.x, w, lo, hi);
    constexpr int U = 2;
    constexpr int64_t S = (int64_t)U * kCoarseThreads;
    T xa[U][3], xb[U][3];
    auto consume = [&](int64_t p0, const T (&x)[U][3]) __attribute__((always_inline)) {
        PointRec<T, 3> rec[U];
        uint32_t key[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + u * kCoarseThreads + tid;
            key[u] = 0xffffffffu;
            if (p < hi) {
                key[u] = coarse_key_of<T>(a, c, x[u], rec[u].r);
                rec[u].idx = (int32_t)p;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (key[u] != 0xffffffffu) sorted[atomicAdd(&cursor[key[u]], 1u)] = rec[u];
    };
    int64_t p0 = lo;
    if (p0 < hi) coarse_load<T, U>(a, p0, hi, tid, xa);
    while (p0 < hi) {
        if (p0 + S < hi) coarse_load<T, U>(a, p0 + S, hi, tid, xb);
        consume(p0, xa);
        p0 += S;
        if (p0 >= hi) break;
        if (p0 + S < hi) coarse_load<T, U>(a, p0 + S, hi, tid, xa);
        consume(p0, xb);
        p0 += S;
    }
    // the histogram has been scanned: clear it for the next set_points (only the first bins of the column layers hold anything)
    __syncthreads();
    for (int k = w * kCoarseThreads + tid; k < c.nkeys; k += (int)gridDim.x * kCoarseThreads) counts[coarse_rep_bin(a.g, c, k)] = 0u;
}


// level 2 of the slab sort: one workgroup per slab — records into LDS, histogram of the slab's fine bins (the atomic's return value is the
// record's rank inside its bin), scan, the slab's fine offsets, the inverse permutation, and the records out in order (coalesced)
constexpr int kSlabThreads = 1024;
constexpr int kSlabIPT = 8;                // records per thread at most (cap <= 8192)
template <typename T>
__global__ __launch_bounds__(kSlabThreads) void slab_sort_kernel(Geom g, CoarseGeom c, int cap, const PointRec<T, 3>* __restrict__ temp,
                                                                 PointRec<T, 3>* __restrict__ sorted, uint32_t* __restrict__ offsets, const uint32_t* flag) {
    extern __shared__ __attribute__((aligned(16))) unsigned char slab_smem[];
    if (*flag == 0u) return;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(slab_smem);                                        // [kSlabMaxBins]
    uint32_t* wsum = cnt + kSlabMaxBins;                                                           // [16] wave totals of the scan
    uint32_t* aux = wsum + 16;                                                                     // [cap] bin << 16 | rank, then source of output j
    PointRec<T, 3>* recs = reinterpret_cast<PointRec<T, 3>*>(aux + cap);                           // [cap]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = blockIdx.x;
    const int cy = (k / c.ncx) % c.ncy;
    const int nbs = min(c.cby, g.nb[1] - cy * c.cby) * g.nb[0];                                     // fine bins of this slab: a contiguous range
    const int64_t rep = coarse_rep_bin(g, c, k);
    const uint32_t base = offsets[rep], n = offsets[rep + nbs] - base;
    for (int i = tid; i < nbs; i += kSlabThreads) cnt[i] = 0u;
    __syncthreads();
    // block-wide exclusive scan of cnt[0 .. nbs) in place (4 consecutive bins per thread, wave scan, wave totals) + the slab's fine offsets
    auto scan_bins = [&]() __attribute__((always_inline)) {
        uint32_t v[4], tsum = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * tid + j;
            v[j] = i < nbs ? cnt[i] : 0u;
            tsum += v[j];
        }
        uint32_t incl = tsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t wbase = 0u;
        for (int w = 0; w < wave; ++w) wbase += wsum[w];
        uint32_t run = wbase + incl - tsum;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 4 * tid + j;
            if (i < nbs) {
                cnt[i] = run;
                offsets[rep + i] = base + run;
            }
            run += v[j];
        }
        __syncthreads();
    };
    auto bin_of = [&](const PointRec<T, 3>& rec) __attribute__((always_inline)) -> uint32_t {
        const int b0 = cell_of(rec.r[0], g.Nover[0]) >> g.blog[0], b1 = cell_of(rec.r[1], g.Nover[1]) >> g.blog[1];
        return (uint32_t)((b1 - cy * c.cby) * g.nb[0] + b0);
    };
    if (n > (uint32_t)cap) {
        // a slab fuller than the LDS holds (denser regions of a non-uniform set): two passes over its records instead — histogram, scan, then
        // every record to the cursor of its bin (scattered stores, but inside the slab's own range and all from this workgroup: they meet in L2)
        for (uint32_t i = (uint32_t)tid; i < n; i += kSlabThreads) atomicAdd(&cnt[bin_of(temp[base + i])], 1u);
        __syncthreads();
        scan_bins();
        for (uint32_t i = (uint32_t)tid; i < n; i += kSlabThreads) {
            const PointRec<T, 3> rec = temp[base + i];
            sorted[base + atomicAdd(&cnt[bin_of(rec)], 1u)] = rec;
        }
        return;
    }
    uint32_t av[kSlabIPT];
#pragma unroll
    for (int u = 0; u < kSlabIPT; ++u) {
        const uint32_t i = (uint32_t)(u * kSlabThreads + tid);
        av[u] = 0u;
        if (i < n) {
            const PointRec<T, 3> rec = temp[base + i];
            recs[i] = rec;
            const uint32_t fb = bin_of(rec);
            av[u] = fb << 16 | atomicAdd(&cnt[fb], 1u);
        }
    }
    __syncthreads();
    scan_bins();
#pragma unroll
    for (int u = 0; u < kSlabIPT; ++u) {
        const uint32_t i = (uint32_t)(u * kSlabThreads + tid);
        if (i < n) aux[cnt[av[u] >> 16] + (av[u] & 0xffffu)] = i;
    }
    __syncthreads();
    for (uint32_t j = (uint32_t)tid; j < n; j += kSlabThreads) sorted[base + j] = recs[aux[j]];
}

static CoarseGeom coarse_geom(const CoarseSort& cs) { return CoarseGeom{cs.cbx, cs.cby, cs.ncx, cs.ncy, cs.nkeys}; }

template <typename T>
static BinArgs<T, 3> bin_args3(const SortArgs& s) {
    BinArgs<T, 3> a;
    for (int d = 0; d < 3; ++d) a.x[d] = static_cast<const T*>(s.coords[d]);
    a.np = s.np;
    a.g = s.g;
    a.point_transform = s.point_transform;
    return a;
}

int slab_sort_lds_bytes(int dtype, int cap) {
    return (kSlabMaxBins + 16) * 4 + cap * (4 + (int)(dtype == NUFFT_F32 ? sizeof(PointRec<float, 3>) : sizeof(PointRec<double, 3>)));
}
int slab_sort_capacity(int dtype, int lds_bytes) {
    const int rb = 4 + (int)(dtype == NUFFT_F32 ? sizeof(PointRec<float, 3>) : sizeof(PointRec<double, 3>));
    int cap = (lds_bytes - (kSlabMaxBins + 16) * 4) / rb;
    cap &= ~63;
    return cap > kSlabIPT * kSlabThreads ? kSlabIPT * kSlabThreads : cap;
}
hipError_t prepare_binsort_slab(int dtype, int lds_bytes) {
    hipError_t e = prepare_binsort_coarse(dtype, kCoarseMaxKeys);
    if (e != hipSuccess) return e;
    return dtype == NUFFT_F32 ? hipFuncSetAttribute(reinterpret_cast<const void*>(&slab_sort_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)
                              : hipFuncSetAttribute(reinterpret_cast<const void*>(&slab_sort_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}
hipError_t prepare_binsort_coarse(int dtype, int nkeys) {
    // (the attribute belongs to the kernel, not to the plan: always the largest table, so that a plan created later cannot lower it
    // under what an earlier plan launches with)
    if (nkeys > kCoarseMaxKeys) return hipErrorInvalidValue;
    const int bytes = kCoarseMaxKeys * 4;
    hipError_t e;
    if (dtype == NUFFT_F32) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&coarse_count_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&coarse_scatter_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&coarse_count_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&coarse_scatter_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    }
    return e;
}

template <typename T>
static hipError_t coarse_count_t(const SortArgs& s, hipStream_t stream) {
    const CoarseGeom c = coarse_geom(s.cs);
    hipError_t e = hipSuccess;
    if (!s.counts_clean) {
        e = launch_zero_fill(s.counts, sizeof(uint32_t) * (size_t)(s.g.nbins + 1), stream);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((coarse_count_kernel<T>), dim3((unsigned)s.cs.groups), dim3(kCoarseThreads), (size_t)c.nkeys * 4, stream, bin_args3<T>(s), c, s.cs.table);
    hipLaunchKernelGGL(coarse_prefix_kernel, dim3((unsigned)((c.nkeys + 63) / 64)), dim3(64), 0, stream, s.g, c, s.cs.groups, s.cs.table, s.counts,
                       s.cs.mode == 2 ? s.cs.flagmem : (uint32_t*)nullptr);
    if (s.cs.mode == 2) hipLaunchKernelGGL(slab_flag_kernel, dim3(1), dim3(1), 0, stream, s.cs.flagmem, (uint32_t)s.cs.cap * (uint32_t)kSlabOverfill);
    size_t tmp = s.scan_tmp_bytes;
    return hipcub::DeviceScan::ExclusiveSum(s.scan_tmp, tmp, s.counts, s.offsets, s.g.nbins + 1, stream);
}

template <typename T>
static hipError_t coarse_finish_t(const SortArgs& s, hipStream_t stream) {
    const CoarseGeom c = coarse_geom(s.cs);
    const BinArgs<T, 3> a = bin_args3<T>(s);
    const uint32_t *fa = s.cs.flag_a, *fb = s.cs.flag_b;
    // fine sort, only where the flags ask for it: clear the fake histogram, count by fine bins; the scan runs either way (on the
    // unchanged fake histogram it reproduces the offsets it already holds)
    hipLaunchKernelGGL(coarse_clear_kernel, dim3((unsigned)((c.nkeys + 255) / 256)), dim3(256), 0, stream, s.g, c, s.counts, fa, fb);
    if (s.np > 0) {
        int64_t blocks = (s.np + kCountThreads * kCountPPT - 1) / (kCountThreads * kCountPPT);
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL((bin_count_kernel<T, 3>), dim3((unsigned)blocks), dim3(kCountThreads), 0, stream, a, s.counts, static_cast<uint2*>(s.binrank), fa, fb);
    }
    size_t tmp = s.scan_tmp_bytes;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(s.scan_tmp, tmp, s.counts, s.offsets, s.g.nbins + 1, stream);
    if (e != hipSuccess) return e;
    // (slab sort: level 1 leaves its records in the temporary array, level 2 sorts every slab by fine bin into `sorted`)
    hipLaunchKernelGGL((coarse_scatter_kernel<T>), dim3((unsigned)s.cs.groups), dim3(kCoarseThreads), (size_t)c.nkeys * 4, stream, a, c, s.cs.table, s.offsets,
                       static_cast<PointRec<T, 3>*>(s.cs.mode == 2 ? s.cs.temp : s.sorted), s.counts, fa, fb, s.cs.fb_a, s.cs.fb_b,
                       (s.cs.fb_a && s.cs.fb_b) ? s.cs.feedback : (uint32_t*)nullptr, s.cs.seq);
    if (s.cs.mode == 2)
        hipLaunchKernelGGL((slab_sort_kernel<T>), dim3((unsigned)c.nkeys), dim3(kSlabThreads), (size_t)s.cs.lds2, stream, s.g, c, s.cs.cap,
                           static_cast<const PointRec<T, 3>*>(s.cs.temp), static_cast<PointRec<T, 3>*>(s.sorted), s.offsets, fa);
    if (s.np > 0) {
        int64_t blocks = (s.np + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        hipLaunchKernelGGL((bin_scatter_kernel<T, 3>), dim3((unsigned)blocks), dim3(256), 0, stream, a, s.offsets, static_cast<const uint2*>(s.binrank),
                           static_cast<PointRec<T, 3>*>(s.sorted), s.counts, s.g.nbins + 1, fa, fb);
    }
    return hipGetLastError();
}

// the rings' decisions for the host's next choice of sort, where no scatter pass of the column-layer sort carries them (CoarseSort::feedback)
__global__ void sort_feedback_kernel(const uint32_t* ra, const uint32_t* rb, uint32_t* feedback, uint32_t seq) {
    __atomic_store_n(&feedback[0], *ra, __ATOMIC_RELAXED);
    __atomic_store_n(&feedback[1], *rb, __ATOMIC_RELAXED);
    __threadfence_system();
    __atomic_store_n(&feedback[2], seq, __ATOMIC_RELAXED);
}
hipError_t launch_sort_feedback(const uint32_t* ra, const uint32_t* rb, uint32_t* feedback, uint32_t seq, hipStream_t stream) {
    hipLaunchKernelGGL(sort_feedback_kernel, dim3(1), dim3(1), 0, stream, ra, rb, feedback, seq);
    return hipGetLastError();
}

hipError_t launch_binsort_coarse_count(const SortArgs& s, hipStream_t stream) {
    return s.dtype == NUFFT_F32 ? coarse_count_t<float>(s, stream) : coarse_count_t<double>(s, stream);
}
hipError_t launch_binsort_coarse_finish(const SortArgs& s, hipStream_t stream) {
    return s.dtype == NUFFT_F32 ? coarse_finish_t<float>(s, stream) : coarse_finish_t<double>(s, stream);
}

size_t binsort_scan_tmp_bytes(int nbins) {
    size_t bytes = 0;
    uint32_t* p = nullptr;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, p, p, nbins + 1, (hipStream_t)0);
    return bytes < 16 ? 16 : bytes;
}

size_t point_record_bytes(int dtype, int D) {
    if (dtype == NUFFT_F32) return D == 1 ? sizeof(PointRec<float, 1>) : D == 2 ? sizeof(PointRec<float, 2>) : sizeof(PointRec<float, 3>);
    return D == 1 ? sizeof(PointRec<double, 1>) : D == 2 ? sizeof(PointRec<double, 2>) : sizeof(PointRec<double, 3>);
}

hipError_t launch_binsort(const SortArgs& s, hipStream_t stream) {
    if (s.dtype == NUFFT_F32) {
        switch (s.D) {
            case 1: return run_binsort<float, 1>(s, stream);
            case 2: return run_binsort<float, 2>(s, stream);
            default: return run_binsort<float, 3>(s, stream);
        }
    }
    switch (s.D) {
        case 1: return run_binsort<double, 1>(s, stream);
        case 2: return run_binsort<double, 2>(s, stream);
        default: return run_binsort<double, 3>(s, stream);
    }
}

// Extracts the permutation (sorted position -> original index) from the sorted records.
template <int REC_BYTES>
__global__ void extract_perm_kernel(const unsigned char* __restrict__ recs, int idx_off, int64_t np, int32_t* __restrict__ perm) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < np) perm[p] = *reinterpret_cast<const int32_t*>(recs + p * REC_BYTES + idx_off);
}

hipError_t launch_extract_perm(int dtype, int D, const void* sorted, int64_t np, int32_t* perm_dev, hipStream_t stream) {
    if (np <= 0) return hipSuccess;
    const size_t rb = point_record_bytes(dtype, D);
    const int idx_off = D * (dtype == NUFFT_F32 ? 4 : 8);
    const unsigned blocks = (unsigned)((np + 255) / 256);
    const unsigned char* r = static_cast<const unsigned char*>(sorted);
    switch (rb) {
        case 8: hipLaunchKernelGGL(extract_perm_kernel<8>, dim3(blocks), dim3(256), 0, stream, r, idx_off, np, perm_dev); break;
        case 16: hipLaunchKernelGGL(extract_perm_kernel<16>, dim3(blocks), dim3(256), 0, stream, r, idx_off, np, perm_dev); break;
        default: hipLaunchKernelGGL(extract_perm_kernel<32>, dim3(blocks), dim3(256), 0, stream, r, idx_off, np, perm_dev); break;
    }
    return hipGetLastError();
}

}  // namespace nufft
