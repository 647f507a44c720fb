// Load balance for non-uniform point distributions.
//
// One workgroup per LDS tile is ideal for uniform points; with clustered points (the reference's own
// benchmark draws folded N(0, 1) coordinates, benchmark/CPU+AMDGPU/run_benchmarks.jl:57-66) a few tiles
// hold most of the points and their workgroups run alone at the end.  After the bin sort, set_points
// therefore measures the work of every tile from the bin offsets and gives heavy tiles several
// *slices*: workgroups that share the tile's points.  Slices of a spreading tile add their partial tiles
// to the grid with global float atomics (the tile interior is zeroed first); slices of an interpolation
// tile simply load the same tile.  The launch grid is fixed (tiles + a budget of extra slices), so no
// device-to-host synchronisation is needed; surplus workgroups exit at once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <hipcub/hipcub.hpp>

#include "device_common.h"
#include "kernels.h"

namespace nufft {

// kWorkLanes lanes per tile of either tiling (groups [0, nsp): spreading tiles, [nsp, nsp + nip): interpolation
// tiles): number of point visits of a spreading tile (points of all bins within M cells of the interior)
// or number of points of an interpolation tile.  (A tile has 6 - 40 runs of bins: a whole wave per tile left most lanes
// idle and cost 0.32 ms for the 7e5 tiles of a 1024^3 grid.)
constexpr int kWorkLanes = 16;

__global__ __launch_bounds__(256) void tile_work_kernel(Geom g, int D, int M, const uint32_t* __restrict__ offsets,
                                                       uint32_t* __restrict__ work, const uint32_t* skip_a, const uint32_t* skip_b) {
    if (skip_a && *skip_a != 0u && *skip_b != 0u) return;      // column-layer sorted point set: both rings serve it
    const int nsp = g.sp.ntiles, nip = g.ip.ntiles;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWorkLanes;      // (the tile of this group of lanes)
    const int lane = threadIdx.x & (kWorkLanes - 1);
    if (wave >= nsp + nip) return;
    const bool interp = wave >= nsp;
    const TileShape& ts = interp ? g.ip : g.sp;
    int rem = interp ? wave - nsp : wave, t[3];
    t[0] = rem % ts.nt[0]; rem /= ts.nt[0];
    t[1] = rem % ts.nt[1]; rem /= ts.nt[1];
    t[2] = rem;
    BinSegs seg[3];
    for (int d = 0; d < 3; ++d) {
        seg[d].n = 1; seg[d].lo[0] = 0; seg[d].len[0] = 1; seg[d].lo[1] = 0; seg[d].len[1] = 0;
        if (d >= D) continue;
        const int org = t[d] * ts.n[d];
        const int neff = min(ts.n[d], g.Nover[d] - org);
        if (interp) {
            seg[d].lo[0] = org >> g.blog[d];
            seg[d].len[0] = ((org + neff - 1) >> g.blog[d]) - seg[d].lo[0] + 1;
        } else {
            seg[d] = bin_segments(org - M, org + neff + M - 1, g.Nover[d], g.blog[d], g.nb[d]);
        }
    }
    const int R2 = seg[1].total(), R3 = seg[2].total();
    const int nruns = R2 * R3 * seg[0].n;
    uint32_t sum = 0;
    for (int item = lane; item < nruns; item += kWorkLanes) {
        const int sg = item % seg[0].n;
        const int r2 = (item / seg[0].n) % R2;
        const int r3 = item / (seg[0].n * R2);
        const int bin0 = (seg[2].bin(r3) * g.nb[1] + seg[1].bin(r2)) * g.nb[0] + (sg ? seg[0].lo[1] : seg[0].lo[0]);
        sum += offsets[bin0 + (sg ? seg[0].len[1] : seg[0].len[0])] - offsets[bin0];
    }
    for (int o = kWorkLanes / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, kWorkLanes);
    if (lane == 0) work[wave] = sum;
}

// Slices of tile t: 1 + its share of the extra budget, proportional to its work.  Two kernels: kSumBlocks workgroups sum
// the work of each tiling into partial sums (a global atomic per tile on one address costs 0.4 ms for 3e4 tiles; one
// workgroup doing everything took 0.62 ms for the 7e5 tiles of a 1024^3 grid), then every workgroup of the second kernel
// adds the partial sums up and writes nslices for its tiles; entry nsp + nip is the terminator of the exclusive scan.
constexpr int kSumBlocks = 128;

__global__ __launch_bounds__(256) void tile_work_sums_kernel(const uint32_t* __restrict__ work, int nsp, int nip,
                                                            unsigned long long* __restrict__ part, const uint32_t* skip_a, const uint32_t* skip_b) {
    __shared__ unsigned long long wsum[2][256 / kWave];
    if (skip_a && *skip_a != 0u && *skip_b != 0u) return;
    const int tid = threadIdx.x, n = nsp + nip;
    unsigned long long acc[2] = {0ull, 0ull};
    for (int t = blockIdx.x * blockDim.x + tid; t < n; t += gridDim.x * blockDim.x) acc[t >= nsp ? 1 : 0] += work[t];
    for (int k = 0; k < 2; ++k) {
        unsigned long long v = acc[k];
        for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
        if ((tid & (kWave - 1)) == 0) wsum[k][tid / kWave] = v;
    }
    __syncthreads();
    if (tid < 2) {
        unsigned long long v = 0;
        for (int w = 0; w < 256 / kWave; ++w) v += wsum[tid][w];
        part[tid * kSumBlocks + blockIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void tile_slices_kernel(const uint32_t* __restrict__ work, int nsp, int nip,
                                                         uint32_t extra_sp, uint32_t extra_ip, uint32_t smax,
                                                         const unsigned long long* __restrict__ part,
                                                         uint32_t* __restrict__ nslices, const uint32_t* skip_a, const uint32_t* skip_b) {
    __shared__ unsigned long long tot[2];
    if (skip_a && *skip_a != 0u && *skip_b != 0u) {      // no tile kernel will run: one slice per tile (nothing for zero_split_tiles to clear)
        const int t = blockIdx.x * blockDim.x + threadIdx.x;
        if (t < nsp + nip) nslices[t] = 1u;
        return;
    }
    const int tid = threadIdx.x, n = nsp + nip;
    if (tid < 2 * kWave) {                              // waves 0, 1: total work of the spreading / interpolation tiling
        const int k = tid / kWave, lane = tid & (kWave - 1);
        unsigned long long v = 0;
        for (int b = lane; b < kSumBlocks; b += kWave) v += part[k * kSumBlocks + b];
        for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_down(v, o, kWave);
        if (lane == 0) tot[k] = v;
    }
    __syncthreads();
    const int t = blockIdx.x * blockDim.x + tid;
    if (t > n) return;
    if (t == n) { nslices[t] = 0; return; }
    const bool interp = t >= nsp;
    const unsigned long long W = tot[interp ? 1 : 0];
    unsigned long long s = W ? (unsigned long long)work[t] * (interp ? extra_ip : extra_sp) / W : 0ull;
    if (s > smax - 1) s = smax - 1;
    nslices[t] = 1u + (uint32_t)s;
}

// descriptor of slot q: x = tile, y = slice << 16 | slices of the tile.  Spreading slots start at desc[0],
// interpolation slots at desc[ip_base]; slots_in_use[0 / 1] receive the two slot counts.
__global__ __launch_bounds__(256) void fill_desc_kernel(const uint32_t* __restrict__ nslices,
                                                       const uint32_t* __restrict__ desc_off, int nsp, int nip,
                                                       uint32_t ip_base, uint2* __restrict__ desc,
                                                       uint32_t* __restrict__ slots_in_use, const uint32_t* skip_a, const uint32_t* skip_b,
                                                       const uint32_t* sp_served) {
    constexpr int kLanes = 8;                            // lanes per tile (most tiles have one slice)
    const int t = (blockIdx.x * blockDim.x + threadIdx.x) / kLanes, lane = threadIdx.x & (kLanes - 1);
    if (skip_a && *skip_a != 0u && *skip_b != 0u) {      // both rings serve this point set: no slots for the tile kernels
        if (blockIdx.x == 0 && threadIdx.x < 2) slots_in_use[threadIdx.x] = 0u;
        return;
    }
    if (t >= nsp + nip) return;
    const uint32_t split = desc_off[nsp];
    const bool interp = t >= nsp;
    const uint32_t S = nslices[t];
    const uint32_t off = interp ? ip_base + (desc_off[t] - split) : desc_off[t];
    const uint32_t tile = interp ? (uint32_t)(t - nsp) : (uint32_t)t;
    for (uint32_t s = lane; s < S; s += kLanes) desc[off + s] = make_uint2(tile, (s << 16) | S);
    if (t == 0 && lane == 0) {
        slots_in_use[0] = (sp_served && *sp_served != 0u) ? 0u : split;      // (a ring spreads this point set: the tile kernel finds no work)
        slots_in_use[1] = desc_off[nsp + nip] - split;
    }
}

// 32-bit words of the work buffer: one counter per tile of both tilings, then (8-byte aligned) the 2 x kSumBlocks
// 64-bit partial sums of tile_work_sums_kernel
size_t balance_work_words(int ntiles_both) {
    return (((size_t)ntiles_both + 1 + 1) & ~(size_t)1) + 4 * (size_t)kSumBlocks;
}

size_t balance_scan_tmp_bytes(int ntiles_both) {
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, ntiles_both + 1);
    return bytes;
}

hipError_t launch_balance(const BalanceArgs& b, hipStream_t stream) {
    const int nsp = b.g.sp.ntiles, nip = b.g.ip.ntiles, n = nsp + nip;
    const int waves_per_block = 256 / kWorkLanes;
    hipLaunchKernelGGL(tile_work_kernel, dim3((unsigned)((n + waves_per_block - 1) / waves_per_block)), dim3(256), 0, stream,
                       b.g, b.D, b.M, b.offsets, b.work, b.skip_a, b.skip_b);
    // (the partial sums live behind the n + 1 work counters: balance_work_words())
    unsigned long long* part = reinterpret_cast<unsigned long long*>(b.work + balance_work_words(n) - 4 * kSumBlocks);
    hipLaunchKernelGGL(tile_work_sums_kernel, dim3(kSumBlocks), dim3(256), 0, stream, b.work, nsp, nip, part, b.skip_a, b.skip_b);
    hipLaunchKernelGGL(tile_slices_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, stream, b.work, nsp, nip,
                       b.enabled ? b.extra_sp : 0u, b.enabled ? b.extra_ip : 0u, b.smax, part, b.nslices, b.skip_a, b.skip_b);
    size_t tmp = b.scan_tmp_bytes;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(b.scan_tmp, tmp, b.nslices, b.desc_off, n + 1, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fill_desc_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, stream, b.nslices, b.desc_off, nsp, nip,
                       (uint32_t)nsp + b.extra_sp, b.desc, b.slots_in_use, b.skip_a, b.skip_b, b.sp_served);
    return hipGetLastError();
}

// ---- tasks of the MFMA-patch engine, per point set ------------------------------------------------------------------
// A patch task is a wave that owns a patch column (4 x pby cube columns) for a segment of cube layers along dimension 3;
// it is not shared between workgroups, so the heaviest task bounds the kernel.  Segments of equal LENGTH serialise on
// non-uniform point sets (folded N(0, 1) coordinates, the reference's own benchmark distribution, put 15x the mean into
// the central tasks), so set_points cuts every column into segments of about equal point COUNT: column c gets
// S_c = 1 + (T - columns) * points(c) / Np of the plan's T tasks, and its segment boundaries are the quantiles of its
// points along dimension 3 (multiples of `zq` layers: the Float32-accumulating kernel retires whole octets).  The table
// {column, first layer, end layer} per task stays on the device; tasks that received nothing are empty entries.
// The last workgroup then decides which engine serves this point set: a segment cannot be shorter than zq layers, so a
// point set that concentrates in a few layers of a few columns still leaves tasks too heavy for the patches and goes to
// the LDS tiles, whose heavy tiles are shared by several workgroups.  choice[2] = 1: patches (and no slots for the LDS-tile
// kernel), 0: tiles.  No host read-back.
constexpr int kPatchMaxLayers = 2048;                   // bin layers of a column the splitter holds in LDS (per wave)

__global__ __launch_bounds__(256) void patch_column_sums_kernel(Geom g, int npx, int npy, int pbx, int pby, int segl, const uint32_t* __restrict__ offsets,
                                                               uint32_t* __restrict__ colsum, uint32_t* __restrict__ choice) {
    __shared__ uint32_t seg_all[256 / kWave][kPatchMaxLayers];          // points per equal-length segment of the wave's column
    const int w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int c = blockIdx.x * (256 / kWave) + w;
    if (c >= npx * npy) return;
    const int bx0 = (c % npx) * pbx, by0 = (c / npx) * pby;
    const int ncx = min(pbx, g.nb[0] - bx0), ncy = min(pby, g.nb[1] - by0);
    const int nz = g.nb[2], nseg = (nz + segl - 1) / segl;
    uint32_t* seg = seg_all[w];
    for (int s = lane; s < nseg; s += kWave) seg[s] = 0u;
    wave_lds_fence();
    uint32_t total = 0;
    for (int z = lane; z < nz; z += kWave) {              // one bin layer per lane and step: all its loads in flight together
        uint32_t n = 0;
        for (int y = 0; y < ncy; ++y) {
            const int64_t bin0 = ((int64_t)z * g.nb[1] + by0 + y) * g.nb[0] + bx0;
            n += offsets[bin0 + ncx] - offsets[bin0];
        }
        total += n;
        if (n) atomicAdd(&seg[z / segl], n);
    }
    wave_lds_fence();
    uint32_t heaviest = 0;                                // of the segments of equal length: the partition uniform point sets keep
    for (int s = lane; s < nseg; s += kWave) heaviest = max(heaviest, seg[s]);
    for (int o = kWave / 2; o > 0; o >>= 1) {
        total += __shfl_down(total, o, kWave);
        heaviest = max(heaviest, (uint32_t)__shfl_down(heaviest, o, kWave));
    }
    if (lane == 0) {
        colsum[c] = total;
        atomicMax(&choice[6], heaviest);                   // (its own word: choice[3] is the equal-length-mode flag; zeroed by its reader)
    }
}

// one workgroup: equal-length segments if their heaviest task is within 10 % (+ 5 sigma of a Poisson count) of the mean
// task — uniform point sets keep exactly the partition they always had, whose tasks differ by a per cent, where quantile
// boundaries on whole layers would make them differ by a layer's worth (9 % at 11 layers per segment) — else
// S_c = round(T points(c) / Np) segments per column; their exclusive scan (first[c]; first[ncols] = tasks in use; the table
// holds T + columns entries, enough for any rounding), empty entries behind.  choice[3] = 1: equal-length mode;
// choice[6]: heaviest equal-length task of this point set (patch_column_sums_kernel), read and zeroed here.
__global__ __launch_bounds__(1024) void patch_task_counts_kernel(int ncols, int ntasks, int ntab, int nseg, int min_seg, int max_seg, unsigned long long np,
                                                                unsigned long long limit, unsigned long long slots_eff, int uniform_always,
                                                                const uint32_t* __restrict__ colsum, uint32_t* __restrict__ first,
                                                                uint2* __restrict__ tasktab, uint32_t* __restrict__ choice,
                                                                uint32_t* __restrict__ slots_in_use) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t carry;
    __shared__ int uniform_mode;
    const int tid = threadIdx.x;
    if (tid == 0) {
        carry = 0;
        const double mean = (double)np / (double)ntasks;
        const uint32_t heaviest = choice[6];
        choice[6] = 0u;                                    // the next set_points starts its maximum from zero
        choice[kHaloStateWord] = 0u;                       // spreading ring: a side buffer of the previous point set is void
        uniform_mode = (double)heaviest <= 1.1 * mean + 5.0 * sqrt(mean) + 8.0;
        choice[3] = uniform_mode ? 1u : 0u;
        if (uniform_mode) {
            // equal-length tasks: the engine is decided here (patch_split_kernel only writes the table then).  The patches
            // always keep such a point set; the ring's estimate needs no halo (clo = chi = 0): heaviest task and np are exact.
            const bool keep = uniform_always || np == 0ull || ((unsigned long long)heaviest <= limit && np <= limit * slots_eff);
            choice[2] = keep ? 1u : 0u;
            if (keep && slots_in_use) slots_in_use[0] = 0u;
        }
    }
    __syncthreads();
    const bool uni = uniform_mode != 0;
    for (int c0 = 0; c0 < ncols; c0 += 1024) {
        const int c = c0 + tid;
        uint32_t S = 0;
        if (c < ncols) {
            if (uni) S = (uint32_t)nseg;
            else {
                S = np ? (uint32_t)(((unsigned long long)ntasks * colsum[c] + np / 2) / np) : 1u;
                S = S < (uint32_t)min_seg ? (uint32_t)min_seg : (S > (uint32_t)max_seg ? (uint32_t)max_seg : S);
            }
        }
        part[tid] = S;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {             // inclusive scan of the chunk
            const uint32_t v = tid >= o ? part[tid - o] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (c < ncols) first[c] = carry + part[tid] - S;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) first[ncols] = carry;
    for (int t = (int)carry + tid; t < ntab; t += 1024) tasktab[t] = make_uint2(0u, 0u);
}

__global__ __launch_bounds__(256) void patch_split_kernel(Geom g, int npx, int npy, int pbx, int pby, int clo, int chi, int zq, int segl, int maxlen,
                                                         const uint32_t* __restrict__ offsets, const uint32_t* __restrict__ first,
                                                         unsigned long long limit, unsigned long long limit_cut, unsigned long long slots_eff, int uniform_always,
                                                         double rho_eff_max, unsigned long long np,
                                                         uint2* __restrict__ tasktab, uint32_t* __restrict__ choice,
                                                         uint32_t* __restrict__ slots_in_use) {
    __shared__ uint32_t cum_all[256 / kWave][kPatchMaxLayers + 1];
    __shared__ uint16_t bnd_all[256 / kWave][kPatchMaxLayers + 2];      // segment boundaries of the wave's column
    const int w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int c = blockIdx.x * (256 / kWave) + w;
    const int nz = g.nb[2];
    if (choice[3] != 0u) {
        // equal-length mode: the table in launch order (segment, column); patch_task_counts_kernel has decided the engine
        if (c < npx * npy) {
            const uint32_t S = first[c + 1] - first[c];
            for (uint32_t k = lane; k < S; k += kWave) {
                const int z0 = min(nz, (int)k * segl), z1 = min(nz, ((int)k + 1) * segl);
                tasktab[k * (uint32_t)(npx * npy) + (uint32_t)c] = make_uint2((uint32_t)c, z1 > z0 ? ((uint32_t)z1 << 16) | (uint32_t)z0 : 0u);
            }
        }
        return;
    }
    uint32_t* cum = cum_all[w];
    unsigned long long wsum = 0, wsq = 0;                 // wsq: sum of n^2 over the column's layers (the density the points see)
    uint32_t wmax = 0;
    if (c < npx * npy) {
        const int bx0 = (c % npx) * pbx, by0 = (c / npx) * pby;
        const int ncx = min(pbx, g.nb[0] - bx0), ncy = min(pby, g.nb[1] - by0);
        // points per layer, then their running sum: lane l owns the layers [l * per, (l + 1) * per)
        const int per = (nz + kWave - 1) / kWave;
        uint32_t run = 0;
        for (int k = 0; k < per; ++k) {
            const int z = lane * per + k;
            if (z < nz) {
                uint32_t n = 0;
                for (int y = 0; y < ncy; ++y) {
                    const int64_t bin0 = ((int64_t)z * g.nb[1] + by0 + y) * g.nb[0] + bx0;
                    n += offsets[bin0 + ncx] - offsets[bin0];
                }
                run += n;
                wsq += (unsigned long long)n * n;
                cum[z + 1] = run;                        // (local to the lane's chunk for now)
            }
        }
        uint32_t incl = run;                             // inclusive scan of the chunk sums over the lanes
        for (int o = 1; o < kWave; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += v;
        }
        const uint32_t before = incl - run;
        for (int k = 0; k < per; ++k) {
            const int z = lane * per + k;
            if (z < nz) cum[z + 1] += before;
        }
        if (lane == 0) cum[0] = 0;
        wave_lds_fence();
        const uint32_t total = cum[nz];
        const uint32_t t0 = first[c], S = first[c + 1] - t0;
        constexpr bool uni = false;                       // (equal-length mode returned above)
        auto boundary = [&](uint32_t k) -> int {          // first layer of segment k (k = S: the end)
            if (k == 0) return 0;
            if (k >= S) return nz;
            if (uni) return min(nz, (int)k * segl);
            int z;
            if (total == 0) {
                z = (int)((unsigned long long)k * nz / S);
            } else {
                const uint32_t target = (uint32_t)(((unsigned long long)k * total + S - 1) / S);
                int lo = 0, hi = nz;                      // smallest z with cum[z] >= target
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (cum[mid] >= target) hi = mid; else lo = mid + 1;
                }
                z = lo;
            }
            z = (z + zq - 1) / zq * zq;
            return z < nz ? z : nz;
        };
        uint16_t* bnd = bnd_all[w];
        for (uint32_t k = lane; k <= S; k += kWave) bnd[k] = (uint16_t)boundary(k);
        wave_lds_fence();
        if (maxlen > 0 && !uni && lane == 0) {
            // no segment longer than the consumer's tables hold (S >= nz / maxlen by construction): pull boundaries forward,
            // then push the ones in front of a long last stretch back
            for (uint32_t k = 1; k < S; ++k) bnd[k] = (uint16_t)min((int)bnd[k], (int)bnd[k - 1] + maxlen);
            for (int k = (int)S - 1; k > 0; --k) bnd[k] = (uint16_t)max((int)bnd[k], (int)bnd[k + 1] - maxlen);
        }
        wave_lds_fence();
        for (uint32_t k = lane; k < S; k += kWave) {
            const int z0 = bnd[k], z1 = bnd[k + 1];
            tasktab[uni ? k * (uint32_t)(npx * npy) + (uint32_t)c : t0 + k] = make_uint2((uint32_t)c, z1 > z0 ? ((uint32_t)z1 << 16) | (uint32_t)z0 : 0u);
            if (z1 > z0) {
                // the points a task visits along dimension 3: its own layers and the stencil's reach beyond them
                const uint32_t work = cum[min(nz, z1 - clo)] - cum[max(0, z0 - chi)];
                wsum += work;
                wmax = max(wmax, work);
            }
        }
        for (int o = kWave / 2; o > 0; o >>= 1) {
            wsum += __shfl_down(wsum, o, kWave);
            wsq += __shfl_down(wsq, o, kWave);
            wmax = max(wmax, (uint32_t)__shfl_down(wmax, o, kWave));
        }
        if (lane == 0) {
            atomicMax(&choice[0], wmax);
            atomicAdd(reinterpret_cast<unsigned long long*>(choice + 4), wsum);
            atomicAdd(reinterpret_cast<unsigned long long*>(choice + 8), wsq);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t ticket = atomicAdd(&choice[1], 1u);
        if (ticket == gridDim.x - 1) {
            __threadfence();
            const unsigned long long mx = atomicExch(&choice[0], 0u);
            const unsigned long long sum = atomicExch(reinterpret_cast<unsigned long long*>(choice + 4), 0ull);
            const unsigned long long sumsq = atomicExch(reinterpret_cast<unsigned long long*>(choice + 8), 0ull);
            choice[1] = 0u;
            // the kernel takes about max(heaviest task, all tasks / wave slots) point visits per wave: patches while that
            // stays within `limit` (launch_patch_tasks)
            const unsigned long long lim = choice[3] != 0u ? limit : limit_cut;      // (the estimate is exact for equal-length tasks)
            bool patches = (mx <= lim && sum <= lim * slots_eff) || sum == 0ull || (uniform_always && choice[3] != 0u);
            // engines whose advantage falls with the density (the interpolation ring): veto by the density the points themselves see
            if (rho_eff_max > 0.0 && np > 0ull && !uniform_always) {
                const double rho_eff = (double)sumsq / (double)np / ((double)(pbx * pby) * 64.0);     // bins of 4^3 cells
                if (rho_eff > rho_eff_max) patches = false;
            }
            choice[2] = patches ? 1u : 0u;
            if (patches && slots_in_use) slots_in_use[0] = 0u;
        }
    }
}

// Launch order of the tasks: segment index first, patch column (x fastest) second — the four waves of a workgroup then
// own neighbouring columns over about the same layers and share the records, values and bin offsets they read (with
// uniform points exactly the order of equal-length segments; column-major order cost 8-16 % there).  One workgroup
// sorts the table in LDS (bitonic, 64-bit keys {segment, column, layers}; empty tasks go last).
constexpr int kPatchSortMax = 16384;

__global__ __launch_bounds__(1024) void patch_task_sort_kernel(int ntasks, int npad, const uint32_t* __restrict__ first, int ncols,
                                                              const uint32_t* __restrict__ choice, uint2* __restrict__ tasktab) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sort[];
    unsigned long long* key = reinterpret_cast<unsigned long long*>(smem_sort);
    const int tid = threadIdx.x;
    if (choice[3] != 0u) return;                          // equal-length mode: patch_split_kernel wrote the table in this order
    for (int t = tid; t < npad; t += 1024) {
        unsigned long long k = ~0ull;
        if (t < ntasks) {
            const uint2 e = tasktab[t];
            if (e.y != 0u) k = ((unsigned long long)(t - first[e.x]) << 48) | ((unsigned long long)e.x << 32) | e.y;
        }
        key[t] = k;
    }
    __syncthreads();
    for (int size = 2; size <= npad; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < npad / 2; i += 1024) {
                const int lo = 2 * i - (i & (stride - 1));            // index of the pair's lower element
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = key[lo], b = key[hi];
                if ((a > b) == up) { key[lo] = b; key[hi] = a; }
            }
            __syncthreads();
        }
    }
    for (int t = tid; t < ntasks; t += 1024) {
        const unsigned long long k = key[t];
        tasktab[t] = k == ~0ull ? make_uint2(0u, 0u) : make_uint2((uint32_t)(k >> 32) & 0xffffu, (uint32_t)k);
    }
}

bool patch_tasks_supported(const Geom& g) { return g.nb[2] <= kPatchMaxLayers; }

// plan creation: the sort kernel's 128 KiB of dynamic LDS (the attribute is per device)
hipError_t prepare_column_tasks() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(patch_task_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPatchSortMax * 8);
}
int patch_task_table_entries(const PatchPlan& pp) { return pp.ntasks + 2 * pp.npx * pp.npy; }      // column_task_table_entries of its columns

// limit / slots_eff: the engine keeps the point set while max(heaviest task, all tasks / slots_eff) <= limit (limit_cut for
// tasks of equal point count)
static hipError_t launch_column_tasks(const Geom& g, const ColumnTasks& ct, const uint32_t* offsets, int64_t np, unsigned long long limit,
                                      unsigned long long limit_cut, unsigned long long slots_eff, bool uniform_always, uint32_t* choice, uint32_t* slots_in_use,
                                      uint32_t* colsum, uint32_t* first, uint2* tasktab, hipStream_t stream, double rho_eff_max = 0.0) {
    const int ncols = ct.ncolx * ct.ncoly, wpb = 256 / kWave;
    const int ntab = column_task_table_entries(ct, g.nb[2]);
    hipLaunchKernelGGL(patch_column_sums_kernel, dim3((unsigned)((ncols + wpb - 1) / wpb)), dim3(256), 0, stream, g, ct.ncolx, ct.ncoly, ct.bxw, ct.byw,
                       ct.segl, offsets, colsum, choice);
    const int min_seg = ct.maxlen > 0 ? (g.nb[2] + ct.maxlen - 1) / ct.maxlen : 1;
    hipLaunchKernelGGL(patch_task_counts_kernel, dim3(1), dim3(1024), 0, stream, ncols, ct.ntasks, ntab, ct.nseg, min_seg, g.nb[2] / ct.zq,
                       (unsigned long long)np, limit, slots_eff, uniform_always ? 1 : 0, colsum, first, tasktab, choice, slots_in_use);
    hipLaunchKernelGGL(patch_split_kernel, dim3((unsigned)((ncols + wpb - 1) / wpb)), dim3(256), 0, stream, g, ct.ncolx, ct.ncoly, ct.bxw, ct.byw,
                       ct.clo, ct.chi, ct.zq, ct.segl, ct.maxlen, offsets, first, limit, limit_cut, slots_eff, uniform_always ? 1 : 0, rho_eff_max,
                       (unsigned long long)np, tasktab, choice, slots_in_use);
    if (ntab <= kPatchSortMax && ncols < 65536) {
        int npad = 2;
        while (npad < ntab) npad <<= 1;
        hipLaunchKernelGGL(patch_task_sort_kernel, dim3(1), dim3(1024), (size_t)npad * 8, stream, ntab, npad, first, ncols, choice, tasktab);
    }
    return hipGetLastError();
}

hipError_t launch_patch_tasks(const Geom& g, const PatchPlan& pp, int clo, int chi, const uint32_t* offsets, int64_t np,
                              int wave_slots, double advantage, uint32_t* choice, uint32_t* slots_in_use, uint32_t* colsum, uint32_t* first,
                              uint2* tasktab, hipStream_t stream) {
    ColumnTasks ct{pp.npx, pp.npy, 4, pp.pby, pp.nseg, pp.segl, pp.ntasks, pp.f32acc ? 2 : 1, clo, chi, 0};
    // Which engine: with `slots` = min(wave slots, tasks) waves at work the patch kernel takes about
    // max(heaviest task, all tasks / slots) point visits per wave, where a task visits its own layers and the ncb - 1 layers
    // the stencils reach beyond them.  For uniform points that is np * (segl + ncb - 1) / segl / slots; the patches keep
    // a point set while their estimate stays within `advantage` x that figure — their measured advantage over the LDS tiles
    // on uniform points.  Short segments in dense regions inflate the visits (a 2-layer segment at
    // m = 8 visits 6 layers), which is why heavily clustered sets still go to the tiles.  advantage <= 0: always the patches.
    const unsigned long long slots_eff = (unsigned long long)std::max(1, std::min(wave_slots, pp.ntasks));
    const double infl0 = (double)(pp.segl + (chi - clo)) / (double)pp.segl;
    const unsigned long long limit = advantage > 0.0 ? (unsigned long long)(advantage * infl0 * (double)np / (double)slots_eff) + 64ull
                                                     : ~0ull / (slots_eff + 1ull);
    return launch_column_tasks(g, ct, offsets, np, limit, limit, slots_eff, true, choice, slots_in_use, colsum, first, tasktab, stream);
}

// The same for the z-marching interpolation ring (march_kernels.h): a task is a workgroup that owns a column of the grid
// for a segment of bin layers.  Points are gathered once, by the task of their own column — cutting a dense column into
// short segments duplicates only the (2M - 1)-plane window load, not point work — so the ring keeps a point set while
// max(heaviest task, all points / workgroups at work) stays within `advantage` x an even share of the whole chip
// (np / cus): small grids, whose few tasks cannot fill the chip, go to the LDS-tile kernel with its slices.
hipError_t launch_march_tasks(const Geom& g, const ColumnTasks& ct, const uint32_t* offsets, int64_t np, int cus, double advantage, double rho_eff_max,
                              uint32_t* choice, uint32_t* colsum, uint32_t* first, uint2* tasktab, hipStream_t stream) {
    const unsigned long long slots_eff = (unsigned long long)std::max(1, std::min(cus, ct.ntasks));
    // advantage <= 0: always the ring (a limit that no product with slots_eff can overflow)
    const bool always = advantage <= 0.0;
    const unsigned long long limit = always ? ~0ull / (slots_eff + 1ull) : (unsigned long long)(advantage * (double)np / (double)cus) + 64ull;
    // tasks of equal point count: x 0.85 for what the estimate leaves out (per-task window loads, the scheduling tail —
    // folded N(0, 1) points at 0.3 points per cell: the ring takes 1.27x its time for uniform points)
    const unsigned long long limit_cut = always ? limit : (unsigned long long)(0.85 * advantage * (double)np / (double)cus) + 64ull;
    return launch_column_tasks(g, ct, offsets, np, limit, limit_cut, slots_eff, always, choice, nullptr, colsum, first, tasktab, stream,
                               always ? 0.0 : rho_eff_max);
}

// The same for the z-marching spreading ring (smarch_kernels.h): a task is a workgroup that owns a column for a segment of bin
// layers and visits the points of the hlo / hhi layers beyond it as well (clipped along z), so short segments in dense
// regions multiply the visits as they do for the patches.  The ring keeps a point set while max(heaviest task, all tasks /
// workgroups at work) stays within `advantage` x an even share of the chip; when it does, the LDS-tile kernel gets no slots.
hipError_t launch_smarch_tasks(const Geom& g, const SMarchPlan& sp, const uint32_t* offsets, int64_t np, int cus, double advantage,
                               uint32_t* choice, uint32_t* slots_in_use, uint32_t* colsum, uint32_t* first, uint2* tasktab, hipStream_t stream) {
    const ColumnTasks& ct = sp.ct;
    const unsigned long long slots_eff = (unsigned long long)std::max(1, std::min(cus, ct.ntasks));
    const double infl0 = (double)(ct.segl + sp.hlo + sp.hhi) / (double)ct.segl;
    const unsigned long long limit = advantage > 0.0 ? (unsigned long long)(advantage * infl0 * (double)np / (double)cus) + 64ull
                                                     : ~0ull / (slots_eff + 1ull);
    return launch_column_tasks(g, ct, offsets, np, limit, limit, slots_eff, advantage <= 0.0, choice, slots_in_use, colsum, first, tasktab, stream);
}

// Zero the interior of the spreading tiles that are processed by several slices (they accumulate with
// atomics; tiles with one slice store every cell exactly once and need no zero fill).
template <typename T>
__global__ __launch_bounds__(256) void zero_split_tiles_kernel(Geom g, int D, int ncr, const uint32_t* __restrict__ nslices,
                                                              T* grid, int64_t grid_stride) {
    const TileShape& ts = g.sp;
    const int tile = blockIdx.x;
    if (nslices[tile] <= 1) return;
    T* gr = grid + (int64_t)blockIdx.y * grid_stride;
    int rem = tile, t[3], org[3], neff[3];
    t[0] = rem % ts.nt[0]; rem /= ts.nt[0];
    t[1] = rem % ts.nt[1]; rem /= ts.nt[1];
    t[2] = rem;
    for (int d = 0; d < 3; ++d) {
        org[d] = t[d] * ts.n[d];
        neff[d] = d < D ? min(ts.n[d], g.Nover[d] - org[d]) : 1;
    }
    const int w_row = ncr * neff[0];
    const int rows = neff[1] * neff[2];
    for (int i = threadIdx.x; i < rows * w_row; i += blockDim.x) {
        const int e = i % w_row, r = i / w_row;
        const int l2 = r % neff[1], l3 = r / neff[1];
        int64_t rowbase = 0;
        if (D >= 2) rowbase = org[1] + l2;
        if (D >= 3) rowbase += (int64_t)(org[2] + l3) * g.Nover[1];
        gr[(rowbase * g.Nover[0] + org[0]) * ncr + e] = T(0);
    }
}

hipError_t launch_zero_split_tiles(int dtype, const Geom& g, int D, int is_complex, int C, const uint32_t* nslices,
                                   void* grid, int64_t grid_stride_reals, hipStream_t stream) {
    const dim3 grid_dim((unsigned)g.sp.ntiles, (unsigned)C, 1);
    const int ncr = is_complex ? 2 : 1;
    if (dtype == NUFFT_F32)
        hipLaunchKernelGGL(zero_split_tiles_kernel<float>, grid_dim, dim3(256), 0, stream, g, D, ncr, nslices,
                           static_cast<float*>(grid), grid_stride_reals);
    else
        hipLaunchKernelGGL(zero_split_tiles_kernel<double>, grid_dim, dim3(256), 0, stream, g, D, ncr, nslices,
                           static_cast<double*>(grid), grid_stride_reals);
    return hipGetLastError();
}

}  // namespace nufft
