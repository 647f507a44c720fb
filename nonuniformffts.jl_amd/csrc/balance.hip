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
#include <hipcub/hipcub.hpp>

#include "device_common.h"
#include "kernels.h"

namespace nufft {

// One wave per tile of either tiling (waves [0, nsp): spreading tiles, [nsp, nsp + nip): interpolation
// tiles): number of point visits of a spreading tile (points of all bins within M cells of the interior)
// or number of points of an interpolation tile.
__global__ __launch_bounds__(256) void tile_work_kernel(Geom g, int D, int M, const uint32_t* __restrict__ offsets,
                                                       uint32_t* __restrict__ work) {
    const int nsp = g.sp.ntiles, nip = g.ip.ntiles;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const int lane = threadIdx.x & (kWave - 1);
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
    for (int item = lane; item < nruns; item += kWave) {
        const int sg = item % seg[0].n;
        const int r2 = (item / seg[0].n) % R2;
        const int r3 = item / (seg[0].n * R2);
        const int bin0 = (seg[2].bin(r3) * g.nb[1] + seg[1].bin(r2)) * g.nb[0] + (sg ? seg[0].lo[1] : seg[0].lo[0]);
        sum += offsets[bin0 + (sg ? seg[0].len[1] : seg[0].len[0])] - offsets[bin0];
    }
    for (int o = kWave / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, kWave);
    if (lane == 0) work[wave] = sum;
}

// Slices of tile t: 1 + its share of the extra budget, proportional to its work.  Two kernels: kSumBlocks workgroups sum
// the work of each tiling into partial sums (a global atomic per tile on one address costs 0.4 ms for 3e4 tiles; one
// workgroup doing everything took 0.62 ms for the 7e5 tiles of a 1024^3 grid), then every workgroup of the second kernel
// adds the partial sums up and writes nslices for its tiles; entry nsp + nip is the terminator of the exclusive scan.
constexpr int kSumBlocks = 128;

__global__ __launch_bounds__(256) void tile_work_sums_kernel(const uint32_t* __restrict__ work, int nsp, int nip,
                                                            unsigned long long* __restrict__ part) {
    __shared__ unsigned long long wsum[2][256 / kWave];
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
                                                         uint32_t* __restrict__ nslices) {
    __shared__ unsigned long long tot[2];
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
__global__ __launch_bounds__(64) void fill_desc_kernel(const uint32_t* __restrict__ nslices,
                                                      const uint32_t* __restrict__ desc_off, int nsp, int nip,
                                                      uint32_t ip_base, uint2* __restrict__ desc,
                                                      uint32_t* __restrict__ slots_in_use) {
    const int t = blockIdx.x;
    const uint32_t split = desc_off[nsp];
    const bool interp = t >= nsp;
    const uint32_t S = nslices[t];
    const uint32_t off = interp ? ip_base + (desc_off[t] - split) : desc_off[t];
    const uint32_t tile = interp ? (uint32_t)(t - nsp) : (uint32_t)t;
    for (uint32_t s = threadIdx.x; s < S; s += blockDim.x) desc[off + s] = make_uint2(tile, (s << 16) | S);
    if (t == 0 && threadIdx.x == 0) {
        slots_in_use[0] = split;
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
    const int waves_per_block = 256 / kWave;
    hipLaunchKernelGGL(tile_work_kernel, dim3((unsigned)((n + waves_per_block - 1) / waves_per_block)), dim3(256), 0, stream,
                       b.g, b.D, b.M, b.offsets, b.work);
    // (the partial sums live behind the n + 1 work counters: balance_work_words())
    unsigned long long* part = reinterpret_cast<unsigned long long*>(b.work + balance_work_words(n) - 4 * kSumBlocks);
    hipLaunchKernelGGL(tile_work_sums_kernel, dim3(kSumBlocks), dim3(256), 0, stream, b.work, nsp, nip, part);
    hipLaunchKernelGGL(tile_slices_kernel, dim3((unsigned)((n + 1 + 255) / 256)), dim3(256), 0, stream, b.work, nsp, nip,
                       b.enabled ? b.extra_sp : 0u, b.enabled ? b.extra_ip : 0u, b.smax, part, b.nslices);
    size_t tmp = b.scan_tmp_bytes;
    hipError_t e = hipcub::DeviceScan::ExclusiveSum(b.scan_tmp, tmp, b.nslices, b.desc_off, n + 1, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(fill_desc_kernel, dim3((unsigned)n), dim3(64), 0, stream, b.nslices, b.desc_off, nsp, nip,
                       (uint32_t)nsp + b.extra_sp, b.desc, b.slots_in_use);
    return hipGetLastError();
}

// Which spreading engine serves this point set (plans whose engine is the MFMA patches): the patches have no slices —
// a wave owns its patch for a whole segment of cube layers — so a point set that concentrates in a few patches would
// serialise on them (folded N(0, 1) coordinates, the reference's own benchmark distribution, put 15x the mean into the
// central tasks).  One wave per patch task counts the points of its own bins; the last workgroup to finish compares
// the heaviest task with an even share of the wave slots and writes the verdict: choice[2] = 1 (patches) and no
// slots for the LDS-tile kernel, or choice[2] = 0 and the tile kernel runs as usual.  No host read-back.
__global__ __launch_bounds__(256) void patch_choice_kernel(Geom g, PatchPlan pp, int pby, const uint32_t* __restrict__ offsets,
                                                          unsigned long long np, unsigned long long slots_num,
                                                          unsigned long long share_den, uint32_t* __restrict__ choice,
                                                          uint32_t* __restrict__ slots_in_use) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (wave < pp.ntasks) {
        const int px = wave % pp.npx, py = (wave / pp.npx) % pp.npy, seg = wave / (pp.npx * pp.npy);
        const int bx0 = px * 4, by0 = py * pby;
        const int ncx = min(4, g.nb[0] - bx0), ncy = min(pby, g.nb[1] - by0);
        const int z0 = seg * pp.segl, z1 = min(z0 + pp.segl, g.nb[2]);
        uint32_t sum = 0;
        for (int item = lane; item < ncy * (z1 - z0); item += kWave) {
            const int by = by0 + item % ncy, bz = z0 + item / ncy;
            const int64_t bin0 = ((int64_t)bz * g.nb[1] + by) * g.nb[0] + bx0;
            sum += offsets[bin0 + ncx] - offsets[bin0];
        }
        for (int o = kWave / 2; o > 0; o >>= 1) sum += __shfl_down(sum, o, kWave);
        if (lane == 0) atomicMax(&choice[0], sum);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t ticket = atomicAdd(&choice[1], 1u);
        if (ticket == gridDim.x - 1) {
            __threadfence();
            const unsigned long long mx = atomicExch(&choice[0], 0u);
            choice[1] = 0u;
            // heaviest task <= np * slots_num / share_den (+ 64 points: tiny point sets fluctuate)
            const bool patches = mx * share_den <= np * slots_num + 64ull * share_den || np == 0;
            choice[2] = patches ? 1u : 0u;
            if (patches) slots_in_use[0] = 0u;
        }
    }
}

hipError_t launch_patch_choice(const Geom& g, const PatchPlan& pp, int pby, const uint32_t* offsets, int64_t np, int wave_slots,
                               uint32_t* choice, uint32_t* slots_in_use, hipStream_t stream) {
    const int waves_per_block = 256 / kWave;
    // Patches while the heaviest task holds at most one even share of the wave slots, max_task <= np / wave_slots — or,
    // on grids with fewer tasks than that (no point set could meet the first bound: the mean task already exceeds it),
    // at most twice the mean task, max_task <= 2 np / ntasks.  A patch task is not shared between workgroups, so its
    // heaviest task bounds the kernel; the LDS tiles split heavy tiles into slices.
    unsigned long long num = 1ull, den = (unsigned long long)wave_slots;
    if (2ull * (unsigned long long)wave_slots > (unsigned long long)pp.ntasks) { num = 2ull; den = (unsigned long long)pp.ntasks; }
    hipLaunchKernelGGL(patch_choice_kernel, dim3((unsigned)((pp.ntasks + waves_per_block - 1) / waves_per_block)), dim3(256), 0, stream,
                       g, pp, pby, offsets, (unsigned long long)np, num, den, choice, slots_in_use);
    return hipGetLastError();
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
