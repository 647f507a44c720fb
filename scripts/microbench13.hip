// Would a column-first pre-sort make the fine bin sort cheaper?  set_points' two expensive transactions — one returning histogram atomic and one
// record store per point — run at ~26 G/s when their addresses are random over the whole table (scripts/microbench11.hip).  After a first pass
// that groups the points by columns of 4 x 4 bins (C3: 4096 columns of 24 414 points, 4096 fine bins = 16 KB of counters and 390 KB of records
// each) the same transactions of a workgroup stay inside one column.  Measured here: 1e8 returning atomics and 1e8 16-byte record stores,
// (a) random over the whole table / array, (b) random inside the column of the point (points visited in column order).
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench13.hip -o scripts/bin/microbench13
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

constexpr uint32_t kCols = 4096, kBinsPerCol = 4096, kBins = kCols * kBinsPerCol;

template <bool LOCAL>
__global__ __launch_bounds__(256) void count_kernel(uint32_t* counts, int64_t np, uint32_t per_col, uint32_t* rank) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) {
        const uint32_t h = hash((uint32_t)p);
        const uint32_t col = min((uint32_t)(p / per_col), kCols - 1u);
        const uint32_t bin = LOCAL ? col * kBinsPerCol + h % kBinsPerCol : h % kBins;
        rank[p] = atomicAdd(&counts[bin], 1u);
    }
}
template <bool LOCAL>
__global__ __launch_bounds__(256) void scatter_kernel(uint4* out, int64_t np, uint32_t per_col) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) {
        const uint32_t h = hash((uint32_t)p * 3u + 1u);
        const int64_t col = min<int64_t>(p / per_col, kCols - 1);
        const int64_t dst = LOCAL ? col * per_col + h % per_col : (int64_t)(((uint64_t)h * (uint64_t)np) >> 32);
        out[dst] = make_uint4((uint32_t)p, h, 0u, 0u);
    }
}

int main() {
    const int64_t np = 100000000;
    const uint32_t per_col = (uint32_t)(np / kCols);
    uint32_t *counts, *rank;
    uint4* out;
    CHECK(hipMalloc(&counts, (size_t)kBins * 4));
    CHECK(hipMalloc(&rank, np * 4));
    CHECK(hipMalloc(&out, np * 16));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) -> int {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHECK(hipMemset(counts, 0, (size_t)kBins * 4));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-64s %.3f ms  %.1f G/s\n", name, best, np / best * 1e-6);
        return 0;
    };
    if (run("returning atomics, random over 16.7M counters (67 MB)", [&] { hipLaunchKernelGGL(count_kernel<false>, dim3(8192), dim3(256), 0, 0, counts, np, per_col, rank); })) return 1;
    if (run("returning atomics, inside the point's column (16 KB each)", [&] { hipLaunchKernelGGL(count_kernel<true>, dim3(8192), dim3(256), 0, 0, counts, np, per_col, rank); })) return 1;
    if (run("16-byte record stores, random over 1.6 GB", [&] { hipLaunchKernelGGL(scatter_kernel<false>, dim3(8192), dim3(256), 0, 0, out, np, per_col); })) return 1;
    if (run("16-byte record stores, inside the point's column (390 KB each)", [&] { hipLaunchKernelGGL(scatter_kernel<true>, dim3(8192), dim3(256), 0, 0, out, np, per_col); })) return 1;
    return 0;
}
