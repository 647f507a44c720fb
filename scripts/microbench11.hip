// Random 32-bit atomic adds on a histogram of 2M counters (set_points' bin_count pattern): agent scope on one table (what
// bin_count_kernel does) against workgroup scope on a per-XCD copy of the table (a workgroup only touches the copy of the XCD it
// runs on, read from HW_REG_XCC_ID: the XCD's L2 can then execute the atomic).  Also: random 4-byte reads from a table of the
// same size and from one 8 times larger (the scatter pass would read per-XCD offsets).
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench11.hip -o scripts/bin/microbench11
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}

template <int MODE>   // 0: agent scope, one table; 1: workgroup scope, per-XCD copy; 2: agent scope, per-XCD copy; 3: returning variants of 1
__global__ __launch_bounds__(256) void atom_kernel(uint32_t* counts, uint32_t nbins, int64_t np, uint32_t* rank_out) {
    const int x = xcc_id();
    uint32_t* tab = MODE == 0 ? counts : counts + (size_t)x * nbins;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) {
        const uint32_t bin = hash((uint32_t)p) % nbins;
        uint32_t r;
        if (MODE == 0 || MODE == 2) r = __hip_atomic_fetch_add(&tab[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else r = __hip_atomic_fetch_add(&tab[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (rank_out) rank_out[p] = r | ((uint32_t)x << 28);
    }
}

__global__ __launch_bounds__(256) void read_kernel(const uint32_t* tab, uint32_t n, int64_t np, uint32_t* out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < np; p += stride) acc += tab[hash((uint32_t)p * 3u + 1u) % n];
    if (acc == 0xdeadbeefu) out[0] = acc;
}

int main() {
    const uint32_t nbins = 128 * 128 * 128;
    const int64_t np = 10000000;
    uint32_t *counts, *rank;
    CHECK(hipMalloc(&counts, (size_t)nbins * 8 * 4));
    CHECK(hipMalloc(&rank, np * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto fn, uint32_t* rk, int copies) -> int {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CHECK(hipMemset(counts, 0, (size_t)nbins * 8 * 4));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(fn, dim3(4096), dim3(256), 0, 0, counts, nbins, np, rk);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<uint32_t> h((size_t)nbins * copies);
        CHECK(hipMemcpy(h.data(), counts, h.size() * 4, hipMemcpyDeviceToHost));
        uint64_t sum = 0; for (uint32_t v : h) sum += v;
        printf("%-44s %.3f ms  %.1f G atomics/s   sum %llu (%s)\n", name, best, np / best * 1e-6, (unsigned long long)sum, sum == (uint64_t)np ? "ok" : "LOST UPDATES");
        return 0;
    };
    if (run("agent scope, one table, no return", atom_kernel<0>, nullptr, 1)) return 1;
    if (run("agent scope, one table, returning", atom_kernel<0>, rank, 1)) return 1;
    if (run("agent scope, per-XCD copies, returning", atom_kernel<2>, rank, 8)) return 1;
    if (run("workgroup scope, per-XCD copies, no return", atom_kernel<1>, nullptr, 8)) return 1;
    if (run("workgroup scope, per-XCD copies, returning", atom_kernel<1>, rank, 8)) return 1;
    for (int mult : {1, 8}) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(read_kernel, dim3(4096), dim3(256), 0, 0, counts, nbins * mult, np, rank);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("random 4-byte reads from a %3d MB table          %.3f ms  %.1f G reads/s\n", (int)((size_t)nbins * mult * 4 >> 20), best, np / best * 1e-6);
    }
    return 0;
}
