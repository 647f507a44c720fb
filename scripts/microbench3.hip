// LDS read throughput on gfx950: cycles per wave instruction of ds_read_b32 / b64 / b128 with the access
// patterns of the interpolation kernel (8 segments of 8 reals: aligned rows, random rows, broadcast).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// PAT 0: 8 rows x 8 contiguous, row stride 24 (conflict-free for b64); 1: 8 pseudo-random segments;
// 2: all groups the same segment (broadcast); 3: row stride 31
template <typename V, int PAT>
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    V* tile = reinterpret_cast<V*>(smem);
    constexpr int NE = 131072 / sizeof(V);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 131072 / 8; i += blockDim.x) reinterpret_cast<double*>(smem)[i] = 1.0;
    __syncthreads();
    const int q = lane & 7, grp = lane >> 3;
    int off;
    if (PAT == 0) off = q + grp * 24;
    else if (PAT == 1) off = q + ((grp * 2654435761u) >> 20) % 3000;
    else if (PAT == 2) off = q;
    else off = q + grp * 31;
    int base = wave * 37;
    V acc{};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const V x = tile[(base + off + j * 713) & (NE - 1)];
            if constexpr (sizeof(V) == 16) { acc.x += x.x; acc.y += x.y; }
            else acc += x;
        }
        base = (base + 5) & 1023;
    }
    double r;
    if constexpr (sizeof(V) == 16) r = acc.x + acc.y; else r = (double)acc;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename V, int PAT>
void run(const char* name, int threads) {
    const int iters = 2000, blocks = 256;
    double* out; CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<V, PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<V, PAT>), dim3(blocks), dim3(threads), 131072, 0, out, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
    }
    const double winstr = (double)blocks * (threads / 64) * iters * 8;
    const double cyc = 2.4e9 / (winstr / 256 / (best * 1e-3));
    printf("%-28s threads=%4d: %7.3f ms  %5.1f cycles/wave-instr/CU  %6.1f B/clk/CU\n", name, threads, best, cyc, 64.0 * sizeof(V) / cyc);
    CK(hipFree(out));
}

int main() {
    for (int t : {1024}) {
        run<float, 0>("b32 rows(stride 24)", t);
        run<float, 1>("b32 random segs", t);
        run<float, 2>("b32 broadcast", t);
        run<double, 0>("b64 rows(stride 24)", t);
        run<double, 3>("b64 rows(stride 31)", t);
        run<double, 1>("b64 random segs", t);
        run<double, 2>("b64 broadcast", t);
        run<double2, 0>("b128 rows(stride 24)", t);
        run<double2, 3>("b128 rows(stride 31)", t);
        run<double2, 1>("b128 random segs", t);
        run<double2, 2>("b128 broadcast", t);
    }
    return 0;
}
