// ds_read_b64 vs ds_read_b128 (16-byte aligned and 8-byte aligned addresses) for the interpolation gather: every group of
// lanes reads contiguous doubles of a row at a random start (the stencil row of a point), rows 31 doubles apart.
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench7.hip -o scripts/bin/microbench7
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));

// MODE 0: 8 lanes per point, b64; 1: 4 lanes per point, b128 (start forced even: aligned); 2: same, arbitrary start (8-byte aligned only)
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, int iters, const int* starts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int G = MODE == 0 ? 8 : 4;
    const int grp = lane / G, q = lane % G;
    double acc = 0;
    for (int it = 0; it < iters; ++it) {
        int s = starts[(it * 16 + wave) * 64 + grp] ;            // random start of the point's stencil in the tile
        if (MODE == 1) s &= ~1;
        const int base = (s % 8000) + (MODE == 0 ? q : 2 * q);
#pragma unroll
        for (int r = 0; r < 16; ++r) {                              // 16 rows of the 64 (a quarter of a point's stencil)
            if constexpr (MODE == 0) {
                double v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((unsigned)((base + r * 31) * 8)));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc += v;
            } else {
                v2d v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)((base + r * 31) * 8)));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc += v.x + v.y;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, const int* dstarts) {
    const int iters = 2000, blocks = 256;
    double* out; CK(hipMalloc(&out, sizeof(double) * blocks * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(1024), 131072, 0, out, iters, dstarts);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
    }
    // bytes read per CU: 16 waves * iters * 16 rows * (64 lanes * 8 or 16 B)
    const double bytes = 16.0 * iters * 16 * 64 * (MODE == 0 ? 8 : 16);
    const double points = 16.0 * iters * (64 / (MODE == 0 ? 8 : 4));      // quarter-stencils of points per CU
    printf("%-44s %7.3f ms  %6.1f B/clk/CU @2.4GHz   %6.1f cycles per quarter-stencil of a point per CU\n", name, best,
           bytes / (best * 1e-3 * 2.4e9), best * 1e-3 * 2.4e9 / points);
    double h; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
    CK(hipFree(out));
}

int main() {
    const int n = 2000 * 16 * 64;
    int* h = (int*)malloc(n * sizeof(int));
    srand(3);
    for (int i = 0; i < n; ++i) h[i] = rand() % 8000;
    int* d; CK(hipMalloc(&d, n * sizeof(int))); CK(hipMemcpy(d, h, n * sizeof(int), hipMemcpyHostToDevice));
    run<0>("ds_read_b64, 8 lanes per point", d);
    run<1>("ds_read_b128, 4 lanes per point, aligned", d);
    run<2>("ds_read_b128, 4 lanes per point, 8-byte aligned", d);
    return 0;
}
