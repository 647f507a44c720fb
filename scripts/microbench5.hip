// ds_add_f64 cost per wave instruction as a function of the active lanes (development helper).
// The spreading kernel clips a point's 8 x 8 stencil face to the tile: which part of the 8.8 cycles of a full face
// does an instruction with fewer active lanes cost?  One 1024-thread workgroup per CU, face mapping
// (lane = j1 + 8 * j2, row stride 24 doubles), 8 planes per "point" as in spread_tile_kernel.
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench5.hip -o /tmp/microbench5
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(1024) void k(double* out, int iters, int k1, int k2, long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = 0.0;
    __syncthreads();
    const int j1 = lane & 7, j2 = lane >> 3;
    const bool active = j1 < k1 && j2 < k2;
    const double v = 1.0 + lane * 1e-3;
    int base = wave * 37;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (active) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double* p = tile + ((base + j1 + j2 * 24 + j * 672) & 16383);
                (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        base = (base + 5) & 1023;
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = tile[threadIdx.x];
}

int main() {
    double* out; long long* cyc;
    const int blocks = 256, iters = 2000;
    CK(hipMalloc(&out, blocks * 1024 * sizeof(double)));
    CK(hipMalloc(&cyc, blocks * sizeof(long long)));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const int ks[][2] = {{8, 8}, {6, 8}, {4, 8}, {2, 8}, {8, 6}, {8, 4}, {8, 2}, {6, 6}, {4, 4}, {2, 2}, {1, 1}};
    for (auto& kk : ks) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 131072, 0, out, 10, kk[0], kk[1], cyc);
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 131072, 0, out, iters, kk[0], kk[1], cyc);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // per CU: 16 waves x iters x 8 instructions; shader clock ~2.4 GHz
        const double instr = 16.0 * iters * 8;
        printf("active %d x %d = %2d lanes: %.2f cycles per wave instruction per CU (%.1f lanes/clk)\n", kk[0], kk[1],
               kk[0] * kk[1], ms * 1e-3 * 2.4e9 / instr, kk[0] * kk[1] / (ms * 1e-3 * 2.4e9 / instr));
    }
    return 0;
}
