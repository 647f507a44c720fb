// LDS atomic flavours on gfx950: is integer accumulation cheaper than ds_add_f64?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int KIND>
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* tile = reinterpret_cast<double*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = 0.0;
    __syncthreads();
    const int off = (lane & 7) + (lane >> 3) * 24;
    int base = wave * 37;
    double v = 1.0 + lane * 1e-3;
    unsigned long long vi = 12345ull + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double* p = tile + ((base + off + j * 648) & 16383);
            if (KIND == 0) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == 1) (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), vi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == 2) (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned int*>(p), (unsigned)vi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (KIND == 3) *p = v;                                  // plain ds_write_b64
            else if (KIND == 4) { double o = *p; *p = o + v; }           // non-atomic read-modify-write
            else if (KIND == 5) (void)__hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        base = (base + 5) & 1023;
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = tile[threadIdx.x];
}

template <int KIND>
void run(const char* name, int threads) {
    const int iters = 2000, blocks = 256;
    double* out; CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 131072, 0, out, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
    }
    const double winstr = (double)blocks * (threads / 64) * iters * 8;
    printf("%-34s threads=%4d: %7.3f ms  %.1f cycles @2.4GHz per wave-instr per CU\n", name, threads, best, 2.4e9 / (winstr / 256 / (best * 1e-3)));
    CK(hipFree(out));
}

int main() {
    for (int t : {512, 1024}) {
        run<0>("ds_add_f64", t);
        run<1>("ds_add_u64", t);
        run<2>("ds_add_u32", t);
        run<3>("ds_write_b64", t);
        run<4>("ds_read_b64 + add + ds_write_b64", t);
        run<5>("ds_max_f64", t);
    }
    return 0;
}
