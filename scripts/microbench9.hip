// LDS atomic rates on gfx950 beyond ds_add_f64 / ds_add_f32 (microbench4.hip): integer adds of 32 and 64 bits, packed
// half adds, and plain read-modify-write by an owning wave (ds_read + v_add + ds_write) — what could replace the 8.5
// cycles per ds_add_f64 wave instruction that bound the LDS-tile spreading kernel?
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench9.hip -o /tmp/microbench9
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: ds_add_f64, 1: ds_add_u64, 2: ds_add_u32, 3: ds_add_f32, 4: read-add-write f64 (non-atomic, wave-owned), 5: read-add-write f32
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int ES = (MODE == 2 || MODE == 3 || MODE == 5) ? 4 : 8;
    constexpr int NE = 131072 / ES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 131072 / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
    __syncthreads();
    const int q = lane & 7, grp = lane >> 3;
    const int off = q + grp * (ES == 8 ? 24 : 40);          // 8 rows x 8 contiguous elements, rows on disjoint banks
    int base = wave * 37;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = (base + off + j * 713) & (NE - 1);
            if constexpr (MODE == 0) (void)__hip_atomic_fetch_add(reinterpret_cast<double*>(smem) + idx, 1.0 + lane * 1e-3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (MODE == 1) (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(smem) + idx, (unsigned long long)(lane + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (MODE == 2) (void)__hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(smem) + idx, (uint32_t)(lane + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (MODE == 3) (void)__hip_atomic_fetch_add(reinterpret_cast<float*>(smem) + idx, 1.0f + lane * 1e-3f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if constexpr (MODE == 4) { volatile double* p = reinterpret_cast<volatile double*>(smem) + idx; *p = *p + (1.0 + lane * 1e-3); }
            if constexpr (MODE == 5) { volatile float* p = reinterpret_cast<volatile float*>(smem) + idx; *p = *p + (1.0f + lane * 1e-3f); }
        }
        base = (base + 5) & 1023;
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)reinterpret_cast<uint32_t*>(smem)[threadIdx.x];
}

template <int MODE>
void run(const char* name) {
    const int iters = 2000, blocks = 256, threads = 1024;
    double* out; CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 131072, 0, out, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
    }
    const double winstr = (double)blocks * (threads / 64) * iters * 8;
    printf("%-44s %7.3f ms  %6.1f cycles @2.4GHz per wave-instr (or read+write pair) per CU\n", name, best, 2.4e9 / (winstr / 256 / (best * 1e-3)));
    CK(hipFree(out));
}

int main() {
    run<0>("ds_add_f64");
    run<1>("ds_add_u64");
    run<2>("ds_add_u32");
    run<3>("ds_add_f32");
    run<4>("ds_read_b64 + v_add_f64 + ds_write_b64");
    run<5>("ds_read_b32 + v_add_f32 + ds_write_b32");
    return 0;
}
