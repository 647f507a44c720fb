// Vector FP32 issue rates on gfx950: what does a v_pk_fma_f32 cost next to a v_fma_f32, in independent chains and in
// one dependent chain, at 1 / 2 / 4 waves per SIMD?  (The complex Float32 interpolation gathers with one v_pk_fma_f32 per
// stencil node; its ISA shows an s_nop between every pair of dependent packed FMAs.)
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench10.hip -o /tmp/microbench10
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 0: 8 independent v_fma_f32 chains; 1: 8 independent v_pk_fma_f32 chains; 2: one dependent v_fma_f32 chain;
// 3: one dependent v_pk_fma_f32 chain; 4: two interleaved v_pk_fma_f32 chains; 5: 8 independent v_fma_f64 chains;
// 6: 8 independent v_pk_mul_f32
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float s) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8];
    v2f p[8];
    double d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = tid * 1e-6f + i; p[i] = v2f{a[i], a[i] + 1.f}; d[i] = a[i]; }
    const float m = s;
    const v2f mp = {s, s * 1.0001f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {      // 64 instructions per loop iteration (the taken branch costs a single wave ~30 cycles)
        if constexpr (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(m));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mp), "v"(mp));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(m));
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("s_nop 0\n\tv_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[0]) : "v"(mp), "v"(mp));
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 1]) : "v"(mp), "v"(mp));
        } else if constexpr (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"((double)m));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(mp));
        }
      }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + p[i][0] + p[i][1] + (float)d[i];
    out[tid] = r;
}

template <int MODE>
void run(const char* name) {
    const int iters = 2500, blocks = 256;
    float* out; CK(hipMalloc(&out, sizeof(float) * blocks * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int threads : {256, 512, 1024}) {
        float best = 1e30f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999f);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); if (r && ms < best) best = ms;
        }
        // wave instructions per SIMD: (threads / 64 / 4) waves x iters x 8
        const double per_simd = (double)(threads / 64 / 4) * iters * 64;
        printf("%-44s %d wave(s)/SIMD %8.3f ms  %5.2f cycles @2.4GHz per wave instruction per SIMD\n", name, threads / 256, best,
               2.4e9 * best * 1e-3 / per_simd);
    }
    CK(hipFree(out));
}

int main() {
    run<0>("v_fma_f32, 8 independent chains");
    run<1>("v_pk_fma_f32, 8 independent chains");
    run<2>("v_fma_f32, one dependent chain");
    run<3>("s_nop 0 + v_pk_fma_f32, one dependent chain");
    run<4>("v_pk_fma_f32, two interleaved chains");
    run<5>("v_fma_f64, 8 independent chains");
    run<6>("v_pk_mul_f32, 8 independent");
    return 0;
}
