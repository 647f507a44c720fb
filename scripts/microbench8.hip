// FP32 matrix instructions on gfx950 for the Float32 spreading patches: operand layouts of v_mfma_f32_16x16x4_f32 and
// v_mfma_f32_4x4x1_16b_f32, their issue rates, and what else issues beside them (v_fma_f32 / v_mul_f32, LDS reads) —
// next to v_mfma_f64_4x4x4_4b_f64 with Float32 vector work beside it.
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench8.hip -o scripts/bin/microbench8
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void layout16(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
__global__ void layout4(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    v4f acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

// MODE 0: 16x16x4 f32; 1: 4x4x1_16b f32; 2: f64 4x4x4_4b; NV = v_fma_f32 per MFMA, NL = ds_read_b32 per MFMA
template <int NACC, int MODE, int NV, int NL>
__global__ __launch_bounds__(1024) void rate(float* out, int iters, long long* cycles) {
    __shared__ float lds[4096];
    const int l = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f + i * 1e-4f;
    float a = 1.0f + l * 1e-3f, b = 1.0f - l * 1e-3f;
    double ad = a, bd = b;
    v4f acc[NACC];
    double s[NACC];
    float f[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = a + i;
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc[i] = v4f{0, 0, 0, 0}; s[i] = 0; }
    float lsum = 0.f;
    const float* lp = lds + l;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if constexpr (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
            if constexpr (MODE == 2) s[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(ad, bd, s[i], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; ++v) f[v & 7] = fmaf(f[v & 7], a, b);
#pragma unroll
            for (int q = 0; q < NL; ++q) {
                float x;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x) : "v"((uint32_t)(uintptr_t)lp), "n"(256 * ((0 * 7 + 1) % 15)));
                lsum += x;
            }
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    float r = lsum;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += f[i];
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + (float)s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NACC, int MODE, int NV, int NL>
void run_rate(const char* name, int threads) {
    const int blocks = 256, iters = 4000;
    float* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(float) * blocks * 1024));
    CK(hipMalloc(&cyc, sizeof(long long) * blocks));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate<NACC, MODE, NV, NL>), dim3(blocks), dim3(threads), 0, 0, out, 10, cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate<NACC, MODE, NV, NL>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const int waves_per_simd = threads / 64 / 4 > 0 ? threads / 64 / 4 : 1;
    const double per_simd = (double)iters * NACC * waves_per_simd;
    printf("%-58s %d waves/SIMD: %7.3f ms -> %6.1f cycles@2.4GHz per MFMA group per SIMD\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / per_simd);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    std::vector<float> a(64), b(64), d(256);
    srand(1);
    for (int i = 0; i < 64; ++i) { a[i] = (rand() % 1000) / 100.0f; b[i] = (rand() % 1000) / 100.0f; }
    float *da, *db, *dd;
    CK(hipMalloc(&da, 64 * 4)); CK(hipMalloc(&db, 64 * 4)); CK(hipMalloc(&dd, 256 * 4));
    CK(hipMemcpy(da, a.data(), 64 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 64 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout16, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost));
    {
        double ref[16][16];
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            ref[i][j] = 0;
            for (int k = 0; k < 4; ++k) ref[i][j] += (double)a[i + 16 * k] * b[j + 16 * k];
        }
        int okA = 1, okB = 1;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const double v = d[l * 4 + r];
            if (fabs(v - ref[4 * (l / 16) + r][l % 16]) > 1e-2) okA = 0;
            if (fabs(v - ref[4 * r + l / 16][l % 16]) > 1e-2) okB = 0;
        }
        printf("16x16x4 f32: A[i][k] lane i+16k, B[k][j] lane j+16k;  D[4*(l/16)+r][l%%16]: %s;  D[4*r+l/16][l%%16]: %s\n", okA ? "MATCH" : "no", okB ? "MATCH" : "no");
    }
    hipLaunchKernelGGL(layout4, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(d.data(), dd, 256 * 4, hipMemcpyDeviceToHost));
    {
        // hypotheses: block = l / 4, A_b[i] lane 4 b + i, B_b[j] lane 4 b + j; D_b[i][j]: (1) reg r = i, lane 4 b + j; (2) reg r = j, lane 4 b + i
        int ok1 = 1, ok2 = 1;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int bb = l / 4, x = l % 4;
            const double v = d[l * 4 + r];
            if (fabs(v - (double)a[4 * bb + r] * b[4 * bb + x]) > 1e-2) ok1 = 0;
            if (fabs(v - (double)a[4 * bb + x] * b[4 * bb + r]) > 1e-2) ok2 = 0;
        }
        printf("4x4x1_16b f32: block = l/4; D_b[i = r][j = l%%4]: %s;  D_b[i = l%%4][j = r]: %s\n", ok1 ? "MATCH" : "no", ok2 ? "MATCH" : "no");
        if (!ok1 && !ok2)
            for (int l = 0; l < 8; ++l) for (int r = 0; r < 4; ++r)
                for (int x = 0; x < 64; ++x) for (int y = 0; y < 64; ++y)
                    if (fabs(d[l * 4 + r] - (double)a[x] * b[y]) < 1e-3) printf("  lane %d reg %d = a[%d] b[%d]\n", l, r, x, y);
    }
    for (int threads : {256, 512}) {
        run_rate<8, 0, 0, 0>("16x16x4 f32, 8 acc", threads);
        run_rate<2, 0, 0, 0>("16x16x4 f32, 2 acc", threads);
        run_rate<1, 0, 0, 0>("16x16x4 f32, 1 acc (dependent)", threads);
        run_rate<8, 0, 4, 0>("16x16x4 f32, 8 acc + 4 v_fma_f32 each", threads);
        run_rate<8, 0, 8, 0>("16x16x4 f32, 8 acc + 8 v_fma_f32 each", threads);
        run_rate<8, 0, 0, 2>("16x16x4 f32, 8 acc + 2 ds_read_b32 each", threads);
        run_rate<8, 0, 4, 2>("16x16x4 f32, 8 acc + 4 v_fma_f32 + 2 ds_read_b32 each", threads);
        run_rate<8, 1, 0, 0>("4x4x1_16b f32, 8 acc", threads);
        run_rate<1, 1, 0, 0>("4x4x1_16b f32, 1 acc (dependent)", threads);
        run_rate<8, 1, 1, 0>("4x4x1_16b f32, 8 acc + 1 v_fma_f32 each", threads);
        run_rate<8, 1, 2, 0>("4x4x1_16b f32, 8 acc + 2 v_fma_f32 each", threads);
        run_rate<8, 1, 1, 1>("4x4x1_16b f32, 8 acc + 1 v_fma_f32 + 1 ds_read_b32 each", threads);
        run_rate<8, 2, 0, 0>("4x4x4_4b f64, 8 acc", threads);
        run_rate<8, 2, 4, 0>("4x4x4_4b f64, 8 acc + 4 v_fma_f32 each", threads);
        run_rate<8, 2, 2, 1>("4x4x4_4b f64, 8 acc + 2 v_fma_f32 + 1 ds_read_b32 each", threads);
    }
    return 0;
}
