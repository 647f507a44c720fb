// v_mfma_f64_16x16x4_f64 / v_mfma_f64_4x4x4_4b_f64 on gfx950: operand layout and issue rate.
// Question behind it: spreading is a contraction over points, G[(x,y), z] += sum_p A[(x,y), p] B[p, z] with
// A = v w1 w2, B = w3 — can the matrix pipe take the accumulation that ds_add_f64 (8.5 cycles per wave
// instruction per CU) bounds today?
// build: hipcc -O3 --offload-arch=gfx950 scripts/microbench6.hip -o /tmp/microbench6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void layout16(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    v4d acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}

__global__ void layout4(const double* a, const double* b, double* d) {
    const int l = threadIdx.x;
    double acc = 0;
    acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], acc, 0, 0, 0);
    d[l] = acc;
}

// NACC independent accumulators per wave, `iters` rounds; MODE 0: MFMA only, 1: MFMA + 4 f64 FMAs per MFMA,
// 2: the VALU FMAs only, 3: 4x4x4 MFMA only
template <int NACC, int MODE>
__global__ __launch_bounds__(1024) void rate(double* out, int iters, long long* cycles) {
    const int l = threadIdx.x & 63;
    double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
    v4d acc[NACC];
    double s[NACC];
    double f0 = a, f1 = b, f2 = a * b, f3 = a + b;
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc[i] = v4d{0, 0, 0, 0}; s[i] = 0; }
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (MODE == 0 || MODE == 1) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            if constexpr (MODE == 3) s[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s[i], 0, 0, 0);
            if constexpr (MODE == 1 || MODE == 2) {
                f0 = fma(f0, a, b); f1 = fma(f1, a, b); f2 = fma(f2, a, b); f3 = fma(f3, a, b);
            }
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    double r = f0 + f1 + f2 + f3;
#pragma unroll
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int NACC, int MODE>
void run_rate(const char* name, int threads) {
    const int blocks = 256, iters = 4000;
    double* out; long long* cyc;
    CK(hipMalloc(&out, sizeof(double) * blocks * 1024));
    CK(hipMalloc(&cyc, sizeof(long long) * blocks));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, out, 10, cyc);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate<NACC, MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c0; CK(hipMemcpy(&c0, cyc, sizeof(long long), hipMemcpyDeviceToHost));
    const int waves_per_simd = threads / 64 / 4 > 0 ? threads / 64 / 4 : 1;
    const double per_simd = (double)iters * NACC * waves_per_simd;      // instructions (or groups) per SIMD
    printf("%-44s threads %4d (%d waves/SIMD): %7.3f ms, clock64 %lld -> %.1f clock64-ticks, %.1f cycles@2.4GHz per MFMA(-group) per SIMD\n",
           name, threads, waves_per_simd, ms, c0, (double)c0 / per_simd, ms * 1e-3 * 2.4e9 / per_simd);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    // ---- layout ----
    std::vector<double> a(64), b(64), d(256);
    srand(1);
    for (int i = 0; i < 64; ++i) { a[i] = (rand() % 1000) / 100.0; b[i] = (rand() % 1000) / 100.0; }
    double *da, *db, *dd;
    CK(hipMalloc(&da, 64 * 8)); CK(hipMalloc(&db, 64 * 8)); CK(hipMalloc(&dd, 256 * 8));
    CK(hipMemcpy(da, a.data(), 64 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, b.data(), 64 * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(layout16, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(d.data(), dd, 256 * 8, hipMemcpyDeviceToHost));
    // hypothesis: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k
    double ref[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        ref[i][j] = 0;
        for (int k = 0; k < 4; ++k) ref[i][j] += a[i + 16 * k] * b[j + 16 * k];
    }
    int okA = 1, okB = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const double v = d[l * 4 + r];
        if (fabs(v - ref[4 * (l / 16) + r][l % 16]) > 1e-9) okA = 0;      // D[i = 4 (l / 16) + r][j = l % 16]
        if (fabs(v - ref[4 * r + l / 16][l % 16]) > 1e-9) okB = 0;        // D[i = 4 r + l / 16][j = l % 16]
    }
    printf("16x16x4 f64: A[i][k] lane i+16k, B[k][j] lane j+16k;  D[4*(l/16)+r][l%%16]: %s;  D[4*r+l/16][l%%16]: %s\n",
           okA ? "MATCH" : "no", okB ? "MATCH" : "no");
    if (!okA && !okB) {
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const double v = d[l * 4 + r];
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j)
                if (fabs(v - ref[i][j]) < 1e-9 && l < 20) printf("  lane %d reg %d = D[%d][%d]\n", l, r, i, j);
        }
    }
    hipLaunchKernelGGL(layout4, dim3(1), dim3(64), 0, 0, da, db, dd);
    CK(hipMemcpy(d.data(), dd, 64 * 8, hipMemcpyDeviceToHost));
    // 4 blocks; hypothesis: block = l / 16?  try: A_b[i][k]: lane?  brute force over a few conventions
    {
        // convention 1: block b = l % 4?? ; print for inspection the match of D lane l with sums over k of a[x]*b[y]
        int ok1 = 1, ok2 = 1;
        for (int l = 0; l < 64; ++l) {
            // conv 1: blocks along l/16? no: for 4x4x4_4b: i = l % 4, block = (l / 4) % 4 ... use search below
            (void)l;
        }
        // generic search: D(l) = sum_k a[la(l,k)] * b[lb(l,k)]; find for each l the set of (la, lb) products
        for (int l = 0; l < 8; ++l) {
            printf("  4x4x4 lane %d = %.4f; candidates:", l, d[l]);
            for (int la0 = 0; la0 < 64; ++la0) for (int lb0 = 0; lb0 < 64; ++lb0)
                for (int sa = 1; sa <= 16; sa *= 2) for (int sb = 1; sb <= 16; sb *= 2) {
                    if (la0 + 3 * sa > 63 || lb0 + 3 * sb > 63) continue;
                    double s = 0;
                    for (int k = 0; k < 4; ++k) s += a[la0 + k * sa] * b[lb0 + k * sb];
                    if (fabs(s - d[l]) < 1e-9) printf(" [a %d+%dk, b %d+%dk]", la0, sa, lb0, sb);
                }
            printf("\n");
        }
        (void)ok1; (void)ok2;
    }
    // ---- rates ----
    for (int threads : {256, 512, 1024}) {
        run_rate<1, 0>("16x16x4 f64, 1 accumulator (dependent)", threads);
        run_rate<4, 0>("16x16x4 f64, 4 accumulators", threads);
        run_rate<8, 0>("16x16x4 f64, 8 accumulators", threads);
        run_rate<4, 1>("16x16x4 f64 x4 acc + 4 v_fma_f64 each", threads);
        run_rate<4, 2>("4 v_fma_f64 groups only", threads);
        run_rate<4, 3>("4x4x4_4b f64, 4 accumulators", threads);
        run_rate<1, 3>("4x4x4_4b f64, 1 accumulator (dependent)", threads);
    }
    return 0;
}
