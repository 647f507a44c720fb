// Can the FP64 matrix pipe beat the LDS atomic floor of the C2 spreading window (8 ds_add_f64 per point x 8.5
// array cycles = 68 of the kernel's 115 CU-cycles per point)?  The scheme of the round-4 review: the points of a 4^3 bin are
// accumulated in REGISTERS first — C[128 face rows of the bin's 11 x 11 footprint x 16 z-planes] += A[face, 4 points] B[4 points, z]
// with v_mfma_f64_16x16x4, 8 instructions per 4 points — and the bin's 11 x 11 x 11 footprint is flushed to the LDS window once, with
// ds_add_f64 from the accumulator layout (lane l, register r of tile t holds face row 16 t + 4 (l / 16) + r, plane l % 16).
// This microbenchmark times the two inner loops alone, one workgroup of 16 waves per CU, on a 133-KB window like the kernel's:
//   A  "atomics":  per point 8 ds_add_f64 wave instructions (64 lanes = the 8 x 8 face, planes at immediate offsets) + 8 products
//   B  "matrix":   per bin of n points: ceil(n / 4) x 8 MFMA (operands formed by one product per lane and instruction), then the flush:
//                  8 tiles x 4 registers = 32 ds_add_f64 wave instructions, lanes outside the 11 x 11 x 11 footprint masked off
//   B' the same with the z-planes of the flush on a padded plane stride (bank spread)
// and prints CU-cycles per point at n = 5 (C2: 4.8 points per bin), 19 (the reference's benchmark density, 0.3 per cell) and 64.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics scripts/microbench12.hip -o scripts/bin/microbench12
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double D4 __attribute__((ext_vector_type(4)));
constexpr int RS = 40, WY = 39, PS = RS * WY, RZ = 11;          // the window of spread_march_kernel<double, false, 4, ., true, true>
constexpr int kThreads = 1024;

__device__ __forceinline__ uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// A: the atomic stream of today's kernel
__global__ __launch_bounds__(kThreads) void atomics_kernel(int npoints_per_wave, double* out, long long* cycles) {
    extern __shared__ double ring[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < RZ * PS; i += kThreads) ring[i] = 0.0;
    __syncthreads();
    const int j1 = lane & 7, j2 = lane >> 3;
    const long long t0 = clock64();
    double w = 1.0 + lane * 1e-3;
    for (int p = 0; p < npoints_per_wave; ++p) {
        const uint32_t h = hash((uint32_t)(p * 16 + wave) * 2654435761u + blockIdx.x);
        const int sx = h % 32, sy = (h >> 8) % 31, dz = (h >> 16) & 3;        // stencil start inside the window, slot of its first plane
        double* base = ring + (sy + j2) * RS + sx + j1 + dz * PS;
        const double v = w * (1.0 + (h & 255) * 1e-3);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < 7 || dz < 4) atomicAdd(base + k * PS, v * (1.0 + k * 0.125));    // (all eight planes inside the 11-plane window: dz + 7 <= 10)
        }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (tid < 64) out[blockIdx.x * 64 + tid] = ring[tid * 17];
}

// B: registers first (matrix pipe), one flush per bin
template <bool PADZ>
__global__ __launch_bounds__(kThreads) void matrix_kernel(int nbins_per_wave, int npts, double* out, long long* cycles) {
    extern __shared__ double ring[];
    constexpr int PSZ = PADZ ? PS + 2 : PS;             // plane stride of the flush: + 16 bytes per plane spreads 16 planes over the banks
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < RZ * PSZ + 64; i += kThreads) ring[i] = 0.0;
    __syncthreads();
    const int zl = lane & 15, rq = lane >> 4;
    const long long t0 = clock64();
    for (int b = 0; b < nbins_per_wave; ++b) {
        const uint32_t h = hash((uint32_t)(b * 16 + wave) * 2654435761u + blockIdx.x);
        const int bx = (h % 8) * 4, by = ((h >> 8) % 7) * 4;                       // the bin's footprint origin inside the window
        D4 acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = D4{0.0, 0.0, 0.0, 0.0};
        for (int p0 = 0; p0 < npts; p0 += 4) {
            // operands of this batch of 4 points: A[face row][point] = w1 w2 (one product per lane and tile), B[point][plane] = v w3
            const double w3v = 1.0 + zl * 0.01 + p0 * 1e-3;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double a = (1.0 + (lane & 15) * 0.01 + t) * (1.0 + rq * 0.1 + p0 * 1e-3);
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, w3v, acc[t], 0, 0, 0);
            }
        }
        // flush: face row f = 16 t + 4 (l / 16) + r -> (f % 11, f / 11) of the 11 x 11 footprint, plane l % 16 (< 11)
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = 16 * t + 4 * rq + r;
                if (f < 121 && zl < 11) atomicAdd(ring + (by + f / 11) * RS + bx + f % 11 + zl * PSZ, acc[t][r]);
            }
    }
    __syncthreads();
    const long long t1 = clock64();
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
    if (tid < 64) out[blockIdx.x * 64 + tid] = ring[tid * 17];
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int ncu = prop.multiProcessorCount;
    double* out;
    long long* cyc;
    CHECK(hipMalloc(&out, (size_t)ncu * 64 * 8));
    CHECK(hipMalloc(&cyc, (size_t)ncu * 8));
    const size_t lds = (size_t)(RZ * (PS + 2) + 64) * 8;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(atomics_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(matrix_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(matrix_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<long long> h(ncu);
    auto mean_cycles = [&]() -> double {
        (void)hipMemcpy(h.data(), cyc, (size_t)ncu * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (long long v : h) s += (double)v;
        return s / ncu;
    };
    // clock64() counts at the constant 100 MHz reference on gfx9: convert with the wall time of the launch instead
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto timed = [&](auto launch) -> float {
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CHECK(hipDeviceSynchronize());
            (void)hipEventRecord(e0);
            launch();
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        return best;
    };
    const double ghz = 2.4;          // nominal shader clock (the spreading kernel itself runs at about this clock)
    const int ppw = 20000;           // points per wave
    {
        const float ms = timed([&] { hipLaunchKernelGGL(atomics_kernel, dim3(ncu), dim3(kThreads), lds, 0, ppw, out, cyc); });
        const double pts_per_cu = 16.0 * ppw;
        printf("A  atomics only (8 ds_add_f64 per point)            : %.3f ms, %.1f CU-cycles per point at %.1f GHz\n", ms, ms * 1e-3 * ghz * 1e9 / pts_per_cu, ghz);
    }
    for (int n : {5, 19, 64}) {
        const int bins = ppw / n;
        const double pts_per_cu = 16.0 * bins * n;
        const float m0 = timed([&] { hipLaunchKernelGGL(matrix_kernel<false>, dim3(ncu), dim3(kThreads), lds, 0, bins, n, out, cyc); });
        const float m1 = timed([&] { hipLaunchKernelGGL(matrix_kernel<true>, dim3(ncu), dim3(kThreads), lds, 0, bins, n, out, cyc); });
        printf("B  matrix + flush, %2d points per bin (%d MFMA + 32 ds_add_f64 per bin): %.3f ms = %.1f CU-cycles per point;  padded plane stride: %.3f ms = %.1f\n",
               n, 8 * ((n + 3) / 4), m0, m0 * 1e-3 * ghz * 1e9 / pts_per_cu, m1, m1 * 1e-3 * ghz * 1e9 / pts_per_cu);
    }
    (void)mean_cycles;
    return 0;
}
