// Micro-benchmarks that inform the tile-kernel design (development helper):
//   1. LDS float atomics: ds_add_f64 / ds_add_f32 wave-instruction throughput per CU,
//      conflict-free rows (8 rows x 8 doubles, stride 24) vs same-address vs random;
//   2. LDS reads: ds_read_b64 with the same footprints;
//   3. global float atomics: f64 / f32 adds of a padded tile into a large grid (the flush shape),
//      vs plain stores of the same shape.
// build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics scripts/microbench.hip -o /tmp/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T, int MODE, bool ATOMIC>
__global__ __launch_bounds__(512) void lds_kernel(T* out, int iters, int stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* tile = reinterpret_cast<T*>(smem);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = T(0);
    __syncthreads();
    int off;
    if (MODE == 0) off = (lane & 7) + (lane >> 3) * stride;          // 8x8 face
    else if (MODE == 1) off = 0;                                      // same address
    else if (MODE == 2) off = (lane * 97 + wave * 13) & 4095;         // scattered
    else if (MODE == 3) off = lane;                                   // contiguous 64
    else off = (lane & 7);                                            // MODE 4/5/6: 8 random 8-element segments
    T acc = T(0);
    T v = T(1) + T(lane) * T(1e-3);
    int base = wave * 37;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 4) {
            // pseudo-random segment base per 8-lane group, changing every iteration
            unsigned h = (unsigned)(it * 2654435761u) ^ (unsigned)((lane >> 3) * 40503u + wave * 9176u);
            h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
            const int seg = (int)(h & 2047) * (MODE == 6 ? 8 : 1);       // MODE 6: 64-B aligned segments
            const bool active = MODE != 5 || (lane & 4) == 0;              // MODE 5: half of the lanes masked
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                T* p = tile + ((seg + off + j * 24) & 16383);
                if (active) {
                    if (ATOMIC) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else acc += *p;
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            T* p = tile + ((base + off + j * 648) & 16383);
            if (ATOMIC) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else acc += *p;
        }
        }
        base = (base + 5) & 1023;
    }
    __syncthreads();
    if (!ATOMIC) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    else out[blockIdx.x * blockDim.x + threadIdx.x] = tile[threadIdx.x];
}

template <typename T, bool ATOMIC>
__global__ __launch_bounds__(512) void flush_kernel(T* grid, int N, int P1, int P2, int P3, int n1, int n2, int n3, int nt1, int nt2) {
    // each block flushes one padded tile of ones
    const int t = blockIdx.x;
    const int t1 = t % nt1, t2 = (t / nt1) % nt2, t3 = t / (nt1 * nt2);
    const int o1 = t1 * n1, o2 = t2 * n2, o3 = t3 * n3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int lpr = P1 <= 32 ? 32 : 64;
    const int rpw = 64 / lpr;
    for (int row = wave * rpw + lane / lpr; row < P2 * P3; row += nw * rpw) {
        const int l2 = row % P2, l3 = row / P2;
        int g2 = o2 + l2; if (g2 >= N) g2 -= N;
        int g3 = o3 + l3; if (g3 >= N) g3 -= N;
        const int e = lane % lpr;
        if (e < P1) {
            int g1 = o1 + e; if (g1 >= N) g1 -= N;
            T* p = grid + ((size_t)g3 * N + g2) * N + g1;
            if (ATOMIC) (void)__hip_atomic_fetch_add(p, T(1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *p = T(1);
        }
    }
}

template <typename F>
float time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

template <typename T, int MODE, bool ATOMIC>
void run_lds(const char* name, int threads, int blocks_per_cu, int stride) {
    const int iters = 2000;
    const int blocks = 256 * blocks_per_cu;
    T* out; CK(hipMalloc(&out, sizeof(T) * blocks * threads));
    const size_t lds = 16384 * sizeof(T);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel<T, MODE, ATOMIC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float ms = time_ms([&] { hipLaunchKernelGGL((lds_kernel<T, MODE, ATOMIC>), dim3(blocks), dim3(threads), lds, 0, out, iters, stride); });
    const double winstr = (double)blocks * (threads / 64) * iters * 8;      // wave instructions
    const double per_cu_per_s = winstr / 256 / (ms * 1e-3);
    printf("%-44s threads=%4d blk/CU=%d : %8.3f ms  %.2f wave-instr/ns/CU  (%.1f cycles @2.4GHz per wave-instr per CU)\n", name, threads,
           blocks_per_cu, ms, per_cu_per_s * 1e-9, 2.4e9 / per_cu_per_s);
    CK(hipFree(out));
}

template <typename T, bool ATOMIC>
void run_flush(const char* name, int n1, int n2, int n3, int halo) {
    const int N = 512;
    const int nt1 = (N + n1 - 1) / n1, nt2 = (N + n2 - 1) / n2, nt3 = (N + n3 - 1) / n3;
    T* grid; CK(hipMalloc(&grid, sizeof(T) * (size_t)N * N * N));
    CK(hipMemset(grid, 0, sizeof(T) * (size_t)N * N * N));
    const int P1 = n1 + halo, P2 = n2 + halo, P3 = n3 + halo;
    float ms = time_ms([&] { hipLaunchKernelGGL((flush_kernel<T, ATOMIC>), dim3(nt1 * nt2 * nt3), dim3(512), 0, 0, grid, N, P1, P2, P3, n1, n2, n3, nt1, nt2); });
    const double bytes = (double)nt1 * nt2 * nt3 * P1 * P2 * P3 * sizeof(T);
    printf("%-30s tile (%d,%d,%d)+%d: %8.3f ms, %.2f GB touched -> %.2f TB/s\n", name, n1, n2, n3, halo, ms, bytes * 1e-9, bytes / (ms * 1e-3) * 1e-12);
    CK(hipFree(grid));
}

int main() {
    printf("== LDS atomics / reads (8 per iteration, 16K-element tile) ==\n");
    for (int threads : {256, 512}) {
        run_lds<double, 4, true>("ds_add_f64 8 random 8-double segments", threads, 1, 0);
        run_lds<double, 6, true>("ds_add_f64 8 random aligned segments", threads, 1, 0);
        run_lds<double, 5, true>("ds_add_f64 random segments, half lanes", threads, 1, 0);
        run_lds<double, 4, false>("ds_read_b64 8 random 8-double segments", threads, 1, 0);
        run_lds<double, 6, false>("ds_read_b64 8 random aligned segments", threads, 1, 0);
        run_lds<float, 4, false>("ds_read_b32 8 random 8-float segments", threads, 1, 0);
        run_lds<double, 0, true>("ds_add_f64 8x8 face stride 24", threads, 1, 24);
        run_lds<double, 0, true>("ds_add_f64 8x8 face stride 32", threads, 1, 32);
        run_lds<double, 3, true>("ds_add_f64 contiguous 64", threads, 1, 0);
        run_lds<double, 1, true>("ds_add_f64 same address", threads, 1, 0);
        run_lds<double, 2, true>("ds_add_f64 scattered", threads, 1, 0);
        run_lds<float, 0, true>("ds_add_f32 8x8 face stride 24", threads, 1, 24);
        run_lds<float, 3, true>("ds_add_f32 contiguous 64", threads, 1, 0);
        run_lds<double, 0, false>("ds_read_b64 8x8 face stride 24", threads, 1, 24);
        run_lds<double, 1, false>("ds_read_b64 same address (broadcast)", threads, 1, 0);
        run_lds<float, 0, false>("ds_read_b32 8x8 face stride 24", threads, 1, 24);
    }
    printf("== flush shapes into a 512^3 grid ==\n");
    run_flush<double, true>("f64 atomic add", 17, 20, 19, 7);
    run_flush<double, false>("f64 plain store", 17, 20, 19, 7);
    run_flush<double, true>("f64 atomic add", 12, 12, 12, 7);
    run_flush<double, true>("f64 atomic add", 25, 16, 16, 7);
    run_flush<double, true>("f64 atomic add", 57, 8, 8, 7);
    run_flush<float, true>("f32 atomic add", 17, 20, 19, 7);
    run_flush<float, true>("f32 atomic add", 25, 24, 24, 7);
    run_flush<double, true>("f64 atomic add (no halo)", 32, 16, 16, 0);
    return 0;
}
