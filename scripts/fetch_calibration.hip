// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths of this library (MI355X_MICROARCH.md, HBM
// section: "exactly 1/2 for wide (16 B per lane) coalesced streaming reads; other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Three kernels stream the same 2 GiB buffer once with 4,
// 8 and 16 bytes per lane (consecutive lanes -> consecutive addresses), a fourth reads 312-byte rows at a 4096-byte
// pitch (the row shape of interp_march_kernel's plane loads at C2).  Run under
//     rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./fetch_calibration
// and compare FETCH_SIZE (KiB) per kernel with the byte counts printed here.
// build: hipcc -O3 --offload-arch=gfx950 scripts/fetch_calibration.hip -o /tmp/fetch_calibration
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename V>
__global__ void stream_read(const V* __restrict__ p, size_t n, V* __restrict__ sink) {
    V acc = V{};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const V v = p[i];
        if constexpr (sizeof(V) == 4) acc += v;
        else if constexpr (sizeof(V) == 8) acc += v;
        else { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    if (threadIdx.x == 0xffff) sink[0] = acc;      // never true: keeps the loads alive
    if constexpr (sizeof(V) == 16) { if (acc.x == 1.2345f) sink[0] = acc; }
    else { if (acc == V(1.2345)) sink[0] = acc; }
}

// rows of 39 doubles (312 B) starting 5 doubles into a 4096-byte pitch: one row per 64-lane wave pass
__global__ void row_read(const double* __restrict__ p, size_t nrows, double* __restrict__ sink) {
    double acc = 0;
    const int lane = threadIdx.x & 63;
    for (size_t r = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; r < nrows; r += (size_t)gridDim.x * (blockDim.x / 64))
        if (lane < 39) acc += p[r * 512 + 5 + lane];
    if (acc == 1.2345) sink[0] = acc;
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    void* buf; void* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(stream_read<float>, dim3(8192), dim3(256), 0, 0, (const float*)buf, bytes / 4, (float*)sink);
        hipLaunchKernelGGL(stream_read<double>, dim3(8192), dim3(256), 0, 0, (const double*)buf, bytes / 8, (double*)sink);
        hipLaunchKernelGGL(stream_read<float4>, dim3(8192), dim3(256), 0, 0, (const float4*)buf, bytes / 16, (float4*)sink);
        hipLaunchKernelGGL(row_read, dim3(8192), dim3(256), 0, 0, (const double*)buf, bytes / 4096, (double*)sink);
    }
    CK(hipDeviceSynchronize());
    printf("stream_read<float/double/float4>: %zu bytes each (%.3f GB); row_read: %zu rows x 312 B = %.3f GB of useful bytes, "
           "%.3f GB if whole 128-byte lines (4 per row) are fetched\n", bytes, bytes / 1e9, bytes / 4096, bytes / 4096 * 312 / 1e9,
           bytes / 4096 * 512 / 1e9);
    return 0;
}
