// Deconvolution kernels between the oversampled spectrum and the caller's uniform arrays.
//
//   deconv_truncate_kernel : copy_deconvolve_to_non_oversampled_kernel!  (reference
//                            src/NonuniformFFTs.jl:387-403): truncation + deconvolution + FFT
//                            normalisation fused, one coalesced write of the output.
//   deconv_pad_kernel      : fill_with_zeros_kernel! (:116-122, launched :260-266) fused with
//                            copy_deconvolve_to_oversampled_kernel! (:453-469): every element of the
//                            oversampled spectrum is written exactly once (value or zero), instead
//                            of a full zero fill followed by a scatter.
// Rows (fixed i2, i3) map to blockIdx.y / .z so that no integer division is needed and whole
// zero rows of the padded spectrum are recognised per workgroup.
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace nufft {

template <typename T> struct Cplx;
template <> struct Cplx<float>  { using type = float2; };
template <> struct Cplx<double> { using type = double2; };

template <typename T>
struct DeconvDev {
    int nout[3], nspec[3];
    const T* phihat[3];
    const int32_t* index_map[3];
    const int32_t* inv_map[3];
    typename Cplx<T>::type* spec[kMaxCompPerLaunch];
    typename Cplx<T>::type* uni[kMaxCompPerLaunch];
    T normfactor;
    int ncomp;
    const T* mode_factors;     // optional real multiplier per output mode (uniform callback menu), or null
};

template <typename T, int D>
__global__ __launch_bounds__(256) void deconv_truncate_kernel(DeconvDev<T> a) {
    using C2 = typename Cplx<T>::type;
    const int i1 = blockIdx.x * blockDim.x + threadIdx.x;
    const int i2 = D >= 2 ? blockIdx.y : 0;
    const int i3 = D >= 3 ? blockIdx.z : 0;
    if (i1 >= a.nout[0]) return;
    T phi = a.phihat[0][i1];
    int64_t j = a.index_map[0][i1];
    int64_t o = i1;
    if constexpr (D >= 2) {
        phi *= a.phihat[1][i2];
        j += (int64_t)a.index_map[1][i2] * a.nspec[0];
        o += (int64_t)i2 * a.nout[0];
    }
    if constexpr (D >= 3) {
        phi *= a.phihat[2][i3];
        j += (int64_t)a.index_map[2][i3] * a.nspec[0] * a.nspec[1];
        o += (int64_t)i3 * a.nout[0] * a.nout[1];
    }
    T f = a.normfactor / phi;         // β = normfactor / prod(ϕ̂), src/NonuniformFFTs.jl:394
    if (a.mode_factors) f *= a.mode_factors[o];     // callbacks.uniform(ŵ, idx), :395-399
    for (int c = 0; c < a.ncomp; ++c) {
        const C2 u = a.spec[c][j];
        C2 w;
        w.x = f * u.x;
        w.y = f * u.y;
        a.uni[c][o] = w;
    }
}

template <typename T, int D>
__global__ __launch_bounds__(256) void deconv_pad_kernel(DeconvDev<T> a) {
    using C2 = typename Cplx<T>::type;
    const int j1 = blockIdx.x * blockDim.x + threadIdx.x;
    const int j2 = D >= 2 ? blockIdx.y : 0;
    const int j3 = D >= 3 ? blockIdx.z : 0;
    if (j1 >= a.nspec[0]) return;
    int64_t j = j1;
    int i2 = 0, i3 = 0;
    if constexpr (D >= 2) { i2 = a.inv_map[1][j2]; j += (int64_t)j2 * a.nspec[0]; }
    if constexpr (D >= 3) { i3 = a.inv_map[2][j3]; j += (int64_t)j3 * a.nspec[0] * a.nspec[1]; }
    const int i1 = a.inv_map[0][j1];
    C2 zero;
    zero.x = T(0);
    zero.y = T(0);
    if (i1 < 0 || i2 < 0 || i3 < 0) {
        for (int c = 0; c < a.ncomp; ++c) a.spec[c][j] = zero;
        return;
    }
    T phi = a.phihat[0][i1];
    int64_t o = i1;
    if constexpr (D >= 2) { phi *= a.phihat[1][i2]; o += (int64_t)i2 * a.nout[0]; }
    if constexpr (D >= 3) { phi *= a.phihat[2][i3]; o += (int64_t)i3 * a.nout[0] * a.nout[1]; }
    T f = T(1) / phi;                 // β = 1 / prod(ϕ̂), src/NonuniformFFTs.jl:460
    if (a.mode_factors) f *= a.mode_factors[o];     // callbacks.uniform(ŵ, idx), :461-464
    for (int c = 0; c < a.ncomp; ++c) {
        const C2 w = a.uni[c][o];
        C2 u;
        u.x = f * w.x;
        u.y = f * w.y;
        a.spec[c][j] = u;
    }
}

template <typename T>
static hipError_t run_deconv(const DeconvArgs& a, void* const* uni, bool pad, hipStream_t stream) {
    using C2 = typename Cplx<T>::type;
    for (int c0 = 0; c0 < a.C; c0 += kMaxCompPerLaunch) {
        DeconvDev<T> d;
        for (int k = 0; k < 3; ++k) {
            d.nout[k] = a.nout[k];
            d.nspec[k] = a.nspec[k];
            d.phihat[k] = static_cast<const T*>(a.phihat[k]);
            d.index_map[k] = a.index_map[k];
            d.inv_map[k] = a.inv_map[k];
        }
        d.normfactor = (T)a.normfactor;
        d.mode_factors = static_cast<const T*>(a.mode_factors);
        d.ncomp = (a.C - c0) < kMaxCompPerLaunch ? (a.C - c0) : kMaxCompPerLaunch;
        for (int c = 0; c < d.ncomp; ++c) {
            d.spec[c] = static_cast<C2*>(a.spec) + (int64_t)(c0 + c) * a.spec_stride;
            d.uni[c] = static_cast<C2*>(uni[c0 + c]);
        }
        const int* ext = pad ? a.nspec : a.nout;
        dim3 block(256);
        dim3 grid((unsigned)((ext[0] + 255) / 256), (unsigned)(a.D >= 2 ? ext[1] : 1), (unsigned)(a.D >= 3 ? ext[2] : 1));
        if (pad) {
            switch (a.D) {
                case 1: hipLaunchKernelGGL((deconv_pad_kernel<T, 1>), grid, block, 0, stream, d); break;
                case 2: hipLaunchKernelGGL((deconv_pad_kernel<T, 2>), grid, block, 0, stream, d); break;
                default: hipLaunchKernelGGL((deconv_pad_kernel<T, 3>), grid, block, 0, stream, d); break;
            }
        } else {
            switch (a.D) {
                case 1: hipLaunchKernelGGL((deconv_truncate_kernel<T, 1>), grid, block, 0, stream, d); break;
                case 2: hipLaunchKernelGGL((deconv_truncate_kernel<T, 2>), grid, block, 0, stream, d); break;
                default: hipLaunchKernelGGL((deconv_truncate_kernel<T, 3>), grid, block, 0, stream, d); break;
            }
        }
    }
    return hipGetLastError();
}

hipError_t launch_deconv_truncate(const DeconvArgs& a, void* const* uhat_out, hipStream_t stream) {
    return a.dtype == NUFFT_F32 ? run_deconv<float>(a, uhat_out, false, stream)
                                : run_deconv<double>(a, uhat_out, false, stream);
}

hipError_t launch_deconv_pad(const DeconvArgs& a, const void* const* uhat_in, hipStream_t stream) {
    return a.dtype == NUFFT_F32 ? run_deconv<float>(a, const_cast<void* const*>(uhat_in), true, stream)
                                : run_deconv<double>(a, const_cast<void* const*>(uhat_in), true, stream);
}

}  // namespace nufft
