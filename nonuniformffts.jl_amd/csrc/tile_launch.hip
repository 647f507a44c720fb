// Host-side dispatch of the tile kernels over (precision, complex?, D, M).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "tile_kernels.h"

namespace nufft {

const void* spread_kernel_f32r(int D, int M, bool flag, bool other);
const void* spread_kernel_f32c(int D, int M, bool flag, bool other);
const void* spread_kernel_f64r(int D, int M, bool flag, bool other);
const void* spread_kernel_f64c(int D, int M, bool flag, bool other);
const void* interp_kernel_f32r(int D, int M, bool flag, bool other);
const void* interp_kernel_f32c(int D, int M, bool flag, bool other);
const void* interp_kernel_f64r(int D, int M, bool flag, bool other);
const void* interp_kernel_f64c(int D, int M, bool flag, bool other);

const void* spread_fixed_f32r(int D, int M, int* n);
const void* spread_fixed_f32c(int D, int M, int* n);
const void* spread_fixed_f64r(int D, int M, int* n);
const void* spread_fixed_f64c(int D, int M, int* n);
void interp_fixed_dims_f32r(int D, int M, int* n);
void interp_fixed_dims_f32c(int D, int M, int* n);
void interp_fixed_dims_f64r(int D, int M, int* n);
void interp_fixed_dims_f64c(int D, int M, int* n);

void interp_fixed_dims(int dtype, int is_complex, int D, int M, int* n) {
    n[0] = n[1] = n[2] = n[3] = 0;
    if (M < 2 || M > 10 || D < 1 || D > 3) return;
    if (dtype == NUFFT_F32) is_complex ? interp_fixed_dims_f32c(D, M, n) : interp_fixed_dims_f32r(D, M, n);
    else is_complex ? interp_fixed_dims_f64c(D, M, n) : interp_fixed_dims_f64r(D, M, n);
}

// Kernel with the compile-time spreading tile (null: none) and the tile itself (n[0..2], n[3] = row stride).
const void* spread_fixed_kernel(int dtype, int is_complex, int D, int M, int* n) {
    n[0] = n[1] = n[2] = n[3] = 0;
    if (M < 2 || M > 10 || D < 1 || D > 3) return nullptr;
    if (dtype == NUFFT_F32) return is_complex ? spread_fixed_f32c(D, M, n) : spread_fixed_f32r(D, M, n);
    return is_complex ? spread_fixed_f64c(D, M, n) : spread_fixed_f64r(D, M, n);
}
void spread_fixed_dims(int dtype, int is_complex, int D, int M, int* n) { (void)spread_fixed_kernel(dtype, is_complex, D, M, n); }

// `flag`: spreading = single-tile axis (wrap variant); interpolation = compile-time tile.
// `other`: window evaluation of the non-default kernels (see needs_other_eval).
static const void* pick(bool interp, int dtype, int is_complex, int D, int M, bool flag, bool other) {
    if (interp) {
        if (dtype == NUFFT_F32) return is_complex ? interp_kernel_f32c(D, M, flag, other) : interp_kernel_f32r(D, M, flag, other);
        return is_complex ? interp_kernel_f64c(D, M, flag, other) : interp_kernel_f64r(D, M, flag, other);
    }
    if (dtype == NUFFT_F32) return is_complex ? spread_kernel_f32c(D, M, flag, other) : spread_kernel_f32r(D, M, flag, other);
    return is_complex ? spread_kernel_f64c(D, M, flag, other) : spread_kernel_f64r(D, M, flag, other);
}

// The polynomial evaluation (FastApproximation of both Kaiser-Bessel kernels) and the sinh form of the
// backwards Kaiser-Bessel kernel live in the default instantiations; everything else in the OTHERK ones.
bool needs_other_eval(int kernel, int evalmode) {
    if (kernel == NUFFT_KERNEL_BACKWARDS_KAISER_BESSEL) return false;
    if (kernel == NUFFT_KERNEL_KAISER_BESSEL) return evalmode == NUFFT_EVAL_DIRECT;
    return true;
}

static hipError_t prepare(bool interp, int dtype, int is_complex, int D, int M, int lds_bytes, bool other) {
    for (int wrap = 0; wrap < 2; ++wrap) {
        const void* fn = pick(interp, dtype, is_complex, D, M, wrap != 0, other);
        if (!fn && interp && wrap) continue;       // no compile-time tile for this instantiation
        if (!fn) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (!interp && !other) {
        int n[4];
        const void* fn = spread_fixed_kernel(dtype, is_complex, D, M, n);
        if (fn) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

hipError_t prepare_spread(int dtype, int is_complex, int D, int M, int lds_bytes, bool other) {
    return prepare(false, dtype, is_complex, D, M, lds_bytes, other);
}
hipError_t prepare_interp(int dtype, int is_complex, int D, int M, int lds_bytes, bool other) {
    return prepare(true, dtype, is_complex, D, M, lds_bytes, other);
}

template <typename T>
static hipError_t launch_t(bool interp, const TileKernelArgs& a, hipStream_t stream) {
    bool wrap = false;
    for (int d = 0; d < a.D; ++d) wrap = wrap || a.g.sp.nt[d] == 1;
    const bool other = needs_other_eval(a.kernel, a.evalmode) || a.weights != nullptr;   // general variant
    const void* fn = pick(interp, a.dtype, a.is_complex, a.D, a.M, interp ? (a.fixed_tile != 0 && !other) : wrap, other);
    if (!interp && a.fixed_tile != 0 && !other && !wrap) {
        int n[4];
        const void* ff = spread_fixed_kernel(a.dtype, a.is_complex, a.D, a.M, n);
        if (ff) fn = ff;
    }
    if (!fn) return hipErrorInvalidValue;
    const int ncr = a.is_complex ? 2 : 1;
    for (int c0 = 0; c0 < a.C; c0 += kMaxCompPerLaunch) {
        const int nc = (a.C - c0) < kMaxCompPerLaunch ? (a.C - c0) : kMaxCompPerLaunch;
        TileArgs<T> k{};
        k.g = a.g;
        k.sorted = a.sorted;
        k.offsets = a.offsets;
        k.coefs = static_cast<const T*>(a.coefs);
        for (int d = 0; d < 3; ++d) {
            k.beta[d] = (T)a.beta[d];
            k.bop[d] = (T)a.beta_over_pi[d];
        }
        for (int c = 0; c < nc; ++c) {
            k.grid[c] = static_cast<T*>(a.grid) + (int64_t)(c0 + c) * a.grid_stride * ncr;
            k.vin[c] = a.values_in ? static_cast<const T*>(a.values_in[c0 + c]) : nullptr;
            k.vout[c] = a.values_out ? static_cast<T*>(a.values_out[c0 + c]) : nullptr;
        }
        k.prefactor = (T)a.prefactor;
        k.weights = static_cast<const T*>(a.weights);
        k.desc = static_cast<const uint2*>(a.desc);
        k.desc_total = a.desc_total;
        k.xcd_chunk = a.xcd_chunk;
        k.evalmode = a.evalmode;
        k.kernel = a.kernel;
        void* params[] = {&k};
        hipError_t e = hipLaunchKernel(fn, dim3((unsigned)a.ntiles, (unsigned)nc, 1), dim3((unsigned)a.threads, 1, 1),
                                       params, (size_t)a.lds_bytes, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_spread(const TileKernelArgs& a, hipStream_t stream) {
    return a.dtype == NUFFT_F32 ? launch_t<float>(false, a, stream) : launch_t<double>(false, a, stream);
}
hipError_t launch_interp(const TileKernelArgs& a, hipStream_t stream) {
    return a.dtype == NUFFT_F32 ? launch_t<float>(true, a, stream) : launch_t<double>(true, a, stream);
}

}  // namespace nufft
