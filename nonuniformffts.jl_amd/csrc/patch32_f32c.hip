// spread_patch32_kernel instantiations (ComplexF32 on the FP32 matrix pipe): one per half-support M.
#include "patch32_kernels.h"

namespace nufft {

template <int M>
static void patch32_entry(const void** fn, int* lds_bytes, int* pby) {
    using P = Patch32Cfg<M>;
    *fn = reinterpret_cast<const void*>(&spread_patch32_kernel<M, false>);
    *lds_bytes = P::lds_bytes();
    *pby = P::PBY;
}

// kernel for half-support M (null: none), its dynamic LDS bytes and the rows of cube columns of its patch
const void* patch32_kernel_f32c(int M, bool other, int* lds_bytes, int* pby) {
    const void* fn = nullptr;
    *lds_bytes = 0;
    *pby = 0;
    if (other) return nullptr;      // the remaining window kernels and per-point weights use the LDS-tile kernel
    switch (M) {
#if defined(NUFFT_PATCH32_ONLY_M)       // development builds: one instantiation (compile time)
        case NUFFT_PATCH32_ONLY_M: patch32_entry<NUFFT_PATCH32_ONLY_M>(&fn, lds_bytes, pby); break;
#else
        case 2: patch32_entry<2>(&fn, lds_bytes, pby); break;
        case 3: patch32_entry<3>(&fn, lds_bytes, pby); break;
        case 4: patch32_entry<4>(&fn, lds_bytes, pby); break;
        case 5: patch32_entry<5>(&fn, lds_bytes, pby); break;
        case 6: patch32_entry<6>(&fn, lds_bytes, pby); break;
        case 7: patch32_entry<7>(&fn, lds_bytes, pby); break;
        case 8: patch32_entry<8>(&fn, lds_bytes, pby); break;
        case 9: patch32_entry<9>(&fn, lds_bytes, pby); break;
        case 10: patch32_entry<10>(&fn, lds_bytes, pby); break;
#endif
        default: break;
    }
    return fn;
}

}  // namespace nufft
