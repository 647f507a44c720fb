// Instantiation + dispatch of spread_patch_kernel with PLANAR components (the ntransforms = 2 / 3 real value vectors of a
// real plan spread together, sharing window evaluation and operand set-up) for one real type.  Included by
// patch_*_p.hip after defining NUFFT_T and NUFFT_PATCH_PLANAR_GETTER.
#include "patch_kernels.h"

namespace nufft {

template <int NP, int M>
static void patch_planar_entry(const void** fn, int* lds_bytes, int* pby) {
    using P = PatchCfg<NP, M, true>;
    *fn = reinterpret_cast<const void*>(&spread_patch_kernel<NUFFT_T, false, M, false, NP>);
    *lds_bytes = P::lds_bytes((int)sizeof(NUFFT_T), kPatchWaves);
    *pby = P::PBY;
}

template <int NP>
static const void* patch_planar_m(int M, int* lds_bytes, int* pby) {
    const void* fn = nullptr;
    switch (M) {
        case 2: patch_planar_entry<NP, 2>(&fn, lds_bytes, pby); break;
        case 3: patch_planar_entry<NP, 3>(&fn, lds_bytes, pby); break;
        case 4: patch_planar_entry<NP, 4>(&fn, lds_bytes, pby); break;
        case 5: patch_planar_entry<NP, 5>(&fn, lds_bytes, pby); break;
        case 6: patch_planar_entry<NP, 6>(&fn, lds_bytes, pby); break;
        default: break;           // wider stencils: one component after the other
    }
    return fn;
}

// kernel for NP = 2 / 3 planar components and half-support M (null: none), its dynamic LDS bytes and patch rows
const void* NUFFT_PATCH_PLANAR_GETTER(int NP, int M, int* lds_bytes, int* pby) {
    *lds_bytes = 0;
    *pby = 0;
    if (NP == 2) return patch_planar_m<2>(M, lds_bytes, pby);
    if (NP == 3) return patch_planar_m<3>(M, lds_bytes, pby);
    return nullptr;
}

}  // namespace nufft
