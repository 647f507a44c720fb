// Instantiation + dispatch of spread_patch_kernel for one (real type, complex?) pair.  Included by patch_*.hip after
// defining NUFFT_T, NUFFT_CPLX and NUFFT_PATCH_GETTER (name of the exported getter).
#include "patch_kernels.h"

namespace nufft {

template <int M, bool OTHER>
static void patch_entry(const void** fn, int* lds_bytes, int* pby) {
    using P = PatchCfg<NUFFT_CPLX ? 2 : 1, M>;
    *fn = reinterpret_cast<const void*>(&spread_patch_kernel<NUFFT_T, NUFFT_CPLX, M, OTHER>);
    *lds_bytes = P::lds_bytes((int)sizeof(NUFFT_T), kPatchWaves);
    *pby = P::PBY;
}

// kernel for half-support M (null: none), its dynamic LDS bytes and the rows of cube columns of its patch
const void* NUFFT_PATCH_GETTER(int M, bool other, int* lds_bytes, int* pby) {
    const void* fn = nullptr;
    *lds_bytes = 0;
    *pby = 0;
    if (other) return nullptr;      // the remaining window kernels and per-point weights use the LDS-tile kernel
    switch (M) {
        case 2: patch_entry<2, false>(&fn, lds_bytes, pby); break;
        case 3: patch_entry<3, false>(&fn, lds_bytes, pby); break;
        case 4: patch_entry<4, false>(&fn, lds_bytes, pby); break;
        case 5: patch_entry<5, false>(&fn, lds_bytes, pby); break;
        case 6: patch_entry<6, false>(&fn, lds_bytes, pby); break;
        case 7: patch_entry<7, false>(&fn, lds_bytes, pby); break;
        case 8: patch_entry<8, false>(&fn, lds_bytes, pby); break;
        case 9: patch_entry<9, false>(&fn, lds_bytes, pby); break;
        case 10: patch_entry<10, false>(&fn, lds_bytes, pby); break;
        default: break;
    }
    return fn;
}

}  // namespace nufft
