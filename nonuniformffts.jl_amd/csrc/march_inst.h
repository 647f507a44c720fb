// Instantiation + dispatch of interp_march_kernel for one (real type, complex?) pair.  Included by march_*.hip after
// defining NUFFT_T, NUFFT_CPLX and NUFFT_MARCH_GETTER (name of the exported getter).
#include "march_kernels.h"

namespace nufft {

template <int M>
static void march_entry(const void** fn, int* lds_bytes, int* n) {
    using C = MarchCfg<NUFFT_T, NUFFT_CPLX, M>;
    if constexpr (C::FITS) {
        *fn = reinterpret_cast<const void*>(&interp_march_kernel<NUFFT_T, NUFFT_CPLX, M>);
        *lds_bytes = C::lds_bytes();
        n[0] = C::N1; n[1] = C::N2; n[2] = C::kSegMax; n[3] = C::THREADS;
    }
}

// kernel for half-support M (null: none), its dynamic LDS bytes, the column interior n[0] x n[1], the longest segment n[2] and the workgroup size n[3]
const void* NUFFT_MARCH_GETTER(int M, int* lds_bytes, int* n) {
    const void* fn = nullptr;
    *lds_bytes = 0;
    n[0] = n[1] = n[2] = n[3] = 0;
    switch (M) {
        case 2: march_entry<2>(&fn, lds_bytes, n); break;
        case 3: march_entry<3>(&fn, lds_bytes, n); break;
        case 4: march_entry<4>(&fn, lds_bytes, n); break;
        case 5: march_entry<5>(&fn, lds_bytes, n); break;
        case 6: march_entry<6>(&fn, lds_bytes, n); break;
        case 7: march_entry<7>(&fn, lds_bytes, n); break;
        case 8: march_entry<8>(&fn, lds_bytes, n); break;
        case 9: march_entry<9>(&fn, lds_bytes, n); break;
        case 10: march_entry<10>(&fn, lds_bytes, n); break;
        default: break;
    }
    return fn;
}

}  // namespace nufft
