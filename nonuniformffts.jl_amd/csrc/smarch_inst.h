// Instantiation + dispatch of spread_march_kernel for one (real type, complex?) pair.  Included by smarch_*.hip after
// defining NUFFT_T, NUFFT_CPLX and NUFFT_SMARCH_GETTER (name of the exported getter).
#include "smarch_kernels.h"

namespace nufft {

template <int M>
static void smarch_entry(bool poly, const void** fn, int* lds_bytes, int* n) {
    using C = SMarchCfg<NUFFT_T, NUFFT_CPLX, M>;
    if constexpr (C::FITS) {
        *fn = poly ? reinterpret_cast<const void*>(&spread_march_kernel<NUFFT_T, NUFFT_CPLX, M, true>)
                   : reinterpret_cast<const void*>(&spread_march_kernel<NUFFT_T, NUFFT_CPLX, M, false>);
        *lds_bytes = C::lds_bytes();
        n[0] = C::N1; n[1] = C::N2; n[2] = C::HLO; n[3] = C::HHI; n[4] = C::THREADS;
    }
}

// kernel for half-support M and window evaluation (polynomial / direct; null: none), its dynamic LDS bytes, the column interior n[0] x n[1], the layers of points a
// segment visits below / above its own n[2], n[3], and the workgroup size n[4]
const void* NUFFT_SMARCH_GETTER(int M, bool poly, int* lds_bytes, int* n) {
    const void* fn = nullptr;
    *lds_bytes = 0;
    n[0] = n[1] = n[2] = n[3] = n[4] = 0;
    switch (M) {
        case 2: smarch_entry<2>(poly, &fn, lds_bytes, n); break;
        case 3: smarch_entry<3>(poly, &fn, lds_bytes, n); break;
        case 4: smarch_entry<4>(poly, &fn, lds_bytes, n); break;
        case 5: smarch_entry<5>(poly, &fn, lds_bytes, n); break;
        case 6: smarch_entry<6>(poly, &fn, lds_bytes, n); break;
        case 7: smarch_entry<7>(poly, &fn, lds_bytes, n); break;
        case 8: smarch_entry<8>(poly, &fn, lds_bytes, n); break;
        case 9: smarch_entry<9>(poly, &fn, lds_bytes, n); break;
        case 10: smarch_entry<10>(poly, &fn, lds_bytes, n); break;
        default: break;
    }
    return fn;
}

}  // namespace nufft
