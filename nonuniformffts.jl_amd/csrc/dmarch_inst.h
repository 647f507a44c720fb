// Instantiation + dispatch of spread_march_dense_kernel for one real type.  Included by dmarch_*.hip after defining NUFFT_T and
// NUFFT_DMARCH_GETTER (name of the exported getter).
#include "dmarch_kernels.h"

namespace nufft {

template <int M>
static void dmarch_entry(bool poly, const void** fn, int* lds_bytes, int* n) {
    using C = DMarchCfg<NUFFT_T, M>;
    if constexpr (C::FITS) {
        *fn = poly ? reinterpret_cast<const void*>(&spread_march_dense_kernel<NUFFT_T, M, true>)
                   : reinterpret_cast<const void*>(&spread_march_dense_kernel<NUFFT_T, M, false>);
        *lds_bytes = C::lds_bytes();
        n[0] = C::N1; n[1] = C::N2; n[2] = C::THREADS; n[3] = C::NT;
    }
}

// kernel for half-support M and window evaluation (null: none), its dynamic LDS bytes, the largest column n[0] x n[1] (the halo variant's of
// spread_march_kernel), the workgroup size n[2] and the matrix instructions per batch of four points n[3]
const void* NUFFT_DMARCH_GETTER(int M, bool poly, int* lds_bytes, int* n) {
    const void* fn = nullptr;
    *lds_bytes = 0;
    n[0] = n[1] = n[2] = n[3] = 0;
    switch (M) {
        case 2: dmarch_entry<2>(poly, &fn, lds_bytes, n); break;
        case 3: dmarch_entry<3>(poly, &fn, lds_bytes, n); break;
        case 4: dmarch_entry<4>(poly, &fn, lds_bytes, n); break;
        case 5: dmarch_entry<5>(poly, &fn, lds_bytes, n); break;
        case 6: dmarch_entry<6>(poly, &fn, lds_bytes, n); break;
        default: break;
    }
    return fn;
}

}  // namespace nufft
