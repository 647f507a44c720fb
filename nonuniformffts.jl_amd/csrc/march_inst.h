// Instantiation + dispatch of interp_march_kernel for one (real type, complex?) pair.  Included by march_*.hip after
// defining NUFFT_T, NUFFT_CPLX and NUFFT_MARCH_GETTER (name of the exported getter).
#include "march_kernels.h"

namespace nufft {

template <int M, bool POLY>
static void march_entry_p(const void** fn, int* lds_bytes, int* n) {
    using C = MarchCfg<NUFFT_T, NUFFT_CPLX, M, POLY>;
    if constexpr (C::FITS) {
        *fn = reinterpret_cast<const void*>(&interp_march_kernel<NUFFT_T, NUFFT_CPLX, M, POLY>);
        *lds_bytes = C::lds_bytes();
        n[0] = C::N1; n[1] = C::N2; n[2] = C::kSegMax; n[3] = C::THREADS;
    }
}
template <int M>
static void march_entry(bool poly, const void** fn, int* lds_bytes, int* n) {
    if (poly) march_entry_p<M, true>(fn, lds_bytes, n);
    else march_entry_p<M, false>(fn, lds_bytes, n);
}

// staged variant (column-layer sorted point sets): every element type, the half-supports the spreading window's halo variant serves.
// Its compile-time column (the largest that fits beside the stage) bounds the column the plan may give both rings (plan.cpp).
template <int M, bool POLY>
static void march_staged_entry_p(const void** fn, int* lds_bytes, int* n) {
    using C = MarchCfg<NUFFT_T, NUFFT_CPLX, M, POLY, true>;
    if constexpr (C::FITS_STAGED && M <= 7) {
        *fn = reinterpret_cast<const void*>(&interp_march_staged_kernel<NUFFT_T, NUFFT_CPLX, M, POLY>);
        *lds_bytes = C::staged_lds_bytes();
        n[0] = C::N1; n[1] = C::N2; n[2] = C::kSegMax; n[3] = C::THREADS;
    }
}
template <int M>
static void march_staged_entry(bool poly, const void** fn, int* lds_bytes, int* n) {
    if (poly) march_staged_entry_p<M, true>(fn, lds_bytes, n);
    else march_staged_entry_p<M, false>(fn, lds_bytes, n);
}

// kernel for half-support M and window evaluation (polynomial / direct; null: none), its dynamic LDS bytes, the column interior n[0] x n[1], the longest segment n[2] and the workgroup size n[3]
const void* NUFFT_MARCH_GETTER(int M, bool poly, int* lds_bytes, int* n) {
    const void* fn = nullptr;
    *lds_bytes = 0;
    n[0] = n[1] = n[2] = n[3] = 0;
    switch (M) {
        case 2: march_entry<2>(poly, &fn, lds_bytes, n); break;
        case 3: march_entry<3>(poly, &fn, lds_bytes, n); break;
        case 4: march_entry<4>(poly, &fn, lds_bytes, n); break;
        case 5: march_entry<5>(poly, &fn, lds_bytes, n); break;
        case 6: march_entry<6>(poly, &fn, lds_bytes, n); break;
        case 7: march_entry<7>(poly, &fn, lds_bytes, n); break;
        case 8: march_entry<8>(poly, &fn, lds_bytes, n); break;
        case 9: march_entry<9>(poly, &fn, lds_bytes, n); break;
        case 10: march_entry<10>(poly, &fn, lds_bytes, n); break;
        default: break;
    }
    return fn;
}

// the staged variant of the same kernel (null: none for this configuration)
const void* NUFFT_MARCH_GETTER_STAGED(int M, bool poly, int* lds_bytes, int* n) {
    const void* fn = nullptr;
    *lds_bytes = 0;
    n[0] = n[1] = n[2] = n[3] = 0;
    switch (M) {
        case 2: march_staged_entry<2>(poly, &fn, lds_bytes, n); break;
        case 3: march_staged_entry<3>(poly, &fn, lds_bytes, n); break;
        case 4: march_staged_entry<4>(poly, &fn, lds_bytes, n); break;
        case 5: march_staged_entry<5>(poly, &fn, lds_bytes, n); break;
        case 6: march_staged_entry<6>(poly, &fn, lds_bytes, n); break;
        case 7: march_staged_entry<7>(poly, &fn, lds_bytes, n); break;
        default: break;
    }
    return fn;
}

}  // namespace nufft
