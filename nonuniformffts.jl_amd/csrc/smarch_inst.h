// Instantiation + dispatch of spread_march_kernel for one (real type, complex?) pair.  Included by smarch_*.hip after
// defining NUFFT_T, NUFFT_CPLX and NUFFT_SMARCH_GETTER (name of the exported getter).
#include "smarch_kernels.h"

#include <algorithm>

namespace nufft {

template <int M, bool HX, bool HY>
static void smarch_entry_h(bool poly, const void** fn, int* lds_bytes, int* n) {
    using C = SMarchCfg<NUFFT_T, NUFFT_CPLX, M, HX, HY>;
    if constexpr (C::FITS) {
        *fn = poly ? reinterpret_cast<const void*>(&spread_march_kernel<NUFFT_T, NUFFT_CPLX, M, true, HX, HY>)
                   : reinterpret_cast<const void*>(&spread_march_kernel<NUFFT_T, NUFFT_CPLX, M, false, HX, HY>);
        *lds_bytes = C::lds_bytes();
        n[0] = C::N1; n[1] = C::N2; n[2] = C::HLO; n[3] = C::HHI; n[4] = C::THREADS;
    }
}
// halo: 0 = output-driven in x and y (clipped: the column visits every point whose stencil reaches it); 2 = the halo variant
// (every point spread once by its own column, the reach into a side buffer: smarch_kernels.h).
template <int M>
static void smarch_entry(int halo, bool poly, const void** fn, int* lds_bytes, int* n) {
    if (halo == 2) { smarch_entry_h<M, true, true>(poly, fn, lds_bytes, n); return; }
    if (halo == 0) smarch_entry_h<M, false, false>(poly, fn, lds_bytes, n);
}

// kernel for half-support M, halo variant and window evaluation (polynomial / direct; null: none), its dynamic LDS bytes, the column interior n[0] x n[1], the layers of points a
// segment visits below / above its own n[2], n[3], and the workgroup size n[4]
const void* NUFFT_SMARCH_GETTER(int M, int halo, bool poly, int* lds_bytes, int* n) {
    const void* fn = nullptr;
    *lds_bytes = 0;
    n[0] = n[1] = n[2] = n[3] = n[4] = 0;
    switch (M) {
        case 2: smarch_entry<2>(halo, poly, &fn, lds_bytes, n); break;
        case 3: smarch_entry<3>(halo, poly, &fn, lds_bytes, n); break;
        case 4: smarch_entry<4>(halo, poly, &fn, lds_bytes, n); break;
        case 5: smarch_entry<5>(halo, poly, &fn, lds_bytes, n); break;
        case 6: smarch_entry<6>(halo, poly, &fn, lds_bytes, n); break;
        case 7: smarch_entry<7>(halo, poly, &fn, lds_bytes, n); break;
        case 8: smarch_entry<8>(halo, poly, &fn, lds_bytes, n); break;
        case 9: smarch_entry<9>(halo, poly, &fn, lds_bytes, n); break;
        case 10: smarch_entry<10>(halo, poly, &fn, lds_bytes, n); break;
        default: break;
    }
    return fn;
}

}  // namespace nufft

namespace nufft {
// grid += side buffer of the halo variant (instantiated once per real type: with the real units)
#if !NUFFT_CPLX_IS_TRUE
hipError_t NUFFT_SMARCH_HALO_ADD(void* grid, const void* halo, int64_t grid_comp_reals, int64_t halo_comp_reals, const Geom& g, int nc, int C,
                                 int n1, int n2, int M, int ntx, int nty, const uint32_t* flag, hipStream_t stream) {
    const int64_t rows = (int64_t)g.Nover[1] * g.Nover[2];
    const unsigned blocks = (unsigned)std::min<int64_t>(rows, 65535 * 4);
    const HaloLayout h = make_halo_layout(n1, n2, M, nc, ntx, nty);
    hipLaunchKernelGGL(smarch_halo_add_kernel<NUFFT_T>, dim3(blocks, (unsigned)C, 1), dim3(256), 0, stream, static_cast<NUFFT_T*>(grid),
                       static_cast<const NUFFT_T*>(halo), grid_comp_reals, halo_comp_reals, g, h, flag);
    return hipGetLastError();
}
#endif
}  // namespace nufft
