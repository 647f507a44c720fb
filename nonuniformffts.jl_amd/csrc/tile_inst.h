// Instantiation + dispatch of the tile kernels for one (real type, complex?) pair.
// Included by spread_*.hip / interp_*.hip after defining NUFFT_T, NUFFT_CPLX, NUFFT_KERNEL
// (spread_tile_kernel | interp_tile_kernel) and NUFFT_GETTER (name of the exported getter).
#include "tile_kernels.h"

namespace nufft {

using TileKernelPtr = void (*)(TileArgs<NUFFT_T>);

#if defined(NUFFT_FIXED_DIMS_GETTER)
// The 5th template flag of the interpolation kernel selects the compile-time tile; instantiate it only
// where such a tile exists (fixed_interp_tile().n[0] > 0).
template <int D, int M, bool FLAG>
static TileKernelPtr inst() {
    constexpr bool ok = !FLAG || fixed_interp_tile((int)sizeof(NUFFT_T), NUFFT_CPLX ? 2 : 1, D, M).n[0] > 0;
    if constexpr (ok) return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, M, FLAG>;
    else return nullptr;
}
#else
template <int D, int M, bool FLAG>
static TileKernelPtr inst() { return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, M, FLAG>; }
#endif

template <int D, bool WRAP>
static TileKernelPtr pick_m(int M) {
    switch (M) {
        case 2: return inst<D, 2, WRAP>();
        case 3: return inst<D, 3, WRAP>();
        case 4: return inst<D, 4, WRAP>();
        case 5: return inst<D, 5, WRAP>();
        case 6: return inst<D, 6, WRAP>();
        case 7: return inst<D, 7, WRAP>();
        case 8: return inst<D, 8, WRAP>();
        case 9: return inst<D, 9, WRAP>();
        case 10: return inst<D, 10, WRAP>();
        default: return nullptr;
    }
}

const void* NUFFT_GETTER(int D, int M, bool wrap) {
#if NUFFT_HAS_WRAP_VARIANT
    if (wrap) {
        switch (D) {
            case 1: return reinterpret_cast<const void*>(pick_m<1, true>(M));
            case 2: return reinterpret_cast<const void*>(pick_m<2, true>(M));
            case 3: return reinterpret_cast<const void*>(pick_m<3, true>(M));
            default: return nullptr;
        }
    }
#endif
    (void)wrap;
    switch (D) {
        case 1: return reinterpret_cast<const void*>(pick_m<1, false>(M));
        case 2: return reinterpret_cast<const void*>(pick_m<2, false>(M));
        case 3: return reinterpret_cast<const void*>(pick_m<3, false>(M));
        default: return nullptr;
    }
}

#if defined(NUFFT_FIXED_DIMS_GETTER)
void NUFFT_FIXED_DIMS_GETTER(int D, int M, int* n) {
    const FixedTileDims fd = fixed_interp_tile((int)sizeof(NUFFT_T), NUFFT_CPLX ? 2 : 1, D, M);
    for (int d = 0; d < 3; ++d) n[d] = fd.n[d];
    n[3] = fd.row_stride;
}
#endif

}  // namespace nufft
