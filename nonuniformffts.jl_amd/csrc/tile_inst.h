// Instantiation + dispatch of the tile kernels for one (real type, complex?) pair.
// Included by spread_*.hip / interp_*.hip after defining NUFFT_T, NUFFT_CPLX, NUFFT_KERNEL
// (spread_tile_kernel | interp_tile_kernel) and NUFFT_GETTER (name of the exported getter).
#include "tile_kernels.h"

namespace nufft {

using TileKernelPtr = void (*)(TileArgs<NUFFT_T>);

template <int D, bool WRAP>
static TileKernelPtr pick_m(int M) {
    switch (M) {
        case 2: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 2, WRAP>;
        case 3: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 3, WRAP>;
        case 4: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 4, WRAP>;
        case 5: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 5, WRAP>;
        case 6: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 6, WRAP>;
        case 7: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 7, WRAP>;
        case 8: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 8, WRAP>;
        case 9: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 9, WRAP>;
        case 10: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 10, WRAP>;
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

}  // namespace nufft
