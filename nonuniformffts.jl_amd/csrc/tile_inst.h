// Instantiation + dispatch of the tile kernels for one (real type, complex?) pair.
// Included by spread_*.hip / interp_*.hip after defining NUFFT_T, NUFFT_CPLX, NUFFT_KERNEL
// (spread_tile_kernel | interp_tile_kernel) and NUFFT_GETTER (name of the exported getter).
#include "tile_kernels.h"

namespace nufft {

using TileKernelPtr = void (*)(TileArgs<NUFFT_T>);

template <int D>
static TileKernelPtr pick_m(int M) {
    switch (M) {
        case 2: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 2>;
        case 3: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 3>;
        case 4: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 4>;
        case 5: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 5>;
        case 6: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 6>;
        case 7: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 7>;
        case 8: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 8>;
        case 9: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 9>;
        case 10: return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, 10>;
        default: return nullptr;
    }
}

const void* NUFFT_GETTER(int D, int M) {
    switch (D) {
        case 1: return reinterpret_cast<const void*>(pick_m<1>(M));
        case 2: return reinterpret_cast<const void*>(pick_m<2>(M));
        case 3: return reinterpret_cast<const void*>(pick_m<3>(M));
        default: return nullptr;
    }
}

}  // namespace nufft
