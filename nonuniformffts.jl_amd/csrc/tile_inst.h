// Instantiation + dispatch of the tile kernels for one (real type, complex?) pair.
// Included by spread_*.hip / interp_*.hip after defining NUFFT_T, NUFFT_CPLX, NUFFT_KERNEL
// (spread_tile_kernel | interp_tile_kernel) and NUFFT_GETTER (name of the exported getter).
// Template flags: FLAG = WRAP (spreading: a single tile spans an axis) or FIXED (interpolation:
// compile-time tile); OTHER = window evaluation of the non-default kernels (WindowEval<.., OTHERK>).
#include "tile_kernels.h"

namespace nufft {

using TileKernelPtr = void (*)(TileArgs<NUFFT_T>);

template <int D, int M, bool FLAG, bool OTHER>
static TileKernelPtr inst() {
#if defined(NUFFT_FIXED_DIMS_GETTER)
    // interpolation: the compile-time tile exists only where something fits, and only for the default
    // window evaluation
    constexpr bool ok = !FLAG || (!OTHER && fixed_interp_tile((int)sizeof(NUFFT_T), NUFFT_CPLX ? 2 : 1, D, M).n[0] > 0);
#else
    constexpr bool ok = true;
#endif
    if constexpr (ok) return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, M, FLAG, OTHER>;
    else return nullptr;
}

#if defined(NUFFT_SPREAD_FIXED_GETTER)
// spreading with the compile-time tile (default window evaluation, no wrap), where such a tile exists
template <int D, int M>
static TileKernelPtr inst_fixed() {
    constexpr bool ok = fixed_spread_tile((int)sizeof(NUFFT_T), NUFFT_CPLX ? 2 : 1, D, M).n[0] > 0;
    if constexpr (ok) return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, M, false, false, true>;
    else return nullptr;
}
// ... and its cube-accumulation variant (real data, 3-D, M <= 4: stencils of three cubes per dimension)
template <int D, int M>
static TileKernelPtr inst_cubes() {
    constexpr FixedTileDims fd = fixed_spread_tile((int)sizeof(NUFFT_T), NUFFT_CPLX ? 2 : 1, D, M);
    constexpr bool ok = !NUFFT_CPLX && D == 3 && M == 4 && fd.n[0] > 0 && fd.n[0] % 4 == 0 && fd.n[1] % 4 == 0 && fd.n[2] % 4 == 0 &&
                        (kWave / lanes_per_point(1, M)) % 4 == 0 &&
                        (8 * fd.plane_stride + 8 * fd.row_stride + fd.n[0]) * 8 < 65536;
    if constexpr (ok) return NUFFT_KERNEL<NUFFT_T, NUFFT_CPLX, D, M, false, false, true, true>;
    else return nullptr;
}
template <int D>
static TileKernelPtr pick_m_cubes(int M) {
    switch (M) {
        case 2: return inst_cubes<D, 2>();  case 3: return inst_cubes<D, 3>();  case 4: return inst_cubes<D, 4>();
        default: return nullptr;
    }
}

template <int D>
static TileKernelPtr pick_m_fixed(int M) {
    switch (M) {
        case 2: return inst_fixed<D, 2>();  case 3: return inst_fixed<D, 3>();  case 4: return inst_fixed<D, 4>();
        case 5: return inst_fixed<D, 5>();  case 6: return inst_fixed<D, 6>();  case 7: return inst_fixed<D, 7>();
        case 8: return inst_fixed<D, 8>();  case 9: return inst_fixed<D, 9>();  case 10: return inst_fixed<D, 10>();
        default: return nullptr;
    }
}
#endif

template <int D, bool FLAG, bool OTHER>
static TileKernelPtr pick_m(int M) {
    switch (M) {
        case 2: return inst<D, 2, FLAG, OTHER>();
        case 3: return inst<D, 3, FLAG, OTHER>();
        case 4: return inst<D, 4, FLAG, OTHER>();
        case 5: return inst<D, 5, FLAG, OTHER>();
        case 6: return inst<D, 6, FLAG, OTHER>();
        case 7: return inst<D, 7, FLAG, OTHER>();
        case 8: return inst<D, 8, FLAG, OTHER>();
        case 9: return inst<D, 9, FLAG, OTHER>();
        case 10: return inst<D, 10, FLAG, OTHER>();
        default: return nullptr;
    }
}

template <bool FLAG, bool OTHER>
static const void* pick_d(int D, int M) {
    switch (D) {
        case 1: return reinterpret_cast<const void*>(pick_m<1, FLAG, OTHER>(M));
        case 2: return reinterpret_cast<const void*>(pick_m<2, FLAG, OTHER>(M));
        case 3: return reinterpret_cast<const void*>(pick_m<3, FLAG, OTHER>(M));
        default: return nullptr;
    }
}

const void* NUFFT_GETTER(int D, int M, bool flag, bool other) {
    if (flag) return other ? pick_d<true, true>(D, M) : pick_d<true, false>(D, M);
    return other ? pick_d<false, true>(D, M) : pick_d<false, false>(D, M);
}

#if defined(NUFFT_SPREAD_FIXED_GETTER)
// kernel with the compile-time spreading tile (or null) and that tile: n[0..2] cells, n[3] = LDS row stride
const void* NUFFT_SPREAD_FIXED_GETTER(int D, int M, int* n) {
    const FixedTileDims fd = fixed_spread_tile((int)sizeof(NUFFT_T), NUFFT_CPLX ? 2 : 1, D, M);
    for (int d = 0; d < 3; ++d) n[d] = fd.n[d];
    n[3] = fd.row_stride;
    n[4] = fd.plane_stride;
    switch (D) {
        case 1: return reinterpret_cast<const void*>(pick_m_fixed<1>(M));
        case 2: return reinterpret_cast<const void*>(pick_m_fixed<2>(M));
        case 3: return reinterpret_cast<const void*>(pick_m_fixed<3>(M));
        default: return nullptr;
    }
}
#endif

#if defined(NUFFT_SPREAD_CUBES_GETTER)
// cube-accumulation variant of the compile-time-tile spreading kernel (or null)
const void* NUFFT_SPREAD_CUBES_GETTER(int D, int M) {
    return D == 3 ? reinterpret_cast<const void*>(pick_m_cubes<3>(M)) : nullptr;
}
#endif

#if defined(NUFFT_FIXED_DIMS_GETTER)
void NUFFT_FIXED_DIMS_GETTER(int D, int M, int* n) {
    const FixedTileDims fd = fixed_interp_tile((int)sizeof(NUFFT_T), NUFFT_CPLX ? 2 : 1, D, M);
    for (int d = 0; d < 3; ++d) n[d] = fd.n[d];
    n[3] = fd.row_stride;
}
#endif

}  // namespace nufft
