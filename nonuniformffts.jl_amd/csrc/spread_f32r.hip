// spread kernels, T = float, complex = false (see tile_kernels.h).
#define NUFFT_T float
#define NUFFT_CPLX false
#define NUFFT_KERNEL spread_tile_kernel
#define NUFFT_GETTER spread_kernel_f32r
#define NUFFT_SPREAD_FIXED_GETTER spread_fixed_f32r
#define NUFFT_SPREAD_CUBES_GETTER spread_cubes_f32r
#include "tile_inst.h"
