// spread kernels, T = double, complex = false (see tile_kernels.h).
#define NUFFT_T double
#define NUFFT_CPLX false
#define NUFFT_KERNEL spread_tile_kernel
#define NUFFT_GETTER spread_kernel_f64r
#define NUFFT_SPREAD_FIXED_GETTER spread_fixed_f64r
#define NUFFT_SPREAD_CUBES_GETTER spread_cubes_f64r
#include "tile_inst.h"
