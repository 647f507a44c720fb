// spread kernels, T = double, complex = true (see tile_kernels.h).
#define NUFFT_T double
#define NUFFT_CPLX true
#define NUFFT_KERNEL spread_tile_kernel
#define NUFFT_GETTER spread_kernel_f64c
#define NUFFT_SPREAD_FIXED_GETTER spread_fixed_f64c
#define NUFFT_SPREAD_CUBES_GETTER spread_cubes_f64c
#include "tile_inst.h"
