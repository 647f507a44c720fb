// interp kernels, T = double, complex = true (see tile_kernels.h).
#define NUFFT_T double
#define NUFFT_CPLX true
#define NUFFT_KERNEL interp_tile_kernel
#define NUFFT_GETTER interp_kernel_f64c
#define NUFFT_FIXED_DIMS_GETTER interp_fixed_dims_f64c
#include "tile_inst.h"
