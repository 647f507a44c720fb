// interp kernels, T = double, complex = false (see tile_kernels.h).
#define NUFFT_T double
#define NUFFT_CPLX false
#define NUFFT_KERNEL interp_tile_kernel
#define NUFFT_GETTER interp_kernel_f64r
#define NUFFT_FIXED_DIMS_GETTER interp_fixed_dims_f64r
#include "tile_inst.h"
