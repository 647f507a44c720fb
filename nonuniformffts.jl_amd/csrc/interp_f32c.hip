// interp kernels, T = float, complex = true (see tile_kernels.h).
#define NUFFT_T float
#define NUFFT_CPLX true
#define NUFFT_KERNEL interp_tile_kernel
#define NUFFT_GETTER interp_kernel_f32c
#define NUFFT_FIXED_DIMS_GETTER interp_fixed_dims_f32c
#include "tile_inst.h"
