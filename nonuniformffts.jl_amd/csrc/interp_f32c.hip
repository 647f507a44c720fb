// interp kernels, T = float, complex = true (see tile_kernels.h).
#define NUFFT_T float
#define NUFFT_CPLX true
#define NUFFT_KERNEL interp_tile_kernel
#define NUFFT_GETTER interp_kernel_f32c
#define NUFFT_HAS_WRAP_VARIANT 0
#include "tile_inst.h"
