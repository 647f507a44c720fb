// spread_patch_kernel instantiations for (float, complex = true): one per half-support M.
#define NUFFT_T float
#define NUFFT_CPLX true
#define NUFFT_PATCH_GETTER patch_kernel_f32c
#include "patch_inst.h"
