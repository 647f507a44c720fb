// spread_patch_kernel instantiations for (float, complex = false): one per half-support M.
#define NUFFT_T float
#define NUFFT_CPLX false
#define NUFFT_PATCH_GETTER patch_kernel_f32r
#include "patch_inst.h"
