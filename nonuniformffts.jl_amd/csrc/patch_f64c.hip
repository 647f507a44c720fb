// spread_patch_kernel instantiations for (double, complex = true): one per half-support M.
#define NUFFT_T double
#define NUFFT_CPLX true
#define NUFFT_PATCH_GETTER patch_kernel_f64c
#include "patch_inst.h"
