// spread_patch_kernel instantiations for (double, complex = false): one per half-support M.
#define NUFFT_T double
#define NUFFT_CPLX false
#define NUFFT_PATCH_GETTER patch_kernel_f64r
#include "patch_inst.h"
