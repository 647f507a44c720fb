// spread_march_kernel instantiations for (float, complex = true): one per half-support M.
#define NUFFT_T float
#define NUFFT_CPLX true
#define NUFFT_SMARCH_GETTER smarch_kernel_f32c
#include "smarch_inst.h"
