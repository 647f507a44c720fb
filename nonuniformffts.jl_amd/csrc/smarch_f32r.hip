// spread_march_kernel instantiations for (float, complex = false): one per half-support M.
#define NUFFT_T float
#define NUFFT_CPLX false
#define NUFFT_SMARCH_GETTER smarch_kernel_f32r
#include "smarch_inst.h"
