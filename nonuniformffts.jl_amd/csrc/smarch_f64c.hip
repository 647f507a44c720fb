// spread_march_kernel instantiations for (double, complex = true): one per half-support M.
#define NUFFT_T double
#define NUFFT_CPLX true
#define NUFFT_SMARCH_GETTER smarch_kernel_f64c
#include "smarch_inst.h"
