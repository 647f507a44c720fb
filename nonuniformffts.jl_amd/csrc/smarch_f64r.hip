// spread_march_kernel instantiations for (double, complex = false): one per half-support M.
#define NUFFT_T double
#define NUFFT_CPLX false
#define NUFFT_SMARCH_GETTER smarch_kernel_f64r
#include "smarch_inst.h"
