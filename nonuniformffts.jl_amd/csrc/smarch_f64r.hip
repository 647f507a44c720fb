// spread_march_kernel instantiations for (double, complex = false): one per half-support M, halo variant and evaluation mode.
#define NUFFT_T double
#define NUFFT_CPLX false
#define NUFFT_CPLX_IS_TRUE 0
#define NUFFT_SMARCH_GETTER smarch_kernel_f64r
#define NUFFT_SMARCH_HALO_ADD smarch_halo_add_f64
#include "smarch_inst.h"
