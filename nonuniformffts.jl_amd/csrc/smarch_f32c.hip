// spread_march_kernel instantiations for (float, complex = true): one per half-support M, halo variant and evaluation mode.
#define NUFFT_T float
#define NUFFT_CPLX true
#define NUFFT_CPLX_IS_TRUE 1
#define NUFFT_SMARCH_GETTER smarch_kernel_f32c
#define NUFFT_SMARCH_ZERO smarch_zero_bands_f32
#include "smarch_inst.h"
