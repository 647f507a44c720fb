// spread_march_kernel instantiations for (float, complex = false): one per half-support M, halo variant and evaluation mode.
#define NUFFT_T float
#define NUFFT_CPLX false
#define NUFFT_CPLX_IS_TRUE 0
#define NUFFT_SMARCH_GETTER smarch_kernel_f32r
#define NUFFT_SMARCH_HALO_ADD smarch_halo_add_f32
#include "smarch_inst.h"
