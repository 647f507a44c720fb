// interp_march_kernel instantiations for (float, complex = true): one per half-support M.
#define NUFFT_T float
#define NUFFT_CPLX true
#define NUFFT_MARCH_GETTER march_kernel_f32c
#define NUFFT_MARCH_GETTER_STAGED march_kernel_f32c_staged
#include "march_inst.h"
