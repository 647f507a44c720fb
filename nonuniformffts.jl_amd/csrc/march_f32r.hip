// interp_march_kernel instantiations for (float, complex = false): one per half-support M.
#define NUFFT_T float
#define NUFFT_CPLX false
#define NUFFT_MARCH_GETTER march_kernel_f32r
#define NUFFT_MARCH_GETTER_STAGED march_kernel_f32r_staged
#include "march_inst.h"
