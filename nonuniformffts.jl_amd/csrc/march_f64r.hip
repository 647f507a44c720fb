// interp_march_kernel instantiations for (double, complex = false): one per half-support M.
#define NUFFT_T double
#define NUFFT_CPLX false
#define NUFFT_MARCH_GETTER march_kernel_f64r
#define NUFFT_MARCH_GETTER_STAGED march_kernel_f64r_staged
#include "march_inst.h"
