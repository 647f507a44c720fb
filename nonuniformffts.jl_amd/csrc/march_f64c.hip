// interp_march_kernel instantiations for (double, complex = true): one per half-support M.
#define NUFFT_T double
#define NUFFT_CPLX true
#define NUFFT_MARCH_GETTER march_kernel_f64c
#define NUFFT_MARCH_GETTER_STAGED march_kernel_f64c_staged
#include "march_inst.h"
