// interp_march_kernel instantiations for (double, complex = true): one per half-support M.
#define NUFFT_T double
#define NUFFT_CPLX true
#define NUFFT_MARCH_GETTER march_kernel_f64c
#include "march_inst.h"
