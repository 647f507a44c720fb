// spread_march_dense_kernel instantiations for double: one per half-support M = 2..6 and evaluation mode.
#define NUFFT_T double
#define NUFFT_DMARCH_GETTER dmarch_kernel_f64r
#include "dmarch_inst.h"
