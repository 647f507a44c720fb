// spread_march_dense_kernel instantiations for float (Float64 window and accumulators, as every spreading engine): one per half-support M = 2..6 and evaluation mode.
#define NUFFT_T float
#define NUFFT_DMARCH_GETTER dmarch_kernel_f32r
#include "dmarch_inst.h"
