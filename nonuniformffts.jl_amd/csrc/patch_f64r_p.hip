// spread_patch_kernel instantiations with planar components for (double, real): ntransforms = 2, 3; M = 2..6.
#define NUFFT_T double
#define NUFFT_PATCH_PLANAR_GETTER patch_planar_kernel_f64r
#include "patch_planar_inst.h"
