// spread_patch_kernel instantiations with planar components for (float, real): ntransforms = 2, 3; M = 2..6.
#define NUFFT_T float
#define NUFFT_PATCH_PLANAR_GETTER patch_planar_kernel_f32r
#include "patch_planar_inst.h"
