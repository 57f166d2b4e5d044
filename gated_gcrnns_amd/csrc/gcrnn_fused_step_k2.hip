// Instantiations of the fused step kernel (gcrnn_fused_step.h) for K = 2 taps.
#include "gcrnn_fused_step.h"

GCRNN_STEP_FOR_K2(GCRNN_STEP_DEFINE)
