// Instantiations of the fused step kernel (gcrnn_fused_step.h) for K = 4 taps.
#include "gcrnn_fused_step.h"

GCRNN_STEP_FOR_K4(GCRNN_STEP_DEFINE)
