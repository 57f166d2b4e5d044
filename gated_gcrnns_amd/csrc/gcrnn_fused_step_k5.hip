// Instantiations of the fused step kernel (gcrnn_fused_step.h) for K = 5 taps.
#define GCRNN_SEQ_STAMPS_READER      // diagnostic builds (-DGCRNN_SEQ_STAMPS): this unit exports the stamp reader of its kernels
#include "gcrnn_fused_step.h"

GCRNN_STEP_FOR_K5(GCRNN_STEP_DEFINE)
