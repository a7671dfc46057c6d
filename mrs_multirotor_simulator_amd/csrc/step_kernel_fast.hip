// FAST flavour of the fused step kernel: FMA contraction + reciprocal simplifications (-ffp-contract=fast).
#define MRS_FAST 1
#include "step_device.inc"
