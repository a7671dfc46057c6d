// LITERAL flavour of the fused step kernel: reference operation order, compiled with -ffp-contract=off.
#define MRS_FAST 0
#include "step_device.inc"
