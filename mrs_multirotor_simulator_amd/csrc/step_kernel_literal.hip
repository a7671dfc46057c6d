// LITERAL flavour of the fused step kernel: reference operation order, compiled with -ffp-contract=off.
#define MRS_FAST 0
#include "step_device.inc"

// variant: 0 = every input mode, 1 = model only (no UAV in a cascade mode)
extern "C" hipError_t mrs_launch_step_literal(SwarmDev sw, double dt, int substeps, int variant, hipStream_t st) {
  const int nb = (sw.n + 63) / 64;
  if (nb <= 0) return hipSuccess;
  if (variant == 1)
    hipLaunchKernelGGL(mrs_uav_model_step_literal, dim3(nb), dim3(64), 0, st, sw, dt, substeps);
  else
    hipLaunchKernelGGL(mrs_uav_step_literal, dim3(nb), dim3(64), 0, st, sw, dt, substeps);
  if (sw.n_mixed > 0) hipLaunchKernelGGL(mrs_uav_step_mixed_literal, dim3(sw.n_mixed), dim3(64), 0, st, sw, dt, substeps);
  return hipGetLastError();
}
