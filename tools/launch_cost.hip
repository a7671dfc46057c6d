// Host cost of one kernel launch on this runtime, by the way it is made: hipLaunchKernelGGL against hipModuleLaunchKernel with a
// pre-resolved function and a packed argument block; one stream and two alternating streams; a kernel argument block as large as
// the step kernels' (SwarmDev + two doubles = ~200 B).  The kernel runs ~4 us on 782 single-wave blocks (a half-swarm launch's
// shape) so that the queues are neither empty nor backed up beyond what a 20-step run sees.
// build: hipcc -O2 --offload-arch=gfx950 tools/launch_cost.hip -o tools/bin/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct Args { double pad[24]; double dt, inv_dt; unsigned long long* out; long long spin; };
extern "C" __global__ void __launch_bounds__(64) k_work(Args a) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < a.spin) {}
  if (a.out && threadIdx.x == 0 && blockIdx.x == 0) a.out[0] = (unsigned long long)t0;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st[2];
  hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking);
  hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking);
  Args a; memset(&a, 0, sizeof a); a.spin = 400;  // 4 us
  hipFunction_t f = nullptr;
  hipError_t e = hipGetFuncBySymbol(&f, (const void*)k_work);
  printf("hipGetFuncBySymbol: %s\n", hipGetErrorString(e));
  size_t sz = sizeof a;
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &a, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
  for (int mode = 0; mode < 2; mode++)
    for (int ns = 1; ns <= 2; ns++)
      for (int K : {40, 400}) {
        double best = 1e30, best_region = 1e30;
        for (int rep = 0; rep < 12; rep++) {
          hipStreamSynchronize(st[0]); hipStreamSynchronize(st[1]);
          const double t0 = now();
          for (int k = 0; k < K; k++) {
            if (mode == 0) hipLaunchKernelGGL(k_work, dim3(782), dim3(64), 0, st[k % ns], a);
            else hipModuleLaunchKernel(f, 782, 1, 1, 64, 1, 1, 0, st[k % ns], nullptr, extra);
          }
          const double t1 = now();
          hipStreamSynchronize(st[0]); hipStreamSynchronize(st[1]);
          const double t2 = now();
          if (t1 - t0 < best) best = t1 - t0;
          if (t2 - t0 < best_region) best_region = t2 - t0;
        }
        printf("%-22s %d stream(s), %3d launches: %.2f us of host time per launch, region %.1f us (%.2f per launch)\n",
               mode ? "hipModuleLaunchKernel" : "hipLaunchKernelGGL", ns, K, best / K, best_region, best_region / K);
      }
  return 0;
}
