// graph_launch.hip — is a hipGraph the cheaper way to issue one short region of steps (the driver's K = 20: two chains of 20 dependent
// launches on two streams)?  Stand-in kernels of 6 us (clock spin, 782 single-wave blocks each), regions of K launches per chain, each
// region behind a synchronize (as bench.py brackets them): device time of a region (events) and wall time, for direct launches and for
// a graph captured once and replayed.
// build: hipcc -O2 --offload-arch=gfx950 tools/graph_launch.hip -o /tmp/graph_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin(long long ticks, unsigned* sink) {
  const long long t0 = wall_clock64();
  unsigned k = 0;
  while (wall_clock64() - t0 < ticks && k < 100000000u) k++;
  if (k == 0xFFFFFFFFu) *sink = k;
}

int main(int argc, char** argv) {
  const int    K = argc > 1 ? atoi(argv[1]) : 20, regions = 200;
  const double us = argc > 2 ? atof(argv[2]) : 6.0;
  unsigned* sink;
  CK(hipMalloc(&sink, 4));
  hipStream_t A, B;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, hi));
  hipEvent_t e0, e1, fork, join;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  auto issue = [&]() {
    CK(hipEventRecord(fork, A));
    CK(hipStreamWaitEvent(B, fork, 0));
    for (int k = 0; k < K; k++) {
      hipLaunchKernelGGL(spin, dim3(782), dim3(64), 0, A, (long long)(us * 100.0), sink);
      hipLaunchKernelGGL(spin, dim3(782), dim3(64), 0, B, (long long)(us * 100.0), sink);
    }
    CK(hipEventRecord(join, B));
    CK(hipStreamWaitEvent(A, join, 0));
  };
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(A, hipStreamCaptureModeThreadLocal));
  issue();
  CK(hipStreamEndCapture(A, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int mode = 0; mode < 2; mode++) {
    std::vector<double> dev, wall;
    for (int r = 0; r < regions + 5; r++) {
      CK(hipStreamSynchronize(A));
      const auto t0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(e0, A));
      if (mode == 0) issue(); else CK(hipGraphLaunch(ge, A));
      CK(hipEventRecord(e1, A));
      CK(hipStreamSynchronize(A));
      const double w = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 5) { dev.push_back(ms * 1e3 / K); wall.push_back(w / K); }
    }
    std::sort(dev.begin(), dev.end()); std::sort(wall.begin(), wall.end());
    printf("K = %d launches of %.0f us per chain, %s: device %.2f us per step (median; p10 %.2f, p90 %.2f), wall %.2f us per step\n", K, us,
           mode == 0 ? "direct launches" : "graph replay   ", dev[dev.size() / 2], dev[dev.size() / 10], dev[dev.size() * 9 / 10], wall[wall.size() / 2]);
  }
  return 0;
}
