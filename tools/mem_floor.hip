// memory-phase floor of one step launch: read 33 SoA fields + flags, write 28 fields, (almost) no arithmetic.
// Variants: 8 B/lane field-major SoA (as the step kernel) vs 16 B/lane block-major AoSoA (double2 pairs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(e) do{hipError_t _e=(e); if(_e!=hipSuccess){printf("hip error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1;}}while(0)
constexpr int NR = 33, NW = 28, NF = 86;

__global__ void __launch_bounds__(64) k_soa(double* S, const unsigned* F, int n, int np) {
  int i = blockIdx.x * 64 + threadIdx.x; if (i >= n) return;
  double v[NR]; unsigned fl = F[i];
#pragma unroll
  for (int f = 0; f < NR; f++) v[f] = S[(size_t)f * np + i];
  double s = (double)fl;
#pragma unroll
  for (int f = 0; f < NR; f++) s += v[f];
#pragma unroll
  for (int f = 0; f < NW; f++) S[(size_t)f * np + i] = v[f] * 0.999 + s * 1e-9;
}
// block-major: block b holds NF fields x 64 lanes contiguous; lane loads double2 = (field 2k, field 2k+1) -> 1 KiB per instruction
__global__ void __launch_bounds__(64) k_aosoa(double2* S, const unsigned* F, int n) {
  int i = blockIdx.x * 64 + threadIdx.x; if (i >= n) return;
  double2* base = S + (size_t)blockIdx.x * (NF / 2) * 64 + threadIdx.x;
  double2 v[17]; unsigned fl = F[i];
#pragma unroll
  for (int f = 0; f < 17; f++) v[f] = base[(size_t)f * 64];
  double s = (double)fl;
#pragma unroll
  for (int f = 0; f < 17; f++) s += v[f].x + v[f].y;
#pragma unroll
  for (int f = 0; f < 14; f++) { double2 o; o.x = v[f].x * 0.999 + s * 1e-9; o.y = v[f].y * 0.999 + s * 1e-9; base[(size_t)f * 64] = o; }
}
int main() {
  for (int n : {65536, 100000, 131072, 524288, 1000000}) {
    int np = (n + 63) / 64 * 64; double* S; unsigned* F;
    CK(hipMalloc(&S, sizeof(double) * (size_t)NF * np)); CK(hipMalloc(&F, 4 * (size_t)np));
    CK(hipMemset(S, 0, sizeof(double) * (size_t)NF * np)); CK(hipMemset(F, 0, 4 * (size_t)np));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int variant = 0; variant < 2; variant++) {
      const int iters = 500;
      for (int w = 0; w < 50; w++) { if (variant == 0) hipLaunchKernelGGL(k_soa, dim3(np / 64), dim3(64), 0, 0, S, F, n, np); else hipLaunchKernelGGL(k_aosoa, dim3(np / 64), dim3(64), 0, 0, (double2*)S, F, n); }
      CK(hipDeviceSynchronize()); hipEventRecord(e0, 0);
      for (int w = 0; w < iters; w++) { if (variant == 0) hipLaunchKernelGGL(k_soa, dim3(np / 64), dim3(64), 0, 0, S, F, n, np); else hipLaunchKernelGGL(k_aosoa, dim3(np / 64), dim3(64), 0, 0, (double2*)S, F, n); }
      hipEventRecord(e1, 0); CK(hipDeviceSynchronize()); float ms; hipEventElapsedTime(&ms, e0, e1);
      double us = ms * 1e3 / iters; double bytes = (variant == 0 ? (NR + NW) * 8.0 + 4 : (34 + 28) * 8.0 + 4) * n;
      printf("N %8d %-6s %7.2f us/launch  %6.2f TB/s\n", n, variant == 0 ? "soa8B" : "aosoa16B", us, bytes / us * 1e-6);
    }
    hipFree(S); hipFree(F);
  }
  return 0;
}
