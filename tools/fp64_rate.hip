// micro-benchmark: FP64 VALU issue rate on one SIMD (v_fma_f64 / v_mul_f64 / v_add_f64 / v_rcp_f64 / v_rsq_f64 / f64 div / sqrt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(e) do{hipError_t _e=(e); if(_e!=hipSuccess){printf("hip error %s at %d\n", hipGetErrorString(_e), __LINE__); return 1;}}while(0)

template <int OP, int CHAINS>
__global__ void __launch_bounds__(64) k(double* out, double seed, int iters, unsigned long long* cyc, unsigned long long* rt) {
  double a[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) a[c] = seed + 1e-3 * (threadIdx.x + c);
  const double m = 1.0000001, b = 1e-9;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (OP == 0) a[c] = __builtin_fma(a[c], m, b);
      if (OP == 1) a[c] = a[c] * m;
      if (OP == 2) a[c] = a[c] + b;
      if (OP == 3) a[c] = __builtin_amdgcn_rcp(a[c]);
      if (OP == 4) a[c] = __builtin_amdgcn_rsq(a[c]);
      if (OP == 5) a[c] = 1.0 / a[c];
      if (OP == 6) a[c] = sqrt(a[c]);
      if (OP == 7) a[c] = (a[c] > 1.0) ? b : a[c] + m;  // cmp + cndmask + add
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += a[c];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int OP, int CHAINS>
int run(const char* name, int blocks) {
  double* out; unsigned long long *cyc, *rt;
  CK(hipMalloc(&out, sizeof(double) * 64 * blocks)); CK(hipMalloc(&cyc, 8 * blocks)); CK(hipMalloc(&rt, 8 * blocks));
  const int iters = 2000;
  for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL((k<OP, CHAINS>), dim3(blocks), dim3(64), 0, 0, out, 1.5, iters, cyc, rt);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> c(blocks), r(blocks);
  CK(hipMemcpy(c.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost)); CK(hipMemcpy(r.data(), rt, 8 * blocks, hipMemcpyDeviceToHost));
  double cs = 0, rs = 0; for (int i = 0; i < blocks; i++) { cs += c[i]; rs += r[i]; }
  cs /= blocks; rs /= blocks;
  printf("%-10s chains %d blocks %5d: %7.2f shader-cycles per op per wave, clock %.2f GHz (memtime/memrealtime*0.1)\n", name, CHAINS, blocks,
         cs / ((double)iters * CHAINS), cs / rs * 0.1);
  hipFree(out); hipFree(cyc); hipFree(rt);
  return 0;
}

int main() {
  for (int blocks : {1, 1024, 2048, 4096}) {
    run<0, 1>("fma", blocks); run<0, 4>("fma", blocks); run<0, 16>("fma", blocks);
    run<1, 16>("mul", blocks); run<2, 16>("add", blocks); run<3, 8>("rcp", blocks); run<4, 8>("rsq", blocks);
    run<5, 8>("div", blocks); run<6, 8>("sqrt", blocks); run<7, 8>("cmpsel+add", blocks);
  }
  return 0;
}
