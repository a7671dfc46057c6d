// precision of v_rsq_f64 / v_rcp_f64 raw and after Newton-Raphson refinement (run on the GPU box)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
__device__ double rsq_nr(double x, int it) {
  double y = __builtin_amdgcn_rsq(x);
  for (int i = 0; i < it; i++) { double e = __builtin_fma(-x * y, y, 1.0); y = __builtin_fma(0.5 * y, e, y); }
  return y;
}
__device__ double rcp_nr(double x, int it) {
  double y = __builtin_amdgcn_rcp(x);
  for (int i = 0; i < it; i++) { double e = __builtin_fma(-x, y, 1.0); y = __builtin_fma(y, e, y); }
  return y;
}
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  for (int it = 0; it < 3; it++) { o[(0 + it) * n + i] = rsq_nr(x[i], it); o[(3 + it) * n + i] = rcp_nr(x[i], it); }
}
int main() {
  const int n = 1 << 20; std::vector<double> x(n), o(6 * n);
  srand(1); for (int i = 0; i < n; i++) { double u = rand() / (double)RAND_MAX; x[i] = exp((u - 0.5) * 40.0) * (1.0 + rand() / (double)RAND_MAX); }
  double *dx, *dout; hipMalloc(&dx, 8 * n); hipMalloc(&dout, 8 * 6 * n); hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n); hipMemcpy(o.data(), dout, 8 * 6 * n, hipMemcpyDeviceToHost);
  const char* names[6] = {"rsq raw", "rsq+1NR", "rsq+2NR", "rcp raw", "rcp+1NR", "rcp+2NR"};
  for (int v = 0; v < 6; v++) { double worst = 0; for (int i = 0; i < n; i++) { long double ref = v < 3 ? 1.0L / sqrtl((long double)x[i]) : 1.0L / (long double)x[i];
      double rel = fabs((double)(((long double)o[v * n + i] - ref) / ref)); if (rel > worst) worst = rel; }
    printf("%-8s max rel err %.3e (%.2f ulp)\n", names[v], worst, worst / 1.11e-16); }
  return 0;
}
