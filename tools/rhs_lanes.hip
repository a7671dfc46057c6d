// rhs_lanes.hip — VERDICT r2 item 5 / north_star "several lanes per UAV, wavefront shuffles": what the ODE right-hand side
// (MultirotorModel::operator(), multirotor_model.hpp:301-366 — re-orthonormalisation R L^-1, thrust / drag, R_dot, omega_dot) costs a
// LONE wave per SIMD in the two layouts, as a dependent chain (every evaluation feeds the next, like the four RK4 stages do):
//   A  one lane per UAV (the product's layout): 64 UAVs per wave, ~250 vector instructions per evaluation
//   B  three lanes per UAV: lane j holds x_j, v_j, omega_j and COLUMN j of R; dot products and the columns of the other two lanes come
//      through DPP row shuffles (quad_perm inside groups of four lanes, the fourth lane idle): 16 UAVs per wave
// Both compute the FAST arithmetic of step_device.inc (minors + three reciprocal square roots, no IEEE sqrt / divide); results are
// compared (B against A) before timing.  Grid: `waves` single-wave workgroups (7 = 400 UAVs in layout A, 25 in layout B).
// build + run: hipcc -O3 --offload-arch=gfx950 -ffp-contract=fast tools/rhs_lanes.hip -o /tmp/rhs_lanes && /tmp/rhs_lanes
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define DEV static __device__ __forceinline__

DEV double rsqrt_h(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-x * y, y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}

struct Par { double inv_mass, g, dk, thrust, t0, t1, t2, J0, J1, J2, Ji0, Ji1, Ji2, dt; };

// ---- layout A: y = x3 v3 R9 (row-major) w3, k = derivative -------------------------------------------------------------------------
DEV void rhs_one_lane(const Par& P, const double y[18], double k[18]) {
  const double* v = &y[3];
  const double* R = &y[6];
  const double* w = &y[15];
  const double a00 = (R[0] * R[0] + R[3] * R[3]) + R[6] * R[6], a10 = (R[1] * R[0] + R[4] * R[3]) + R[7] * R[6];
  const double a11 = (R[1] * R[1] + R[4] * R[4]) + R[7] * R[7], a20 = (R[2] * R[0] + R[5] * R[3]) + R[8] * R[6];
  const double a21 = (R[2] * R[1] + R[5] * R[4]) + R[8] * R[7], a22 = (R[2] * R[2] + R[5] * R[5]) + R[8] * R[8];
  const double m01 = a10 * a22 - a21 * a20, m02 = a10 * a21 - a11 * a20, m00 = a11 * a22 - a21 * a21;
  const double d1 = a00 * a11 - a10 * a10, d2 = (a00 * m00 - a10 * m01) + a20 * m02;
  const double r0 = rsqrt_h(a00), r1 = rsqrt_h(d1), r2 = rsqrt_h(d2);
  const double ra = r0, rc = (a00 * r0) * r1, rf = (d1 * r1) * r2;
  const double b = a10 * ra, d = a20 * ra, e = (a21 - d * b) * rc, erc = e * rc;
  const double i00 = ra, i11 = rc, i22 = rf, i10 = -(b * ra) * rc, i21 = -erc * rf, i20 = ((b * erc - d) * ra) * rf;
  double Rh[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double q0 = R[3 * i], q1 = R[3 * i + 1], q2 = R[3 * i + 2];
    Rh[3 * i] = (q0 * i00 + q1 * i10) + q2 * i20; Rh[3 * i + 1] = q1 * i11 + q2 * i21; Rh[3 * i + 2] = q2 * i22;
  }
  k[0] = v[0]; k[1] = v[1]; k[2] = v[2];
  const double sq = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2], vn = sq > 0 ? sq * rsqrt_h(sq) : sq, dk = P.dk * vn, tm = P.thrust * P.inv_mass;
  k[3] = tm * Rh[2] - dk * v[0]; k[4] = tm * Rh[5] - dk * v[1]; k[5] = (tm * Rh[8] - P.g) - dk * v[2];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double q0 = Rh[3 * i], q1 = Rh[3 * i + 1], q2 = Rh[3 * i + 2];
    k[6 + 3 * i] = q1 * w[2] - q2 * w[1]; k[7 + 3 * i] = q2 * w[0] - q0 * w[2]; k[8 + 3 * i] = q0 * w[1] - q1 * w[0];
  }
  const double Jw0 = P.J0 * w[0], Jw1 = P.J1 * w[1], Jw2 = P.J2 * w[2];
  k[15] = P.Ji0 * (P.t0 - (w[1] * Jw2 - w[2] * Jw1)); k[16] = P.Ji1 * (P.t1 - (w[2] * Jw0 - w[0] * Jw2)); k[17] = P.Ji2 * (P.t2 - (w[0] * Jw1 - w[1] * Jw0));
}

__global__ void __launch_bounds__(64, 1) chain_one_lane(const double* y0, double* out, Par P, int reps) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  double y[18], k[18];
#pragma unroll
  for (int c = 0; c < 18; c++) y[c] = y0[(size_t)c * gridDim.x * 64 + i];
  for (int r = 0; r < reps; r++) {  // an explicit Euler march: every evaluation depends on the previous one
    rhs_one_lane(P, y, k);
#pragma unroll
    for (int c = 0; c < 18; c++) y[c] = __builtin_fma(P.dt, k[c], y[c]);
  }
#pragma unroll
  for (int c = 0; c < 18; c++) out[(size_t)c * gridDim.x * 64 + i] = y[c];
}

// ---- layout B: lane j (of a quad, j < 3) holds x_j v_j w_j and column j of R: c[0..2] = R[0][j], R[1][j], R[2][j] ---------------------
// DPP quad_perm moves inside groups of four lanes: q(v, a) = the value of lane (quad base + a)
template <int SEL>
DEV double quad(double v) {
  const int lo = __builtin_amdgcn_mov_dpp((int)__double2loint(v), SEL, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp((int)__double2hiint(v), SEL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
#define QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
DEV double from0(double v) { return quad<QP(0, 0, 0, 0)>(v); }
DEV double from1(double v) { return quad<QP(1, 1, 1, 1)>(v); }
DEV double from2(double v) { return quad<QP(2, 2, 2, 2)>(v); }
DEV double next1(double v) { return quad<QP(1, 2, 0, 3)>(v); }  // lane j gets the value of lane (j + 1) % 3
DEV double next2(double v) { return quad<QP(2, 0, 1, 3)>(v); }  // ... of lane (j + 2) % 3

struct Lane3 { double x, v, w, c0, c1, c2; };

DEV void rhs_three_lanes(const Par& P, const int j, const double Jj, const double Jij, const double tj, const Lane3& s, Lane3& k) {
  // columns of the other two lanes (cyclic): p = column (j+1)%3, q = column (j+2)%3
  const double p0 = next1(s.c0), p1 = next1(s.c1), p2 = next1(s.c2), q0 = next2(s.c0), q1 = next2(s.c1), q2 = next2(s.c2);
  // A = R^T R: this lane's row of it, in cyclic order: ajj, aj(j+1), aj(j+2)
  const double ajj = (s.c0 * s.c0 + s.c1 * s.c1) + s.c2 * s.c2, ajp = (s.c0 * p0 + s.c1 * p1) + s.c2 * p2, ajq = (s.c0 * q0 + s.c1 * q1) + s.c2 * q2;
  // every lane needs the six entries for ITS column of L^-1; bring them to all lanes (a00 a10 a11 a20 a21 a22)
  const double a00 = from0(ajj), a11 = from1(ajj), a22 = from2(ajj), a10 = from0(ajp), a21 = from1(ajp), a20 = from2(ajp);  // lane0: a01, lane1: a12, lane2: a20
  (void)ajq;
  // minors: lane 0 -> a00, lane 1 -> d1, lane 2 -> d2; ONE reciprocal square root per lane, then shared
  const double m01 = a10 * a22 - a21 * a20, m02 = a10 * a21 - a11 * a20, m00 = a11 * a22 - a21 * a21;
  const double d1 = a00 * a11 - a10 * a10, d2 = (a00 * m00 - a10 * m01) + a20 * m02;
  const double mine = j == 0 ? a00 : (j == 1 ? d1 : d2);
  const double rj = rsqrt_h(mine);
  const double r0 = from0(rj), r1 = from1(rj), r2 = from2(rj);
  const double ra = r0, rc = (a00 * r0) * r1, rf = (d1 * r1) * r2;
  const double b = a10 * ra, d = a20 * ra, e = (a21 - d * b) * rc, erc = e * rc;
  // column j of L^-1 (rows 0..2): col0 = (i00, i10, i20), col1 = (0, i11, i21), col2 = (0, 0, i22)
  const double i0 = j == 0 ? ra : 0.0;
  const double i1 = j == 0 ? -(b * ra) * rc : (j == 1 ? rc : 0.0);
  const double i2 = j == 0 ? ((b * erc - d) * ra) * rf : (j == 1 ? -erc * rf : rf);
  // column j of Rh = R * (column j of L^-1) = col0 * i0 + col1 * i1 + col2 * i2: needs the three columns of R in FIXED order
  const double A0 = from0(s.c0), A1 = from0(s.c1), A2 = from0(s.c2), B0 = from1(s.c0), B1 = from1(s.c1), B2 = from1(s.c2), C0 = from2(s.c0), C1 = from2(s.c1),
               C2 = from2(s.c2);
  const double h0 = (A0 * i0 + B0 * i1) + C0 * i2, h1 = (A1 * i0 + B1 * i1) + C1 * i2, h2 = (A2 * i0 + B2 * i1) + C2 * i2;  // Rh[0..2][j]
  // translation: lane j needs Rh[j][2] = component j of lane 2's column
  const double z0 = from2(h0), z1 = from2(h1), z2 = from2(h2);
  const double zj = j == 0 ? z0 : (j == 1 ? z1 : z2);
  const double vv = s.v * s.v, sq = (from0(vv) + from1(vv)) + from2(vv), vn = sq > 0 ? sq * rsqrt_h(sq) : sq, dk = P.dk * vn, tm = P.thrust * P.inv_mass;
  k.x = s.v;
  k.v = (tm * zj - (j == 2 ? P.g : 0.0)) - dk * s.v;
  // R_dot column j = Rh * (column j of the omega tensor): col0 = Rh1 w2 - Rh2 w1, col1 = Rh2 w0 - Rh0 w2, col2 = Rh0 w1 - Rh1 w0  (cyclic)
  const double wp = next1(s.w), wq = next2(s.w);  // w_(j+1), w_(j+2)
  const double hp0 = next1(h0), hp1 = next1(h1), hp2 = next1(h2), hq0 = next2(h0), hq1 = next2(h1), hq2 = next2(h2);
  k.c0 = hp0 * wq - hq0 * wp; k.c1 = hp1 * wq - hq1 * wp; k.c2 = hp2 * wq - hq2 * wp;
  // omega_dot_j = Jinv_j (t_j - (w_(j+1) J_(j+2) w_(j+2) - w_(j+2) J_(j+1) w_(j+1)))
  const double Jw = Jj * s.w, Jwp = next1(Jw), Jwq = next2(Jw);
  k.w = Jij * (tj - (wp * Jwq - wq * Jwp));
}

__global__ void __launch_bounds__(64, 1) chain_three_lanes(const double* y0, double* out, Par P, int reps, int n_uav) {
  const int lane = threadIdx.x, j = lane & 3, u = blockIdx.x * 16 + (lane >> 2);
  const bool live = j < 3 && u < n_uav;
  const int  jj = j < 3 ? j : 0, uu = u < n_uav ? u : 0;
  const size_t np = (size_t)n_uav;
  Lane3 s, k;
  s.x = y0[(size_t)(0 + jj) * np + uu]; s.v = y0[(size_t)(3 + jj) * np + uu]; s.w = y0[(size_t)(15 + jj) * np + uu];
  s.c0 = y0[(size_t)(6 + 0 + jj) * np + uu]; s.c1 = y0[(size_t)(6 + 3 + jj) * np + uu]; s.c2 = y0[(size_t)(6 + 6 + jj) * np + uu];
  const double Jj = jj == 0 ? P.J0 : (jj == 1 ? P.J1 : P.J2), Jij = jj == 0 ? P.Ji0 : (jj == 1 ? P.Ji1 : P.Ji2), tj = jj == 0 ? P.t0 : (jj == 1 ? P.t1 : P.t2);
  for (int r = 0; r < reps; r++) {
    rhs_three_lanes(P, jj, Jj, Jij, tj, s, k);
    s.x = __builtin_fma(P.dt, k.x, s.x); s.v = __builtin_fma(P.dt, k.v, s.v); s.w = __builtin_fma(P.dt, k.w, s.w);
    s.c0 = __builtin_fma(P.dt, k.c0, s.c0); s.c1 = __builtin_fma(P.dt, k.c1, s.c1); s.c2 = __builtin_fma(P.dt, k.c2, s.c2);
  }
  if (live) {
    out[(size_t)(0 + j) * np + u] = s.x; out[(size_t)(3 + j) * np + u] = s.v; out[(size_t)(15 + j) * np + u] = s.w;
    out[(size_t)(6 + 0 + j) * np + u] = s.c0; out[(size_t)(6 + 3 + j) * np + u] = s.c1; out[(size_t)(6 + 6 + j) * np + u] = s.c2;
  }
}

int main(int argc, char** argv) {
  const int n_uav = argc > 1 ? atoi(argv[1]) : 400, reps = 2000;
  const int wavesA = (n_uav + 63) / 64, wavesB = (n_uav + 15) / 16, nA = wavesA * 64;
  Par P = {1.0 / 2.0, 9.81, 0.002, 19.6, 0.01, -0.02, 0.005, 0.0329, 0.0329, 0.0625, 1 / 0.0329, 1 / 0.0329, 1 / 0.0625, 2.5e-4};
  std::vector<double> h((size_t)18 * nA, 0.0), a((size_t)18 * nA), bb((size_t)18 * nA);
  srand(1);
  for (int i = 0; i < nA; i++) {
    double q[4], n = 0;
    for (double& x : q) { x = rand() / (double)RAND_MAX - 0.5; n += x * x; }
    for (double& x : q) x /= sqrt(n);
    const double R[9] = {1 - 2 * (q[2] * q[2] + q[3] * q[3]), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2]),
                         2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1] * q[1] + q[3] * q[3]), 2 * (q[2] * q[3] - q[0] * q[1]),
                         2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1] * q[1] + q[2] * q[2])};
    for (int c = 0; c < 3; c++) { h[(size_t)c * nA + i] = rand() % 100; h[(size_t)(3 + c) * nA + i] = rand() / (double)RAND_MAX - 0.5; h[(size_t)(15 + c) * nA + i] = rand() / (double)RAND_MAX - 0.5; }
    for (int c = 0; c < 9; c++) h[(size_t)(6 + c) * nA + i] = R[c];
  }
  double *dy, *dA, *dB;
  CK(hipMalloc(&dy, sizeof(double) * h.size())); CK(hipMalloc(&dA, sizeof(double) * h.size())); CK(hipMalloc(&dB, sizeof(double) * h.size()));
  CK(hipMemcpy(dy, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
  CK(hipMemset(dB, 0, sizeof(double) * h.size()));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float msA = 0, msB = 0;
  for (int pass = 0; pass < 3; pass++) {
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(chain_one_lane, dim3(wavesA), dim3(64), 0, 0, dy, dA, P, reps); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&msA, e0, e1));
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(chain_three_lanes, dim3(wavesB), dim3(64), 0, 0, dy, dB, P, reps, nA); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&msB, e0, e1));
  }
  CK(hipMemcpy(a.data(), dA, sizeof(double) * a.size(), hipMemcpyDeviceToHost));
  CK(hipMemcpy(bb.data(), dB, sizeof(double) * bb.size(), hipMemcpyDeviceToHost));
  double worst = 0;
  for (int c = 0; c < 18; c++)
    for (int i = 0; i < n_uav; i++) {
      const double x = a[(size_t)c * nA + i], y = bb[(size_t)c * nA + i], dlt = fabs(x - y) / fmax(fabs(x), 1.0);
      if (dlt > worst || dlt != dlt) worst = dlt != dlt ? 1e300 : dlt;
    }
  printf("%d UAVs, %d chained evaluations of the right-hand side (FAST arithmetic), one wave per SIMD:\n", n_uav, reps);
  printf("  A one lane per UAV   : %3d waves, %.3f us per evaluation\n", wavesA, msA * 1e3 / reps);
  printf("  B three lanes per UAV: %3d waves, %.3f us per evaluation   (B / A = %.2f; largest relative difference of the final states %.2e)\n", wavesB,
         msB * 1e3 / reps, msB / msA, worst);
  return 0;
}
