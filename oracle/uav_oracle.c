/*
 * uav_oracle.c — CPU oracle (test infrastructure, NOT product code; see uav_oracle.h).
 *
 * Scalar FP64 C99 restatement of the reference's UavSystem::makeStep() path.
 * Every function cites the reference lines it follows; paths are relative to
 * /root/reference/include/mrs_multirotor_simulator/uav_system unless absolute.
 * Build with -O2 -ffp-contract=off (never -ffast-math): NaN semantics are load-bearing.
 *
 * Parity status: dynamics/cascade "parity unpinned" (no reference tests/vectors
 * exist; reference headers need Eigen3+Boost, absent here), EXCEPT the PID
 * (pid.hpp compiles alone: pinned bit for bit against the reference's own class,
 * oracle/_ref/libref_pid.so + tests/golden/pid_reference_vectors.npz); collision
 * neighbour set pinned against the reference's nanoflann (oracle/_ref).  Cross-checked at 1e-9
 * against a second, independently written restatement (tests/independent_model.py).
 */
#include "uav_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ */
/* Eigen leaf semantics restated                                        */
/* ------------------------------------------------------------------ */

#define M3(m, i, j) ((m)[(i)*3 + (j)])

/* fixed-size 3x3 * 3x3 lazy coefficient product: (a0*b0 + a1*b1) + a2*b2 */
static void mat3_mul(const double a[9], const double b[9], double out[9]) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      t[i * 3 + j] = (M3(a, i, 0) * M3(b, 0, j) + M3(a, i, 1) * M3(b, 1, j)) + M3(a, i, 2) * M3(b, 2, j);
  memcpy(out, t, sizeof t);
}

static void mat3_transpose(const double a[9], double out[9]) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[i * 3 + j] = M3(a, j, i);
  memcpy(out, t, sizeof t);
}

static void mat3_vec(const double a[9], const double v[3], double out[3]) {
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = (M3(a, i, 0) * v[0] + M3(a, i, 1) * v[1]) + M3(a, i, 2) * v[2];
  memcpy(out, t, sizeof t);
}

static double vec3_sqnorm(const double v[3]) { return (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]; }
static double vec3_norm(const double v[3]) { return sqrt(vec3_sqnorm(v)); }
static double vec3_dot(const double a[3], const double b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

/* Eigen normalize()/normalized(): z = squaredNorm(); if (z > 0) v /= sqrt(z) */
static void vec3_normalize(double v[3]) {
  double z = vec3_sqnorm(v);
  if (z > 0) {
    double n = sqrt(z);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
  }
}

static void vec3_cross(const double a[3], const double b[3], double out[3]) {
  double t[3];
  t[0] = a[1] * b[2] - a[2] * b[1];
  t[1] = a[2] * b[0] - a[0] * b[2];
  t[2] = a[0] * b[1] - a[1] * b[0];
  memcpy(out, t, sizeof t);
}

/* Eigen cofactor_3x3<i,j>: m(i1,j1)*m(i2,j2) - m(i1,j2)*m(i2,j1) */
static double cofactor3(const double m[9], int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return M3(m, i1, j1) * M3(m, i2, j2) - M3(m, i1, j2) * M3(m, i2, j1);
}

/* Eigen compute_inverse<Matrix3d>: cofactors of column 0, det = sum(c_col0 .* m.col(0)), invdet = 1/det */
void orc_inverse3(const double m[9], double out[9]) {
  double c0[3] = {cofactor3(m, 0, 0), cofactor3(m, 1, 0), cofactor3(m, 2, 0)};
  double det    = (c0[0] * M3(m, 0, 0) + c0[1] * M3(m, 1, 0)) + c0[2] * M3(m, 2, 0);
  double invdet = 1.0 / det;
  double r[9];
  M3(r, 0, 0) = c0[0] * invdet;
  M3(r, 0, 1) = c0[1] * invdet;
  M3(r, 0, 2) = c0[2] * invdet;
  M3(r, 1, 0) = cofactor3(m, 0, 1) * invdet;
  M3(r, 1, 1) = cofactor3(m, 1, 1) * invdet;
  M3(r, 1, 2) = cofactor3(m, 2, 1) * invdet;
  M3(r, 2, 0) = cofactor3(m, 0, 2) * invdet;
  M3(r, 2, 1) = cofactor3(m, 1, 2) * invdet;
  M3(r, 2, 2) = cofactor3(m, 2, 2) * invdet;
  memcpy(out, r, sizeof r);
}

/* Eigen LLT<Matrix3d> (llt_inplace<Lower>::unblocked) followed by matrixL() -> dense lower-triangular P.
 * A failing pivot (x <= 0) stops the factorisation and leaves the remaining columns untouched. */
static void llt_lower3(const double a[9], double P[9]) {
  double m[9];
  memcpy(m, a, sizeof m);
  for (int k = 0; k < 3; k++) {
    double x = M3(m, k, k);
    if (k == 1) x -= M3(m, 1, 0) * M3(m, 1, 0);
    if (k == 2) x -= (M3(m, 2, 0) * M3(m, 2, 0) + M3(m, 2, 1) * M3(m, 2, 1));
    if (x <= 0) break;
    x           = sqrt(x);
    M3(m, k, k) = x;
    if (k == 1) M3(m, 2, 1) -= M3(m, 2, 0) * M3(m, 1, 0);
    for (int r = k + 1; r < 3; r++) M3(m, r, k) /= x;
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M3(P, i, j) = (j <= i) ? M3(m, i, j) : 0.0;
}

/* multirotor_model.hpp:249-253 and :314-316:  R * inverse(matrixL(LLT(R^T R))) */
void orc_llt_reorth(const double R[9], double out[9]) {
  double Rt[9], RtR[9], P[9], Pinv[9];
  mat3_transpose(R, Rt);
  mat3_mul(Rt, R, RtR);
  llt_lower3(RtR, P);
  orc_inverse3(P, Pinv);
  mat3_mul(R, Pinv, out);
}

/* Eigen dynamic-size inverse() = PartialPivLU().inverse(): LU with row pivoting, then solve against I.
 * Row-major n x n in/out. Returns 0 on success (no singularity check in the reference either). */
int orc_inverse_lu(const double* a, int n, double* out) {
  double lu[64], b[64];
  int    perm[8];
  if (n > 8) return -1;
  for (int i = 0; i < n * n; i++) lu[i] = a[i];
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) {
    int    piv = k;
    double big = fabs(lu[k * n + k]);
    for (int r = k + 1; r < n; r++) {
      double val = fabs(lu[r * n + k]);
      if (val > big) {
        big = val;
        piv = r;
      }
    }
    if (piv != k) {
      for (int c = 0; c < n; c++) {
        double t        = lu[k * n + c];
        lu[k * n + c]   = lu[piv * n + c];
        lu[piv * n + c] = t;
      }
      int t     = perm[k];
      perm[k]   = perm[piv];
      perm[piv] = t;
    }
    for (int r = k + 1; r < n; r++) lu[r * n + k] /= lu[k * n + k];
    for (int r = k + 1; r < n; r++)
      for (int c = k + 1; c < n; c++) lu[r * n + c] -= lu[r * n + k] * lu[k * n + c];
  }
  /* solve L U X = P I, column by column */
  for (int col = 0; col < n; col++) {
    for (int r = 0; r < n; r++) b[r] = (perm[r] == col) ? 1.0 : 0.0;
    for (int r = 0; r < n; r++)
      for (int c = 0; c < r; c++) b[r] -= lu[r * n + c] * b[c];
    for (int r = n - 1; r >= 0; r--) {
      for (int c = r + 1; c < n; c++) b[r] -= lu[r * n + c] * b[c];
      b[r] /= lu[r * n + r];
    }
    for (int r = 0; r < n; r++) out[r * n + col] = b[r];
  }
  return 0;
}

/* Eigen redux (sum) over a dynamic VectorXd, SSE2 packets of 2 doubles, 16-byte aligned heap data:
 * redux_impl<..., LinearVectorizedTraversal, NoUnrolling>. */
static double dyn_sum(const double* v, int n) {
  const int ps = 2;
  int       aligned_size2 = (n / (2 * ps)) * (2 * ps);
  int       aligned_size  = (n / ps) * ps;
  double    res;
  if (aligned_size) {
    double p0[2] = {v[0], v[1]};
    if (aligned_size > ps) {
      double p1[2] = {v[2], v[3]};
      for (int idx = 2 * ps; idx < aligned_size2; idx += 2 * ps) {
        p0[0] += v[idx];
        p0[1] += v[idx + 1];
        p1[0] += v[idx + 2];
        p1[1] += v[idx + 3];
      }
      p0[0] += p1[0];
      p0[1] += p1[1];
      if (aligned_size > aligned_size2) {
        p0[0] += v[aligned_size2];
        p0[1] += v[aligned_size2 + 1];
      }
    }
    res = p0[0] + p0[1];
    for (int idx = aligned_size; idx < n; idx++) res += v[idx];
  } else {
    res = v[0];
    for (int idx = 1; idx < n; idx++) res += v[idx];
  }
  return res;
}
static double dyn_mean(const double* v, int n) { return dyn_sum(v, n) / (double)n; }

/* ------------------------------------------------------------------ */
/* data model                                                           */
/* ------------------------------------------------------------------ */

typedef struct {
  double kp, kd, ki, saturation, antiwindup; /* pid.hpp:15-25 */
  double last_error, integral;
} pid_state_t;

typedef struct {
  /* MultirotorModel (multirotor_model.hpp:133-150) */
  orc_model_params_t p;
  double x[3], v[3], v_prev[3], R[9], omega[3], motor_rpm[ORC_MAX_MOTORS];
  double imu[3], input[ORC_MAX_MOTORS], ext_force[3], ext_moment[3], initial_pos[3];
  double y[18]; /* internal_state_ */
  /* UavSystem (uav_system.hpp:77-117) */
  int crashed, active_input;
  int hold; /* UavSystemRos: no input yet / input timed out and iterate_without_input == false (src/uav_system_ros.cpp:265) */
  double actuators[ORC_MAX_MOTORS];
  double control_group[4];            /* roll pitch yaw throttle */
  double attitude_rate[4];            /* rx ry rz throttle */
  double attitude_R[9], attitude_throttle;
  double tilt[3], tilt_heading_rate, tilt_throttle;
  double acc_hr[3], acc_hr_rate;
  double acc_h[3], acc_h_heading;
  double vel_hr[3], vel_hr_rate;
  double vel_h[3], vel_h_heading;
  double pos[3], pos_heading;
  int    has_ff[4];
  double ff_vec[4][3], ff_scalar[4];
  /* controllers */
  orc_mixer_params_t    mixer_p;
  double                alloc_inv[ORC_MAX_MOTORS * 4]; /* n x 4 */
  orc_rate_params_t     rate_p;
  orc_attitude_params_t att_p;
  orc_velocity_params_t vel_p;
  orc_position_params_t pos_p;
  pid_state_t           pid_pos[3], pid_vel[3], pid_att[3], pid_rate[3];
} uav_t;

struct orc_swarm {
  int32_t    n;
  uav_t*     u;
  orc_diag_t diag;
};

/* ------------------------------------------------------------------ */
/* params                                                               */
/* ------------------------------------------------------------------ */

void orc_calculate_inertia(orc_model_params_t* p) { /* src/uav_system_ros.cpp:664-671 == multirotor_model.hpp:44-47 */
  memset(p->J, 0, sizeof p->J);
  p->J[0] = p->mass * (3.0 * p->arm_length * p->arm_length + p->body_height * p->body_height) / 12.0;
  p->J[4] = p->mass * (3.0 * p->arm_length * p->arm_length + p->body_height * p->body_height) / 12.0;
  p->J[8] = (p->mass * p->arm_length * p->arm_length) / 2.0;
}

void orc_scale_allocation(orc_model_params_t* p) { /* src/uav_system_ros.cpp:100-103 == multirotor_model.hpp:59-62 */
  for (int m = 0; m < p->n_motors; m++) {
    p->allocation_matrix[0 * ORC_MAX_MOTORS + m] *= p->arm_length * p->kf;
    p->allocation_matrix[1 * ORC_MAX_MOTORS + m] *= p->arm_length * p->kf;
    p->allocation_matrix[2 * ORC_MAX_MOTORS + m] *= p->km * (3.0 * p->prop_radius) * p->kf;
    p->allocation_matrix[3 * ORC_MAX_MOTORS + m] *= p->kf;
  }
}

void orc_model_params_default(orc_model_params_t* p) { /* multirotor_model.hpp:26-66 */
  static const double a[4][4] = {{-0.707, 0.707, 0.707, -0.707}, {-0.707, 0.707, -0.707, 0.707}, {-1, -1, 1, 1}, {1, 1, 1, 1}};
  memset(p, 0, sizeof *p);
  p->n_motors             = 4;
  p->g                    = 9.81;
  p->mass                 = 2.0;
  p->kf                   = 0.00000027087;
  p->km                   = 0.07;
  p->prop_radius          = 0.15;
  p->arm_length           = 0.25;
  p->body_height          = 0.1;
  p->motor_time_constant  = 0.03;
  p->max_rpm              = 7800;
  p->min_rpm              = 1170;
  p->air_resistance_coeff = 0.30;
  orc_calculate_inertia(p);
  for (int r = 0; r < 4; r++)
    for (int m = 0; m < 4; m++) p->allocation_matrix[r * ORC_MAX_MOTORS + m] = a[r][m];
  orc_scale_allocation(p);
  p->ground_enabled        = 0;
  p->ground_z              = 0.0; /* uninitialised in the reference (:85) */
  p->takeoff_patch_enabled = 1;
}

/* ------------------------------------------------------------------ */
/* PID — pid.hpp:67-96                                                  */
/* ------------------------------------------------------------------ */

double orc_pid_update(double kp, double kd, double ki, double saturation, double antiwindup, double* last_error,
                      double* integral, double error, double dt) {
  double difference = (error - *last_error) / dt;
  *last_error       = error;
  double p_component = kp * error;
  double d_component = kd * difference;
  double i_component = ki * *integral;
  double sum         = p_component + d_component + i_component;
  if (saturation > 0) {
    if (sum >= saturation) {
      sum = saturation;
    } else if (sum <= -saturation) {
      sum = -saturation;
    }
  }
  if (antiwindup > 0) {
    if (fabs(sum) < antiwindup) {
      *integral += error * dt;
    }
  }
  return sum;
}

static double pid_update(pid_state_t* c, double error, double dt) {
  return orc_pid_update(c->kp, c->kd, c->ki, c->saturation, c->antiwindup, &c->last_error, &c->integral, error, dt);
}

static void pid_set(pid_state_t* c, double kp, double kd, double ki, double sat, double aw) {
  c->last_error = 0; /* reset(), pid.hpp:61-65 */
  c->integral   = 0;
  c->kp         = kp;
  c->kd         = kd;
  c->ki         = ki;
  c->saturation = sat;
  c->antiwindup = aw;
}

/* ------------------------------------------------------------------ */
/* controllers: init                                                    */
/* ------------------------------------------------------------------ */

static void rate_init_pids(uav_t* u) { /* rate_controller.hpp:56-65 */
  for (int i = 0; i < 3; i++) {
    double Jii = u->p.J[i * 3 + i];
    pid_set(&u->pid_rate[i], u->rate_p.kp * Jii, u->rate_p.kd * Jii, u->rate_p.ki * Jii, -1, 1.0);
  }
}
static void att_init_pids(uav_t* u) { /* attitude_controller.hpp:160-171 */
  pid_set(&u->pid_att[0], u->att_p.kp, u->att_p.kd, u->att_p.ki, u->att_p.max_rate_roll_pitch, 0.1);
  pid_set(&u->pid_att[1], u->att_p.kp, u->att_p.kd, u->att_p.ki, u->att_p.max_rate_roll_pitch, 0.1);
  pid_set(&u->pid_att[2], u->att_p.kp, u->att_p.kd, u->att_p.ki, u->att_p.max_rate_yaw, 0.1);
}
static void vel_init_pids(uav_t* u) { /* velocity_controller.hpp:108-119 */
  for (int i = 0; i < 3; i++) pid_set(&u->pid_vel[i], u->vel_p.kp, u->vel_p.kd, u->vel_p.ki, u->vel_p.max_acceleration, 1.0);
}
static void pos_init_pids(uav_t* u) { /* position_controller.hpp:92-103 */
  for (int i = 0; i < 3; i++) pid_set(&u->pid_pos[i], u->pos_p.kp, u->pos_p.kd, u->pos_p.ki, u->pos_p.max_velocity, 1.0);
}

/* Mixer::calculateAllocation — mixer.hpp:72-101 */
static void mixer_calculate_allocation(uav_t* u) {
  const int n = u->p.n_motors;
  double    AAt[16], AAt_inv[16];
  const double* A = u->p.allocation_matrix;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < n; k++) s += A[i * ORC_MAX_MOTORS + k] * A[j * ORC_MAX_MOTORS + k];
      AAt[i * 4 + j] = s;
    }
  orc_inverse_lu(AAt, 4, AAt_inv);
  for (int m = 0; m < n; m++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += A[k * ORC_MAX_MOTORS + m] * AAt_inv[k * 4 + j];
      u->alloc_inv[m * 4 + j] = s;
    }
  for (int m = 0; m < n; m++) { /* block(i,0,1,2).normalize() */
    double* r = &u->alloc_inv[m * 4];
    double  z = r[0] * r[0] + r[1] * r[1];
    if (z > 0) {
      double nn = sqrt(z);
      r[0] /= nn;
      r[1] /= nn;
    }
  }
  for (int m = 0; m < n; m++) {
    double* r = &u->alloc_inv[m * 4];
    if (r[2] > 1e-2) {
      r[2] = 1.0;
    } else if (r[2] < -1e-2) {
      r[2] = -1.0;
    } else {
      r[2] = 0.0;
    }
  }
  for (int m = 0; m < n; m++) u->alloc_inv[m * 4 + 3] = 1.0;
}

/* UavSystem::initializeControllers — uav_system.hpp:159-169: every controller re-created with its default Params */
static void initialize_controllers(uav_t* u) {
  u->mixer_p.desaturation = 1; /* mixer.hpp:16 */
  mixer_calculate_allocation(u);
  u->rate_p = (orc_rate_params_t){4.0, 0.04, 0.0}; /* rate_controller.hpp:16-18 */
  rate_init_pids(u);
  u->att_p = (orc_attitude_params_t){6.0, 0.05, 0.01, 10.0, 1.0}; /* attitude_controller.hpp:16-20 */
  att_init_pids(u);
  u->vel_p = (orc_velocity_params_t){2.0, 0.05, 0.01, 4.0}; /* velocity_controller.hpp:16-19 */
  vel_init_pids(u);
  u->pos_p = (orc_position_params_t){2.0, 0.15, 0.2, 6.0}; /* position_controller.hpp:16-19 */
  pos_init_pids(u);
}

/* ------------------------------------------------------------------ */
/* MultirotorModel                                                      */
/* ------------------------------------------------------------------ */

static void update_internal_state(uav_t* u) { /* multirotor_model.hpp:204-214 */
  for (int i = 0; i < 3; i++) {
    u->y[0 + i]  = u->x[i];
    u->y[3 + i]  = u->v[i];
    u->y[6 + i]  = M3(u->R, i, 0);
    u->y[9 + i]  = M3(u->R, i, 1);
    u->y[12 + i] = M3(u->R, i, 2);
    u->y[15 + i] = u->omega[i];
  }
}

static void initialize_state(uav_t* u) { /* multirotor_model.hpp:183-198 */
  memset(u->x, 0, sizeof u->x);
  memset(u->v, 0, sizeof u->v);
  memset(u->v_prev, 0, sizeof u->v_prev);
  memset(u->R, 0, sizeof u->R);
  u->R[0] = u->R[4] = u->R[8] = 1.0;
  memset(u->omega, 0, sizeof u->omega);
  memset(u->imu, 0, sizeof u->imu);
  memset(u->motor_rpm, 0, sizeof u->motor_rpm);
  memset(u->input, 0, sizeof u->input);
  memset(u->ext_force, 0, sizeof u->ext_force);
  memset(u->ext_moment, 0, sizeof u->ext_moment);
}

/* Eigen::AngleAxisd(angle, (0,0,1)).toRotationMatrix() — Eigen/src/Geometry/AngleAxis.h */
static void angle_axis_z(double angle, double R[9]) {
  const double ax[3] = {0, 0, 1};
  double       s = sin(angle), c = cos(angle);
  double       sin_axis[3]  = {s * ax[0], s * ax[1], s * ax[2]};
  double       cos1_axis[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
  double       tmp;
  tmp         = cos1_axis[0] * ax[1];
  M3(R, 0, 1) = tmp - sin_axis[2];
  M3(R, 1, 0) = tmp + sin_axis[2];
  tmp         = cos1_axis[0] * ax[2];
  M3(R, 0, 2) = tmp + sin_axis[1];
  M3(R, 2, 0) = tmp - sin_axis[1];
  tmp         = cos1_axis[1] * ax[2];
  M3(R, 1, 2) = tmp - sin_axis[0];
  M3(R, 2, 1) = tmp + sin_axis[0];
  M3(R, 0, 0) = cos1_axis[0] * ax[0] + c;
  M3(R, 1, 1) = cos1_axis[1] * ax[1] + c;
  M3(R, 2, 2) = cos1_axis[2] * ax[2] + c;
}

static void set_state_pos(uav_t* u, const double pos[3], double heading) { /* multirotor_model.hpp:439-446 */
  memcpy(u->initial_pos, pos, sizeof u->initial_pos);
  memcpy(u->x, pos, sizeof u->x);
  angle_axis_z(-heading, u->R);
  update_internal_state(u);
}

/* MultirotorModel::operator() — multirotor_model.hpp:301-366 */
static void model_rhs(const uav_t* u, const double* x, double* dxdt) {
  const orc_model_params_t* p = &u->p;
  double cx_v[3], cR[9], cw[3];
  for (int i = 0; i < 3; i++) {
    cx_v[i]      = x[3 + i];
    M3(cR, i, 0) = x[6 + i];
    M3(cR, i, 1) = x[9 + i];
    M3(cR, i, 2) = x[12 + i];
    cw[i]        = x[15 + i];
  }
  double R[9];
  orc_llt_reorth(cR, R); /* :314-316 */

  double W[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; /* :323-330 */
  M3(W, 2, 1) = cw[0];
  M3(W, 1, 2) = -cw[0];
  M3(W, 0, 2) = cw[1];
  M3(W, 2, 0) = -cw[1];
  M3(W, 1, 0) = cw[2];
  M3(W, 0, 1) = -cw[2];

  double rpm_sq[ORC_MAX_MOTORS]; /* :332 — uses the member state_.motor_rpm, not the RK stage state */
  for (int m = 0; m < p->n_motors; m++) rpm_sq[m] = u->motor_rpm[m] * u->motor_rpm[m];
  double tt[4]; /* :334 */
  for (int r = 0; r < 4; r++) {
    double s = 0;
    for (int m = 0; m < p->n_motors; m++) s += p->allocation_matrix[r * ORC_MAX_MOTORS + m] * rpm_sq[m];
    tt[r] = s;
  }
  double thrust = tt[3];

  double vn         = vec3_norm(cx_v);
  double resistance = p->air_resistance_coeff * M_PI * (p->arm_length) * (p->arm_length) * vn * vn; /* :337 */
  double vnorm[3]   = {cx_v[0], cx_v[1], cx_v[2]};
  if (vec3_norm(vnorm) != 0) vec3_normalize(vnorm); /* :339-342 */

  const double G[3] = {0, 0, p->g};
  double       v_dot[3];
  for (int i = 0; i < 3; i++) /* :346 */
    v_dot[i] = -G[i] + thrust * M3(R, i, 2) / p->mass + u->ext_force[i] / p->mass - resistance * vnorm[i] / p->mass;

  double R_dot[9];
  mat3_mul(R, W, R_dot); /* :348 */

  double Jinv[9], Jw[3], wxJw[3], rhs3[3], omega_dot[3]; /* :350 */
  orc_inverse3(p->J, Jinv);
  mat3_vec(p->J, cw, Jw);
  vec3_cross(cw, Jw, wxJw);
  for (int i = 0; i < 3; i++) rhs3[i] = tt[i] - wxJw[i] + u->ext_moment[i];
  mat3_vec(Jinv, rhs3, omega_dot);

  for (int i = 0; i < 3; i++) { /* :352-359 */
    dxdt[0 + i]  = cx_v[i];
    dxdt[3 + i]  = v_dot[i];
    dxdt[6 + i]  = M3(R_dot, i, 0);
    dxdt[9 + i]  = M3(R_dot, i, 1);
    dxdt[12 + i] = M3(R_dot, i, 2);
    dxdt[15 + i] = omega_dot[i];
  }
  for (int i = 0; i < 18; i++) /* :361-365 */
    if (isnan(dxdt[i])) dxdt[i] = 0;
}

/* odeint runge_kutta4 single step, in == out — ode/.../stepper/base/explicit_stepper_base.hpp:193-199,
 * stepper/detail/generic_rk_algorithm.hpp:190-216, generic_rk_operations.hpp:30-68,
 * algebra/default_operations.hpp:77-154 (scale_sumN: t1 = a1*t2 + a2*t3 + ... left to right),
 * coefficients stepper/runge_kutta4.hpp:43-95 */
static void rk4_step(const uav_t* u, double* y, double dt) {
  const double a1 = 1.0 / 2.0, a2[2] = {0.0, 1.0 / 2.0}, a3[3] = {0.0, 0.0, 1.0};
  const double b[4] = {1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0};
  double       k1[18], k2[18], k3[18], k4[18], t[18];
  model_rhs(u, y, k1);
  for (int i = 0; i < 18; i++) t[i] = 1.0 * y[i] + (a1 * dt) * k1[i];
  model_rhs(u, t, k2);
  for (int i = 0; i < 18; i++) t[i] = 1.0 * y[i] + (a2[0] * dt) * k1[i] + (a2[1] * dt) * k2[i];
  model_rhs(u, t, k3);
  for (int i = 0; i < 18; i++) t[i] = 1.0 * y[i] + (a3[0] * dt) * k1[i] + (a3[1] * dt) * k2[i] + (a3[2] * dt) * k3[i];
  model_rhs(u, t, k4);
  for (int i = 0; i < 18; i++)
    y[i] = 1.0 * y[i] + (b[0] * dt) * k1[i] + (b[1] * dt) * k2[i] + (b[2] * dt) * k3[i] + (b[3] * dt) * k4[i];
}

/* MultirotorModel::setInput — multirotor_model.hpp:392-410 */
static void model_set_input(uav_t* u, const double* motors) {
  for (int i = 0; i < u->p.n_motors; i++) {
    double val = motors[i];
    if (!isfinite(val)) val = 0;
    if (val < 0.0) {
      val = 0.0;
    } else if (val > 1.0) {
      val = 1.0;
    }
    u->input[i] = u->p.min_rpm + (u->p.max_rpm - u->p.min_rpm) * val;
  }
}

/* MultirotorModel::step — multirotor_model.hpp:220-286 */
static void model_step(uav_t* u, double dt, orc_diag_t* diag) {
  orc_model_params_t* p = &u->p;
  double              save[18];
  memcpy(save, u->y, sizeof save);
  rk4_step(u, u->y, dt);
  for (int i = 0; i < 18; i++) {
    if (isnan(u->y[i])) {
      memcpy(u->y, save, sizeof save);
      if (diag) diag->nan_rollback++;
      break;
    }
  }
  for (int i = 0; i < 3; i++) {
    u->x[i]        = u->y[0 + i];
    u->v[i]        = u->y[3 + i];
    M3(u->R, i, 0) = u->y[6 + i];
    M3(u->R, i, 1) = u->y[9 + i];
    M3(u->R, i, 2) = u->y[12 + i];
    u->omega[i]    = u->y[15 + i];
  }
  double filter_const = exp((-dt) / (p->motor_time_constant)); /* :244 */
  for (int m = 0; m < p->n_motors; m++) u->motor_rpm[m] = filter_const * u->motor_rpm[m] + (1.0 - filter_const) * u->input[m];

  double Rn[9]; /* :249-253 */
  orc_llt_reorth(u->R, Rn);
  memcpy(u->R, Rn, sizeof Rn);

  if (p->ground_enabled) { /* :256-262 */
    if (u->x[2] < p->ground_z && u->v[2] < 0) {
      u->x[2] = p->ground_z;
      memset(u->v, 0, sizeof u->v);
      memset(u->omega, 0, sizeof u->omega);
    }
  }
  if (p->takeoff_patch_enabled) { /* :264-277 */
    const double hover_rpm = sqrt((p->mass * p->g) / (p->n_motors * p->kf));
    if (dyn_mean(u->input, p->n_motors) <= 0.90 * hover_rpm) {
      if (u->x[2] < u->initial_pos[2] && u->v[2] < 0) {
        u->x[2] = u->initial_pos[2];
        memset(u->v, 0, sizeof u->v);
        memset(u->omega, 0, sizeof u->omega);
      }
    } else {
      p->takeoff_patch_enabled = 0;
    }
  }
  double acc[3], Rt[9]; /* :280-281 */
  const double G[3] = {0, 0, p->g};
  for (int i = 0; i < 3; i++) acc[i] = ((u->v[i] - u->v_prev[i]) / dt) + G[i];
  mat3_transpose(u->R, Rt);
  mat3_vec(Rt, acc, u->imu);
  memcpy(u->v_prev, u->v, sizeof u->v);
  update_internal_state(u);
}

/* ------------------------------------------------------------------ */
/* controllers: step                                                    */
/* ------------------------------------------------------------------ */

/* PositionController::getControlSignal — position_controller.hpp:73-86 */
static void position_controller(uav_t* u, double dt) {
  double e[3];
  for (int i = 0; i < 3; i++) e[i] = u->pos[i] - u->x[i];
  for (int i = 0; i < 3; i++) u->vel_h[i] = pid_update(&u->pid_pos[i], e[i], dt);
  u->vel_h_heading = u->pos_heading;
}

/* VelocityController::getControlSignal — velocity_controller.hpp:68-102 (both overloads share the PIDs) */
static void velocity_controller(uav_t* u, const double vref[3], double out[3], double dt) {
  double e[3];
  for (int i = 0; i < 3; i++) e[i] = vref[i] - u->v[i];
  for (int i = 0; i < 3; i++) out[i] = pid_update(&u->pid_vel[i], e[i], dt);
}

/* shared tail of both AccelerationController overloads — acceleration_controller.hpp:89-94,116-119 */
static double acceleration_throttle(const uav_t* u, const double fd[3]) {
  const orc_model_params_t* p = &u->p;
  double col2[3]      = {M3(u->R, 0, 2), M3(u->R, 1, 2), M3(u->R, 2, 2)};
  double thrust_force = vec3_dot(fd, col2);
  return (sqrt(thrust_force / (p->kf * p->n_motors)) - p->min_rpm) / (p->max_rpm - p->min_rpm);
}

/* AccelerationController::getControlSignal(AccelerationHdg) — acceleration_controller.hpp:44-97 */
static void acceleration_controller_hdg(uav_t* u) {
  const orc_model_params_t* p = &u->p;
  const double G[3] = {0, 0, p->g};
  double       fd[3], fdn[3];
  for (int i = 0; i < 3; i++) fd[i] = (u->acc_h[i] + G[i]) * p->mass;
  memcpy(fdn, fd, sizeof fd);
  vec3_normalize(fdn);
  const double bxd[3] = {cos(u->acc_h_heading), sin(u->acc_h_heading), 0.0};

  double proj[9]; /* I - fdn fdn^T (:58) */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M3(proj, i, j) = ((i == j) ? 1.0 : 0.0) - fdn[i] * fdn[j];
  double A[3][2], B[3][2] = {{1, 0}, {0, 1}, {0, 0}}; /* :61-68 */
  for (int i = 0; i < 3; i++) {
    A[i][0] = M3(proj, i, 0);
    A[i][1] = M3(proj, i, 1);
  }
  double BtA[2][2]; /* :71 */
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) BtA[i][j] = (B[0][i] * A[0][j] + B[1][i] * A[1][j]) + B[2][i] * A[2][j];
  double MtM[4], MtM_inv[4], pinv[2][2]; /* :72 */
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) MtM[i * 2 + j] = BtA[0][i] * BtA[0][j] + BtA[1][i] * BtA[1][j];
  orc_inverse_lu(MtM, 2, MtM_inv);
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++) pinv[i][j] = MtM_inv[i * 2 + 0] * BtA[j][0] + MtM_inv[i * 2 + 1] * BtA[j][1];
  double AP[3][2], obl[9]; /* :73 */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 2; j++) AP[i][j] = A[i][0] * pinv[0][j] + A[i][1] * pinv[1][j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M3(obl, i, j) = AP[i][0] * B[j][0] + AP[i][1] * B[j][1];

  double c0[3], c1[3]; /* :75-83 */
  mat3_vec(obl, bxd, c0);
  vec3_normalize(c0);
  vec3_cross(fdn, c0, c1);
  vec3_normalize(c1);
  for (int i = 0; i < 3; i++) {
    M3(u->attitude_R, i, 0) = c0[i];
    M3(u->attitude_R, i, 1) = c1[i];
    M3(u->attitude_R, i, 2) = fdn[i];
  }
  u->attitude_throttle = acceleration_throttle(u, fd);
}

/* AccelerationController::getControlSignal(AccelerationHdgRate) — acceleration_controller.hpp:103-122 */
static void acceleration_controller_hdg_rate(uav_t* u) {
  const orc_model_params_t* p = &u->p;
  const double G[3] = {0, 0, p->g};
  double       fd[3], fdn[3];
  for (int i = 0; i < 3; i++) fd[i] = (u->acc_hr[i] + G[i]) * p->mass;
  memcpy(fdn, fd, sizeof fd);
  vec3_normalize(fdn);
  memcpy(u->tilt, fdn, sizeof fdn);
  u->tilt_heading_rate = u->acc_hr_rate;
  u->tilt_throttle     = acceleration_throttle(u, fd);
}

/* orientation error -> 3 PIDs, shared by both AttitudeController overloads — attitude_controller.hpp:81-97,117-133 */
static void attitude_error_pids(uav_t* u, const double Rd[9], double dt, double rate[3]) {
  double Rdt[9], Rt[9], A[9], Bm[9], E[9], e[3];
  mat3_transpose(Rd, Rdt);
  mat3_transpose(u->R, Rt);
  mat3_mul(Rdt, u->R, A);
  mat3_mul(Rt, Rd, Bm);
  for (int i = 0; i < 9; i++) E[i] = 0.5 * (A[i] - Bm[i]);
  e[0] = (M3(E, 1, 2) - M3(E, 2, 1)) / 2.0;
  e[1] = (M3(E, 2, 0) - M3(E, 0, 2)) / 2.0;
  e[2] = (M3(E, 0, 1) - M3(E, 1, 0)) / 2.0;
  for (int i = 0; i < 3; i++) rate[i] = pid_update(&u->pid_att[i], e[i], dt);
}

/* AttitudeController::getControlSignal(Attitude) — attitude_controller.hpp:79-100 */
static void attitude_controller_att(uav_t* u, double dt) {
  double rate[3];
  attitude_error_pids(u, u->attitude_R, dt, rate);
  u->attitude_rate[0] = rate[0];
  u->attitude_rate[1] = rate[1];
  u->attitude_rate[2] = rate[2];
  u->attitude_rate[3] = u->attitude_throttle;
}

/* attitude_controller.hpp:177-206 */
static double intrinsic_body_rate_to_heading_rate(const double R[9], const double w[3], orc_diag_t* diag) {
  double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  double R_d[9];
  mat3_mul(R, W, R_d);
  double rx = M3(R, 0, 0), ry = M3(R, 1, 0);
  double denom = rx * rx + ry * ry;
  double atan2_d_x = 0, atan2_d_y = 0;
  if (fabs(denom) <= 1e-5) {
    if (diag) diag->hdg_rate_denom_small++;
  } else {
    atan2_d_x = -ry / denom;
    atan2_d_y = rx / denom;
  }
  return atan2_d_x * M3(R_d, 0, 0) + atan2_d_y * M3(R_d, 1, 0);
}

/* attitude_controller.hpp:212-251 */
static double get_yaw_rate_intrinsic(const double R[9], double heading_rate, orc_diag_t* diag) {
  if (fabs(heading_rate) < 1e-3) return 0;
  double heading_vector[3] = {M3(R, 0, 0), M3(R, 1, 0), 0};
  double hr_vec[3] = {0, 0, heading_rate}, ez[3] = {0, 0, 1};
  double orbital_velocity[3], b_orb[3];
  vec3_cross(hr_vec, heading_vector, orbital_velocity);
  vec3_cross(ez, heading_vector, b_orb);
  vec3_normalize(b_orb);
  double P[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) M3(P, i, j) = b_orb[i] * b_orb[j];
  double col1[3] = {M3(R, 0, 1), M3(R, 1, 1), M3(R, 2, 1)}, projected[3];
  mat3_vec(P, col1, projected);
  double orbital_velocity_norm = vec3_norm(orbital_velocity);
  double projected_norm        = vec3_norm(projected);
  if (fabs(projected_norm) < 1e-5) {
    if (diag) diag->projected_norm_small++;
    return 0;
  }
  double d         = vec3_dot(orbital_velocity, projected);
  int    direction = (0.0 < d) - (d < 0.0); /* signum, :153-156 */
  double output_yaw_rate = direction * (orbital_velocity_norm / projected_norm);
  if (!isfinite(output_yaw_rate)) {
    if (diag) diag->yaw_rate_not_finite++;
    return 0;
  }
  return output_yaw_rate;
}

/* AttitudeController::getControlSignal(TiltHdgRate) — attitude_controller.hpp:106-145 */
static void attitude_controller_tilt(uav_t* u, double dt, orc_diag_t* diag) {
  double c2[3] = {u->tilt[0], u->tilt[1], u->tilt[2]}, c1[3], c0[3];
  double Rcol0[3] = {M3(u->R, 0, 0), M3(u->R, 1, 0), M3(u->R, 2, 0)};
  vec3_normalize(c2);
  vec3_cross(c2, Rcol0, c1);
  vec3_normalize(c1);
  vec3_cross(c1, c2, c0);
  vec3_normalize(c0);
  double Rd[9];
  for (int i = 0; i < 3; i++) {
    M3(Rd, i, 0) = c0[i];
    M3(Rd, i, 1) = c1[i];
    M3(Rd, i, 2) = c2[i];
  }
  double rate[3];
  attitude_error_pids(u, Rd, dt, rate);
  double parasitic = intrinsic_body_rate_to_heading_rate(u->R, rate, diag);
  rate[2] += get_yaw_rate_intrinsic(u->R, u->tilt_heading_rate - parasitic, diag);
  u->attitude_rate[0] = rate[0];
  u->attitude_rate[1] = rate[1];
  u->attitude_rate[2] = rate[2];
  u->attitude_rate[3] = u->tilt_throttle;
}

/* RateController::getControlSignal — rate_controller.hpp:67-81 */
static void rate_controller(uav_t* u, double dt) {
  double e[3];
  for (int i = 0; i < 3; i++) e[i] = u->attitude_rate[i] - u->omega[i];
  for (int i = 0; i < 3; i++) u->control_group[i] = pid_update(&u->pid_rate[i], e[i], dt);
  u->control_group[3] = u->attitude_rate[3];
}

/* Mixer::getControlSignal — mixer.hpp:107-144 */
static void mixer(uav_t* u) {
  const int n = u->p.n_motors;
  double    cg[4] = {u->control_group[0], u->control_group[1], u->control_group[2], u->control_group[3]};
  double*   m = u->actuators;
#define MIX()                                                                                   \
  for (int i = 0; i < n; i++) {                                                                 \
    const double* r = &u->alloc_inv[i * 4];                                                     \
    double        s = 0;                                                                        \
    for (int k = 0; k < 4; k++) s += r[k] * cg[k];                                              \
    m[i] = s;                                                                                   \
  }
  MIX();
  if (u->mixer_p.desaturation) {
    double mn = m[0];
    for (int i = 1; i < n; i++)
      if (m[i] < mn) mn = m[i];
    /* mixer.hpp:121 calls UNQUALIFIED abs(min) on a double.  Which overload that is depends on the translation unit: with only
       <cmath> + <cstdlib> visible it is ::abs(int) (the offset truncates to 0); as soon as a libstdc++ C wrapper header
       (<stdlib.h> / <math.h>) has been included it is std::abs(double).  Every TU that can hold this line includes Eigen
       (references.hpp:5), and Eigen/Core includes <emmintrin.h> on x86-64 (SSE2 is on by default), whose <mm_malloc.h> includes
       <stdlib.h>: the double overload — and ros/ros.h does the same for the nodelet.  That is the flavour restated here
       (probe recorded in DESIGN.md §9).  -DORC_MIXER_ABS_INT builds the truncating variant (`make -C oracle absint`), kept so that
       a reference-held fixture, should one ever exist, can settle it; tests/test_oracle_kat.py::test_mixer_desaturation_abs_overload
       fails if the default build ever truncates. */
    if (mn < 0.0) {
#ifdef ORC_MIXER_ABS_INT
      const double off = (double)abs((int)mn);
#else
      const double off = fabs(mn);
#endif
      for (int i = 0; i < n; i++) m[i] += off;
    }
    double mx = m[0];
    for (int i = 1; i < n; i++)
      if (m[i] > mx) mx = m[i];
    if (mx > 1.0) {
      if (u->control_group[3] > 1e-2) {
        for (int i = 0; i < 3; i++) cg[i] = cg[i] / (dyn_mean(m, n) / u->control_group[3]);
        MIX();
      } else {
        for (int i = 0; i < n; i++) m[i] /= mx;
      }
    }
  }
#undef MIX
}

/* UavSystem::makeStep — uav_system.hpp:304-380 */
static void uav_make_step(uav_t* u, double dt, orc_diag_t* diag) {
  int active_input = u->active_input;
  if (u->crashed || u->active_input == ORC_INPUT_UNKNOWN) {
    for (int i = 0; i < u->p.n_motors; i++) u->actuators[i] = 0.0;
  } else {
    if (active_input == ORC_POSITION_CMD) {
      position_controller(u, dt);
      active_input = ORC_VELOCITY_HDG_CMD;
      if (u->has_ff[ORC_FF_VELOCITY_HDG]) {
        for (int i = 0; i < 3; i++) u->vel_h[i] += u->ff_vec[ORC_FF_VELOCITY_HDG][i];
      } else if (u->has_ff[ORC_FF_VELOCITY_HDG_RATE]) {
        for (int i = 0; i < 3; i++) u->vel_h[i] += u->ff_vec[ORC_FF_VELOCITY_HDG_RATE][i];
      }
    }
    if (active_input == ORC_VELOCITY_HDG_CMD) {
      velocity_controller(u, u->vel_h, u->acc_h, dt);
      u->acc_h_heading = u->vel_h_heading;
      active_input     = ORC_ACCELERATION_HDG_CMD;
      if (u->has_ff[ORC_FF_ACCELERATION_HDG]) {
        for (int i = 0; i < 3; i++) u->acc_h[i] += u->ff_vec[ORC_FF_ACCELERATION_HDG][i];
      } else if (u->has_ff[ORC_FF_ACCELERATION_HDG_RATE]) {
        for (int i = 0; i < 3; i++) u->acc_h[i] += u->ff_vec[ORC_FF_ACCELERATION_HDG_RATE][i];
      }
    } else if (active_input == ORC_VELOCITY_HDG_RATE_CMD) {
      velocity_controller(u, u->vel_hr, u->acc_hr, dt);
      u->acc_hr_rate = u->vel_hr_rate;
      active_input   = ORC_ACCELERATION_HDG_RATE_CMD;
      if (u->has_ff[ORC_FF_ACCELERATION_HDG_RATE]) {
        for (int i = 0; i < 3; i++) u->acc_hr[i] += u->ff_vec[ORC_FF_ACCELERATION_HDG_RATE][i];
        u->acc_hr_rate += u->ff_scalar[ORC_FF_ACCELERATION_HDG_RATE];
      } else if (u->has_ff[ORC_FF_ACCELERATION_HDG]) {
        for (int i = 0; i < 3; i++) u->acc_hr[i] += u->ff_vec[ORC_FF_ACCELERATION_HDG][i];
      }
    }
    if (active_input == ORC_ACCELERATION_HDG_CMD) {
      acceleration_controller_hdg(u);
      active_input = ORC_ATTITUDE_CMD;
    } else if (active_input == ORC_ACCELERATION_HDG_RATE_CMD) {
      acceleration_controller_hdg_rate(u);
      active_input = ORC_TILT_HDG_RATE_CMD;
    }
    if (active_input == ORC_ATTITUDE_CMD) {
      attitude_controller_att(u, dt);
      active_input = ORC_ATTITUDE_RATE_CMD;
    } else if (active_input == ORC_TILT_HDG_RATE_CMD) {
      attitude_controller_tilt(u, dt, diag);
      active_input = ORC_ATTITUDE_RATE_CMD;
    }
    if (active_input == ORC_ATTITUDE_RATE_CMD) {
      rate_controller(u, dt);
      active_input = ORC_CONTROL_GROUP_CMD;
    }
    if (active_input == ORC_CONTROL_GROUP_CMD) {
      mixer(u);
      active_input = ORC_ACTUATOR_CMD;
    }
  }
  model_set_input(u, u->actuators);
  model_step(u, dt, diag);
}

/* One component of the path on UAV state — the counterpart of mrs_swarm_debug_component (include/mrs_swarm.h: same component
 * ids, same row layouts, matrices row-major).  Each case sets the inputs the reference's function reads, calls exactly that
 * function of this file, and returns what it wrote. */
static void debug_component_one(uav_t* u, int comp, const double* a, double* o, double dt, orc_diag_t* diag) {
  switch (comp) {
    case 1: orc_llt_reorth(a, o); break; /* R * inverse(matrixL(LLT(R^T R))) */
    case 2: {                            /* MultirotorModel::operator(): internal order x v Rcol0 Rcol1 Rcol2 w (:204-214) */
      double y[18], k[18];
      for (int i = 0; i < 3; i++) {
        y[i] = a[i];
        y[3 + i] = a[3 + i];
        y[6 + i] = a[6 + 3 * i + 0];
        y[9 + i] = a[6 + 3 * i + 1];
        y[12 + i] = a[6 + 3 * i + 2];
        y[15 + i] = a[15 + i];
      }
      model_rhs(u, y, k);
      for (int i = 0; i < 3; i++) {
        o[i] = k[i];
        o[3 + i] = k[3 + i];
        o[6 + 3 * i + 0] = k[6 + i];
        o[6 + 3 * i + 1] = k[9 + i];
        o[6 + 3 * i + 2] = k[12 + i];
        o[15 + i] = k[15 + i];
      }
    } break;
    case 3:
      memcpy(u->control_group, a, 4 * sizeof(double));
      mixer(u);
      for (int m = 0; m < ORC_MAX_MOTORS; m++) o[m] = m < u->p.n_motors ? u->actuators[m] : 0.0;
      break;
    case 4:
      memcpy(u->pos, a, 3 * sizeof(double));
      position_controller(u, dt);
      memcpy(o, u->vel_h, 3 * sizeof(double));
      break;
    case 5: velocity_controller(u, a, o, dt); break;
    case 6:
      memcpy(u->acc_h, a, 3 * sizeof(double));
      u->acc_h_heading = a[3];
      acceleration_controller_hdg(u);
      memcpy(o, u->attitude_R, 9 * sizeof(double));
      o[9] = u->attitude_throttle;
      break;
    case 7:
      memcpy(u->acc_hr, a, 3 * sizeof(double));
      u->acc_hr_rate = a[3];
      acceleration_controller_hdg_rate(u);
      memcpy(o, u->tilt, 3 * sizeof(double));
      o[3] = u->tilt_heading_rate;
      o[4] = u->tilt_throttle;
      break;
    case 8:
      memcpy(u->attitude_R, a, 9 * sizeof(double));
      u->attitude_throttle = a[9];
      attitude_controller_att(u, dt);
      memcpy(o, u->attitude_rate, 4 * sizeof(double));
      break;
    case 9:
      memcpy(u->tilt, a, 3 * sizeof(double));
      u->tilt_heading_rate = a[3];
      u->tilt_throttle     = a[4];
      attitude_controller_tilt(u, dt, diag);
      memcpy(o, u->attitude_rate, 4 * sizeof(double));
      break;
    case 10:
      memcpy(u->attitude_rate, a, 4 * sizeof(double));
      rate_controller(u, dt);
      memcpy(o, u->control_group, 4 * sizeof(double));
      break;
    default: break;
  }
}

void orc_swarm_debug_component(orc_swarm_t* s, int32_t component, int32_t first, int32_t count, const double* in, int32_t in_stride, double* out,
                               int32_t out_stride, double dt) {
  for (int k = 0; k < count; k++)
    debug_component_one(&s->u[first + k], component, in + (size_t)k * in_stride, out + (size_t)k * out_stride, dt, &s->diag);
}

/* ------------------------------------------------------------------ */
/* swarm API                                                            */
/* ------------------------------------------------------------------ */

orc_swarm_t* orc_swarm_create(int32_t n) {
  orc_swarm_t* s = (orc_swarm_t*)calloc(1, sizeof *s);
  s->n           = n;
  s->u           = (uav_t*)calloc((size_t)(n > 0 ? n : 1), sizeof(uav_t));
  orc_model_params_t def;
  orc_model_params_default(&def);
  for (int i = 0; i < n; i++) {
    s->u[i].p = def;
    initialize_state(&s->u[i]);
  }
  if (n > 0) orc_swarm_construct(s, 0, n, NULL, NULL, NULL);
  return s;
}

void orc_swarm_destroy(orc_swarm_t* s) {
  if (!s) return;
  free(s->u);
  free(s);
}

int32_t orc_swarm_size(const orc_swarm_t* s) { return s->n; }

void orc_swarm_construct(orc_swarm_t* s, int32_t first, int32_t count, const orc_model_params_t* params, const double* pos,
                         const double* heading) {
  for (int k = 0; k < count; k++) {
    uav_t* u = &s->u[first + k];
    memset(u, 0, sizeof *u);
    if (params)
      u->p = *params; /* uav_system.hpp:137,146 */
    else
      orc_model_params_default(&u->p); /* :127-132 */
    initialize_state(u);
    update_internal_state(u);
    if (pos) set_state_pos(u, &pos[3 * k], heading ? heading[k] : 0.0); /* :150 */
    u->crashed      = 0;
    u->active_input = ORC_INPUT_UNKNOWN;
    u->attitude_R[0] = u->attitude_R[4] = u->attitude_R[8] = 1.0; /* references.hpp:102-104 */
    u->tilt[0]                                            = 1.0; /* Vector3d::Identity(), references.hpp:123 */
    initialize_controllers(u);
  }
}

void orc_swarm_set_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_model_params_t* params) {
  for (int k = 0; k < count; k++) {
    uav_t* u = &s->u[first + k];
    u->p     = *params;
    initialize_controllers(u);
  }
}

void orc_swarm_get_params(const orc_swarm_t* s, int32_t uav, orc_model_params_t* out) { *out = s->u[uav].p; }

void orc_swarm_set_mixer_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_mixer_params_t* p) {
  for (int k = 0; k < count; k++) {
    uav_t* u   = &s->u[first + k];
    u->mixer_p = *p;
    mixer_calculate_allocation(u); /* mixer.hpp:61-66 */
  }
}
void orc_swarm_set_rate_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_rate_params_t* p) {
  for (int k = 0; k < count; k++) {
    s->u[first + k].rate_p = *p;
    rate_init_pids(&s->u[first + k]);
  }
}
void orc_swarm_set_attitude_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_attitude_params_t* p) {
  for (int k = 0; k < count; k++) {
    s->u[first + k].att_p = *p;
    att_init_pids(&s->u[first + k]);
  }
}
void orc_swarm_set_velocity_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_velocity_params_t* p) {
  for (int k = 0; k < count; k++) {
    s->u[first + k].vel_p = *p;
    vel_init_pids(&s->u[first + k]);
  }
}
void orc_swarm_set_position_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_position_params_t* p) {
  for (int k = 0; k < count; k++) {
    s->u[first + k].pos_p = *p;
    pos_init_pids(&s->u[first + k]);
  }
}

void orc_swarm_set_input(orc_swarm_t* s, int32_t first, int32_t count, int32_t mode, const double* payload, int32_t stride) {
  for (int k = 0; k < count; k++) {
    uav_t*        u = &s->u[first + k];
    const double* q = payload ? &payload[(size_t)k * stride] : NULL;
    switch (mode) {
      case ORC_ACTUATOR_CMD:
        for (int i = 0; i < u->p.n_motors; i++) u->actuators[i] = q[i];
        break;
      case ORC_CONTROL_GROUP_CMD: memcpy(u->control_group, q, 4 * sizeof(double)); break;
      case ORC_ATTITUDE_RATE_CMD: memcpy(u->attitude_rate, q, 4 * sizeof(double)); break;
      case ORC_ATTITUDE_CMD:
        memcpy(u->attitude_R, q, 9 * sizeof(double));
        u->attitude_throttle = q[9];
        break;
      case ORC_TILT_HDG_RATE_CMD:
        memcpy(u->tilt, q, 3 * sizeof(double));
        u->tilt_heading_rate = q[3];
        u->tilt_throttle     = q[4];
        break;
      case ORC_ACCELERATION_HDG_RATE_CMD:
        memcpy(u->acc_hr, q, 3 * sizeof(double));
        u->acc_hr_rate = q[3];
        break;
      case ORC_ACCELERATION_HDG_CMD:
        memcpy(u->acc_h, q, 3 * sizeof(double));
        u->acc_h_heading = q[3];
        break;
      case ORC_VELOCITY_HDG_RATE_CMD:
        memcpy(u->vel_hr, q, 3 * sizeof(double));
        u->vel_hr_rate = q[3];
        break;
      case ORC_VELOCITY_HDG_CMD:
        memcpy(u->vel_h, q, 3 * sizeof(double));
        u->vel_h_heading = q[3];
        break;
      case ORC_POSITION_CMD:
        memcpy(u->pos, q, 3 * sizeof(double));
        u->pos_heading = q[3];
        break;
      default: break;
    }
    u->active_input = mode;
  }
}

void orc_swarm_set_feedforward(orc_swarm_t* s, int32_t first, int32_t count, int32_t kind, const double* payload, int32_t stride) {
  for (int k = 0; k < count; k++) {
    uav_t*        u = &s->u[first + k];
    const double* q = &payload[(size_t)k * stride];
    u->has_ff[kind] = 1;
    memcpy(u->ff_vec[kind], q, 3 * sizeof(double));
    u->ff_scalar[kind] = q[3];
  }
}

void orc_swarm_step(orc_swarm_t* s, double dt) {
  for (int i = 0; i < s->n; i++)
    if (!s->u[i].hold) uav_make_step(&s->u[i], dt, &s->diag); /* src/uav_system_ros.cpp:265-271 */
}

typedef struct {
  orc_swarm_t* s;
  int          lo, hi, n_steps;
  double       dt;
  orc_diag_t   diag;
} step_job_t;

static void* step_worker(void* arg) {
  step_job_t* j = (step_job_t*)arg;
  for (int k = 0; k < j->n_steps; k++)
    for (int i = j->lo; i < j->hi; i++)
      if (!j->s->u[i].hold) uav_make_step(&j->s->u[i], j->dt, &j->diag);
  return NULL;
}

void orc_swarm_step_n(orc_swarm_t* s, double dt, int32_t n_steps, int32_t n_threads) {
  if (n_threads <= 1) {
    for (int k = 0; k < n_steps; k++) orc_swarm_step(s, dt);
    return;
  }
  if (n_threads > 256) n_threads = 256;
  pthread_t  th[256];
  step_job_t job[256];
  for (int t = 0; t < n_threads; t++) {
    memset(&job[t], 0, sizeof job[t]);
    job[t].s       = s;
    job[t].lo      = (int)((long long)s->n * t / n_threads);
    job[t].hi      = (int)((long long)s->n * (t + 1) / n_threads);
    job[t].n_steps = n_steps;
    job[t].dt      = dt;
    pthread_create(&th[t], NULL, step_worker, &job[t]);
  }
  for (int t = 0; t < n_threads; t++) {
    pthread_join(th[t], NULL);
    s->diag.hdg_rate_denom_small += job[t].diag.hdg_rate_denom_small;
    s->diag.projected_norm_small += job[t].diag.projected_norm_small;
    s->diag.yaw_rate_not_finite += job[t].diag.yaw_rate_not_finite;
    s->diag.nan_rollback += job[t].diag.nan_rollback;
  }
}

/* ------------------------------------------------------------------ */
/* handleCollisions — /root/reference/src/multirotor_simulator.cpp:295-359 */
/* ------------------------------------------------------------------ */

typedef struct {
  int64_t key;
  int32_t idx;
} cell_entry_t;

static int cell_cmp(const void* a, const void* b) {
  const cell_entry_t *x = (const cell_entry_t*)a, *y = (const cell_entry_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx);
}

#define CELL_BITS 21
#define CELL_OFF (1 << (CELL_BITS - 1))
static int64_t cell_key(int64_t cx, int64_t cy, int64_t cz) {
  return ((cx + CELL_OFF) << (2 * CELL_BITS)) | ((cy + CELL_OFF) << CELL_BITS) | (cz + CELL_OFF);
}

static int int_cmp(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

void orc_swarm_handle_collisions(orc_swarm_t* s, int32_t enabled, int32_t crash, double rebounce) {
  if (!(crash || enabled)) return; /* :299-301 */
  const int n = s->n;
  if (n == 0) return;
  const double radius = 3.0;        /* :326 — compared against the SQUARED distance (nanoflann.hpp:273-305) */
  const double cell   = 1.75;       /* > sqrt(3): every pair with d^2 < 3 lies in adjacent cells */
  cell_entry_t* ent   = (cell_entry_t*)malloc(sizeof(cell_entry_t) * (size_t)n);
  int           all_finite = 1;
  for (int i = 0; i < n; i++) {
    const double* x = s->u[i].x;
    if (!isfinite(x[0]) || !isfinite(x[1]) || !isfinite(x[2]) || fabs(x[0]) > 1e6 || fabs(x[1]) > 1e6 || fabs(x[2]) > 1e6) all_finite = 0;
  }
  if (all_finite) {
    for (int i = 0; i < n; i++) {
      const double* x = s->u[i].x;
      ent[i].key      = cell_key((int64_t)floor(x[0] / cell), (int64_t)floor(x[1] / cell), (int64_t)floor(x[2] / cell));
      ent[i].idx      = i;
    }
    qsort(ent, (size_t)n, sizeof *ent, cell_cmp);
  }
  double* forces = (double*)calloc((size_t)n * 3, sizeof(double)); /* :315-319 */
  int*    cand   = (int*)malloc(sizeof(int) * (size_t)n);

  for (int i = 0; i < n; i++) { /* :321 */
    const uav_t* u1 = &s->u[i];
    int          nc = 0;
    if (all_finite) {
      int64_t cx = (int64_t)floor(u1->x[0] / cell), cy = (int64_t)floor(u1->x[1] / cell), cz = (int64_t)floor(u1->x[2] / cell);
      for (int64_t dx = -1; dx <= 1; dx++)
        for (int64_t dy = -1; dy <= 1; dy++)
          for (int64_t dz = -1; dz <= 1; dz++) {
            int64_t key = cell_key(cx + dx, cy + dy, cz + dz);
            int     lo = 0, hi = n;
            while (lo < hi) {
              int mid = (lo + hi) / 2;
              if (ent[mid].key < key)
                lo = mid + 1;
              else
                hi = mid;
            }
            for (; lo < n && ent[lo].key == key; lo++) cand[nc++] = ent[lo].idx;
          }
      qsort(cand, (size_t)nc, sizeof(int), int_cmp); /* deterministic accumulation order: ascending index */
    } else {
      for (int j = 0; j < n; j++) cand[nc++] = j;
    }
    for (int c = 0; c < nc; c++) {
      const int    idx = cand[c];
      const uav_t* u2  = &s->u[idx];
      /* nanoflann L2_Adaptor::evalMetric, size 3: result = ((0 + d0^2) + d1^2) + d2^2, diff = query - point (nanoflann.hpp:443-486) */
      double dist = 0;
      for (int d = 0; d < 3; d++) {
        double diff0 = u1->x[d] - u2->x[d];
        dist += diff0 * diff0;
      }
      if (!(dist < radius)) continue; /* RadiusResultSet::addPoint */
      if (idx == i) continue;         /* :335 */
      const double crit_dist = u1->p.arm_length + u1->p.prop_radius + u2->p.arm_length + u2->p.prop_radius; /* :342 */
      double rel_pos[3] = {u1->x[0] - u2->x[0], u1->x[1] - u2->x[1], u1->x[2] - u2->x[2]};
      if (dist < crit_dist) { /* :346 — squared distance vs un-squared threshold, reproduced */
        if (crash) {
          s->u[idx].crashed = 1; /* :348 */
        } else {
          vec3_normalize(rel_pos); /* :350 */
          for (int d = 0; d < 3; d++)
            forces[3 * i + d] += rebounce * rel_pos[d] * u1->p.mass * (u2->p.mass / (u1->p.mass + u2->p.mass));
        }
      }
    }
  }
  for (int i = 0; i < n; i++) memcpy(s->u[i].ext_force, &forces[3 * i], 3 * sizeof(double)); /* :356-358 */
  free(cand);
  free(forces);
  free(ent);
}

/* ------------------------------------------------------------------ */
/* accessors                                                            */
/* ------------------------------------------------------------------ */

void orc_swarm_apply_force(orc_swarm_t* s, int32_t first, int32_t count, const double* force) {
  for (int k = 0; k < count; k++) memcpy(s->u[first + k].ext_force, &force[3 * k], 3 * sizeof(double));
}
void orc_swarm_set_hold(orc_swarm_t* s, int32_t first, int32_t count, int32_t hold) {
  for (int k = 0; k < count; k++) s->u[first + k].hold = hold != 0;
}
void orc_swarm_crash(orc_swarm_t* s, int32_t first, int32_t count) {
  for (int k = 0; k < count; k++) s->u[first + k].crashed = 1;
}
void orc_swarm_has_crashed(const orc_swarm_t* s, int32_t first, int32_t count, int32_t* out) {
  for (int k = 0; k < count; k++) out[k] = s->u[first + k].crashed;
}

void orc_swarm_get_state(const orc_swarm_t* s, int32_t first, int32_t count, double* x, double* v, double* v_prev, double* R,
                         double* omega, double* motor_rpm) {
  for (int k = 0; k < count; k++) {
    const uav_t* u = &s->u[first + k];
    if (x) memcpy(&x[3 * k], u->x, 3 * sizeof(double));
    if (v) memcpy(&v[3 * k], u->v, 3 * sizeof(double));
    if (v_prev) memcpy(&v_prev[3 * k], u->v_prev, 3 * sizeof(double));
    if (R) memcpy(&R[9 * k], u->R, 9 * sizeof(double));
    if (omega) memcpy(&omega[3 * k], u->omega, 3 * sizeof(double));
    if (motor_rpm) memcpy(&motor_rpm[ORC_MAX_MOTORS * k], u->motor_rpm, ORC_MAX_MOTORS * sizeof(double));
  }
}

void orc_swarm_set_state(orc_swarm_t* s, int32_t first, int32_t count, const double* x, const double* v, const double* R,
                         const double* omega, const double* motor_rpm) { /* multirotor_model.hpp:424-433 */
  for (int k = 0; k < count; k++) {
    uav_t* u = &s->u[first + k];
    if (x) memcpy(u->x, &x[3 * k], 3 * sizeof(double));
    if (v) memcpy(u->v, &v[3 * k], 3 * sizeof(double));
    if (R) memcpy(u->R, &R[9 * k], 9 * sizeof(double));
    if (omega) memcpy(u->omega, &omega[3 * k], 3 * sizeof(double));
    if (motor_rpm)
      for (int m = 0; m < u->p.n_motors; m++) u->motor_rpm[m] = motor_rpm[ORC_MAX_MOTORS * k + m];
    update_internal_state(u);
  }
}

void orc_swarm_get_imu(const orc_swarm_t* s, int32_t first, int32_t count, double* imu) {
  for (int k = 0; k < count; k++) memcpy(&imu[3 * k], s->u[first + k].imu, 3 * sizeof(double));
}
void orc_swarm_get_external_force(const orc_swarm_t* s, int32_t first, int32_t count, double* f) {
  for (int k = 0; k < count; k++) memcpy(&f[3 * k], s->u[first + k].ext_force, 3 * sizeof(double));
}
void orc_swarm_get_pid(const orc_swarm_t* s, int32_t first, int32_t count, double* pid) {
  for (int k = 0; k < count; k++) {
    const uav_t*       u       = &s->u[first + k];
    const pid_state_t* sets[4] = {u->pid_pos, u->pid_vel, u->pid_att, u->pid_rate};
    for (int c = 0; c < 4; c++)
      for (int a = 0; a < 3; a++) {
        pid[24 * k + c * 6 + a * 2 + 0] = sets[c][a].last_error;
        pid[24 * k + c * 6 + a * 2 + 1] = sets[c][a].integral;
      }
  }
}
/* test hook (the reference has no accessor for the controllers' PID members, pid.hpp:20-21): puts a recorded PID state back, so
 * that a test can restart both sides from IDENTICAL inputs in the middle of a closed-loop run */
void orc_swarm_set_pid(orc_swarm_t* s, int32_t first, int32_t count, const double* pid) {
  for (int k = 0; k < count; k++) {
    uav_t*       u       = &s->u[first + k];
    pid_state_t* sets[4] = {u->pid_pos, u->pid_vel, u->pid_att, u->pid_rate};
    for (int c = 0; c < 4; c++)
      for (int a = 0; a < 3; a++) {
        sets[c][a].last_error = pid[24 * k + c * 6 + a * 2 + 0];
        sets[c][a].integral   = pid[24 * k + c * 6 + a * 2 + 1];
      }
  }
}
void orc_swarm_get_mixer_allocation(const orc_swarm_t* s, int32_t uav, double* out) {
  memcpy(out, s->u[uav].alloc_inv, sizeof(double) * 4 * (size_t)s->u[uav].p.n_motors);
}
void orc_swarm_get_diag(const orc_swarm_t* s, orc_diag_t* out) { *out = s->diag; }

/* Eigen::Quaterniond(Matrix3d) — Eigen/src/Geometry/Quaternion.h quaternionbase_assign_impl<Other,3,3>::run.
 * q = {x, y, z, w} */
static void quat_from_matrix(const double m[9], double q[4]) {
  double t = (M3(m, 0, 0) + M3(m, 1, 1)) + M3(m, 2, 2);
  if (t > 0) {
    t    = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t    = 0.5 / t;
    q[0] = (M3(m, 2, 1) - M3(m, 1, 2)) * t;
    q[1] = (M3(m, 0, 2) - M3(m, 2, 0)) * t;
    q[2] = (M3(m, 1, 0) - M3(m, 0, 1)) * t;
  } else {
    int i = 0;
    if (M3(m, 1, 1) > M3(m, 0, 0)) i = 1;
    if (M3(m, 2, 2) > M3(m, i, i)) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t    = sqrt(M3(m, i, i) - M3(m, j, j) - M3(m, k, k) + 1.0);
    q[i] = 0.5 * t;
    t    = 0.5 / t;
    q[3] = (M3(m, k, j) - M3(m, j, k)) * t;
    q[j] = (M3(m, j, i) + M3(m, i, j)) * t;
    q[k] = (M3(m, k, i) + M3(m, i, k)) * t;
  }
}

void orc_swarm_get_outputs(const orc_swarm_t* s, int32_t first, int32_t count, orc_uav_output_t* out) {
  for (int n = 0; n < count; n++) {
    const uav_t*      u = &s->u[first + n];
    orc_uav_output_t* o = &out[n];
    double            Rt[9];
    memcpy(o->position, u->x, sizeof u->x);             /* src/uav_system_ros.cpp:352-354 */
    quat_from_matrix(u->R, o->orientation);             /* :350 */
    mat3_transpose(u->R, Rt);
    mat3_vec(Rt, u->v, o->velocity_body);               /* :356 */
    memcpy(o->angular_velocity, u->omega, sizeof u->omega);
    memcpy(o->linear_acceleration, u->imu, sizeof u->imu);
    /* publishRangefinder, :403-419 */
    const double body_z[3]          = {M3(u->R, 0, 2), M3(u->R, 1, 2), M3(u->R, 2, 2)};
    const double rangefinder_dir[3] = {-body_z[0], -body_z[1], -body_z[2]};
    const double down[3]            = {0, 0, -1};
    double       tilt = acos(vec3_dot(rangefinder_dir, down));
    double       range;
    if (body_z[2] > 0) {
      range = (u->x[2] - u->p.ground_z) / cos(tilt) + 0.01;
    } else {
      range = 1.7976931348623157e308;
    }
    if (range > 40.0) range = 41.0;
    o->range = range;
  }
}

/* mrs_lib::AttitudeConverter(R).getHeading() (ctu-mrs/mrs_lib src/attitude_converter/attitude_converter.cpp): the stored
 * quaternion is Eigen::Quaterniond(R); the heading is atan2 of the rotated body-x axis, obtained through
 * tf2::Transform(q) * (1,0,0), i.e. the first column of tf2::Matrix3x3::setRotation(q). */
static double heading_of(const double R[9]) {
  double q[4];
  quat_from_matrix(R, q);
  const double d  = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]; /* tf2 Quaternion::length2 */
  const double sc = 2.0 / d;
  const double ys = q[1] * sc, zs = q[2] * sc;
  const double wz = q[3] * zs, xy = q[0] * ys, yy = q[1] * ys, zz = q[2] * zs;
  const double m00 = 1.0 - (yy + zz), m10 = xy + wz;
  return atan2(m10, m00);
}

/* mrs_lib::AttitudeConverter(0, 0, heading) -> Eigen::Matrix3d: tf2::Quaternion::setRPY then Eigen's
 * Quaternion::toRotationMatrix (Eigen/src/Geometry/Quaternion.h) */
static void attitude_from_heading(double heading, double R[9]) {
  const double hy = heading * 0.5, hp = 0.0 * 0.5, hr = 0.0 * 0.5;
  const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
  const double x = sr * cp * cy - cr * sp * sy, y = cr * sp * cy + sr * cp * sy, z = cr * cp * sy - sr * sp * cy,
               w = cr * cp * cy + sr * sp * sy;
  const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y,
               tzz = tz * z;
  M3(R, 0, 0) = 1.0 - (tyy + tzz);
  M3(R, 0, 1) = txy - twz;
  M3(R, 0, 2) = txz + twy;
  M3(R, 1, 0) = txy + twz;
  M3(R, 1, 1) = 1.0 - (txx + tzz);
  M3(R, 1, 2) = tyz - twx;
  M3(R, 2, 0) = txz - twy;
  M3(R, 2, 1) = tyz + twx;
  M3(R, 2, 2) = 1.0 - (txx + tyy);
}

void orc_swarm_timeout_input(orc_swarm_t* s, int32_t first, int32_t count) {
  for (int n = 0; n < count; n++) {
    uav_t* u = &s->u[first + n];
    switch (u->active_input) { /* last_input_mode_ == the mode of the last received command */
      case ORC_POSITION_CMD: /* :482-495 */
        memcpy(u->pos, u->x, sizeof u->x);
        u->pos_heading = heading_of(u->R);
        break;
      case ORC_VELOCITY_HDG_CMD: /* :497-510 */
        memset(u->vel_h, 0, sizeof u->vel_h);
        u->vel_h_heading = heading_of(u->R);
        break;
      case ORC_VELOCITY_HDG_RATE_CMD: /* :512-525 */
        memset(u->vel_hr, 0, sizeof u->vel_hr);
        u->vel_hr_rate = 0;
        break;
      case ORC_ACCELERATION_HDG_CMD: /* :527-540 */
        memset(u->acc_h, 0, sizeof u->acc_h);
        u->acc_h_heading = heading_of(u->R);
        break;
      case ORC_ACCELERATION_HDG_RATE_CMD: /* :542-555 */
        memset(u->acc_hr, 0, sizeof u->acc_hr);
        u->acc_hr_rate = 0;
        break;
      case ORC_ATTITUDE_CMD: /* :557-572 */
        attitude_from_heading(heading_of(u->R), u->attitude_R);
        u->attitude_throttle = 0.0;
        break;
      case ORC_TILT_HDG_RATE_CMD: /* :574-587 (heading_rate keeps its default 0) */
        u->tilt[0] = 0; u->tilt[1] = 0; u->tilt[2] = 1;
        u->tilt_heading_rate = 0;
        u->tilt_throttle     = 0.0;
        break;
      case ORC_ATTITUDE_RATE_CMD: memset(u->attitude_rate, 0, sizeof u->attitude_rate); break; /* :589-604 */
      case ORC_CONTROL_GROUP_CMD: memset(u->control_group, 0, sizeof u->control_group); break; /* :606-621 */
      case ORC_ACTUATOR_CMD: memset(u->actuators, 0, sizeof u->actuators); break;             /* :623-635 */
      default: u->active_input = ORC_INPUT_UNKNOWN; break;                                     /* :637-645 */
    }
  }
}

void orc_swarm_set_mass(orc_swarm_t* s, int32_t first, int32_t count, double mass) { /* src/uav_system_ros.cpp:1036-1047 */
  for (int n = 0; n < count; n++) {
    uav_t*             u = &s->u[first + n];
    orc_model_params_t p = u->p; /* getParams(): includes the mutated take-off flag */
    const double original_mass = p.mass;
    p.mass = mass;
    for (int m = 0; m < p.n_motors; m++) p.allocation_matrix[2 * ORC_MAX_MOTORS + m] = p.mass * (p.allocation_matrix[2 * ORC_MAX_MOTORS + m] / original_mass);
    orc_calculate_inertia(&p);
    u->p = p;
    initialize_controllers(u);
  }
}

void orc_swarm_set_ground_z(orc_swarm_t* s, int32_t first, int32_t count, double ground_z) { /* :1063-1073 */
  for (int n = 0; n < count; n++) {
    uav_t* u      = &s->u[first + n];
    u->p.ground_z = ground_z;
    initialize_controllers(u);
  }
}
