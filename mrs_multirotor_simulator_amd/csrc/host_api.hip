// host_api.hip — the C ABI of include/mrs_swarm.h except the hot path: parameter helpers, lifetime, construction, commands,
// state access, publisher payloads, probes.  Owns the device SoA state of one swarm shard, the interned per-airframe type table
// and the bookkeeping that turns the reference's per-UAV setters into column uploads.
// No CPU fallback exists: without a usable HIP device mrs_swarm_create fails.
#include "host_internal.h"

namespace mrs_host {
static thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
}  // namespace mrs_host

// ------------------------------------------------------------------------------------------------
// host-side derivations (init-time arithmetic of the reference, restated)
// ------------------------------------------------------------------------------------------------

// Eigen fixed-size 3x3 inverse(): cofactors / determinant (Eigen/src/LU/InverseImpl.h, size 3)
static void inverse3_cofactor(const double m[9], double r[9]) {
  auto cof = [&](int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
  };
  const double c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
  const double det    = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
  const double invdet = 1.0 / det;
  r[0] = c0 * invdet;
  r[1] = c1 * invdet;
  r[2] = c2 * invdet;
  r[3] = cof(0, 1) * invdet;
  r[4] = cof(1, 1) * invdet;
  r[5] = cof(2, 1) * invdet;
  r[6] = cof(0, 2) * invdet;
  r[7] = cof(1, 2) * invdet;
  r[8] = cof(2, 2) * invdet;
}

// Eigen dynamic inverse(): PartialPivLU then solve against the identity (4x4 here)
static void inverse_lu4(const double a[16], double out[16]) {
  double lu[16];
  int    perm[4] = {0, 1, 2, 3};
  memcpy(lu, a, sizeof lu);
  for (int k = 0; k < 4; k++) {
    int piv = k;
    for (int r = k + 1; r < 4; r++)
      if (fabs(lu[r * 4 + k]) > fabs(lu[piv * 4 + k])) piv = r;
    if (piv != k) {
      for (int c = 0; c < 4; c++) std::swap(lu[k * 4 + c], lu[piv * 4 + c]);
      std::swap(perm[k], perm[piv]);
    }
    for (int r = k + 1; r < 4; r++) {
      lu[r * 4 + k] /= lu[k * 4 + k];
      for (int c = k + 1; c < 4; c++) lu[r * 4 + c] -= lu[r * 4 + k] * lu[k * 4 + c];
    }
  }
  for (int col = 0; col < 4; col++) {
    double b[4];
    for (int r = 0; r < 4; r++) b[r] = (perm[r] == col) ? 1.0 : 0.0;
    for (int r = 1; r < 4; r++)
      for (int c = 0; c < r; c++) b[r] -= lu[r * 4 + c] * b[c];
    for (int r = 3; r >= 0; r--) {
      for (int c = r + 1; c < 4; c++) b[r] -= lu[r * 4 + c] * b[c];
      b[r] /= lu[r * 4 + r];
    }
    for (int r = 0; r < 4; r++) out[r * 4 + col] = b[r];
  }
}

// Mixer::calculateAllocation — controllers/mixer.hpp:72-101
static void mixer_allocation(const mrs_model_params_t& p, double ainv[MRS_MAXM * 4]) {
  const int     n = p.n_motors;
  const double* A = p.allocation_matrix;
  double        AAt[16], AAt_inv[16];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < n; k++) s += A[i * MRS_MAX_MOTORS + k] * A[j * MRS_MAX_MOTORS + k];
      AAt[i * 4 + j] = s;
    }
  inverse_lu4(AAt, AAt_inv);
  memset(ainv, 0, sizeof(double) * MRS_MAXM * 4);
  for (int m = 0; m < n; m++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += A[k * MRS_MAX_MOTORS + m] * AAt_inv[k * 4 + j];
      ainv[m * 4 + j] = s;
    }
  for (int m = 0; m < n; m++) {
    double*      r = &ainv[m * 4];
    const double z = r[0] * r[0] + r[1] * r[1];
    if (z > 0) {
      const double nn = sqrt(z);
      r[0] /= nn;
      r[1] /= nn;
    }
    r[2] = (r[2] > 1e-2) ? 1.0 : ((r[2] < -1e-2) ? -1.0 : 0.0);
    r[3] = 1.0;
  }
}


static void default_controllers(TypeKey& k) {  // UavSystem::initializeControllers, uav_system.hpp:159-169
  k.mixer = mrs_mixer_params_t{1, 0};
  k.rate  = mrs_rate_params_t{4.0, 0.04, 0.0};
  k.att   = mrs_attitude_params_t{6.0, 0.05, 0.01, 10.0, 1.0};
  k.vel   = mrs_velocity_params_t{2.0, 0.05, 0.01, 4.0};
  k.pos   = mrs_position_params_t{2.0, 0.15, 0.2, 6.0};
}

static void derive_type(const TypeKey& k, double dt, TypeParams& t) {
  const mrs_model_params_t& p = k.mp;
  memset(&t, 0, sizeof t);
  t.n_motors       = p.n_motors;
  t.ground_enabled = p.ground_enabled;
  t.desaturation   = k.mixer.desaturation;
  t.g              = p.g;
  t.mass           = p.mass;
  t.inv_mass       = 1.0 / p.mass;
  t.min_rpm        = p.min_rpm;
  t.max_rpm        = p.max_rpm;
  t.kf_n           = p.kf * p.n_motors;
  t.resist_k       = p.air_resistance_coeff * M_PI * (p.arm_length) * (p.arm_length);
  t.hover_thr      = 0.90 * sqrt((p.mass * p.g) / (p.n_motors * p.kf));
  t.ground_z       = p.ground_z;
  t.tau            = p.motor_time_constant;
  t.inv_kf_n       = 1.0 / t.kf_n;
  t.inv_rpm_range  = 1.0 / (p.max_rpm - p.min_rpm);
  t.filt_c         = exp((-dt) / (p.motor_time_constant));
  t.filt_1mc       = 1.0 - t.filt_c;
  t.arm_length     = p.arm_length;
  t.prop_radius    = p.prop_radius;
  memcpy(t.J, p.J, sizeof t.J);
  inverse3_cofactor(p.J, t.Jinv);
  for (int r = 0; r < 4; r++)
    for (int m = 0; m < MRS_MAXM; m++) t.alloc[r * MRS_MAXM + m] = (m < p.n_motors) ? p.allocation_matrix[r * MRS_MAX_MOTORS + m] : 0.0;
  mixer_allocation(p, t.alloc_inv);
  t.pos_kp = k.pos.kp; t.pos_kd = k.pos.kd; t.pos_ki = k.pos.ki; t.pos_sat = k.pos.max_velocity;
  t.vel_kp = k.vel.kp; t.vel_kd = k.vel.kd; t.vel_ki = k.vel.ki; t.vel_sat = k.vel.max_acceleration;
  t.att_kp = k.att.kp; t.att_kd = k.att.kd; t.att_ki = k.att.ki;
  t.att_sat_rp = k.att.max_rate_roll_pitch; t.att_sat_yaw = k.att.max_rate_yaw;
  {  // displacement bound (swarm_layout.h): thrust <= sum_m alloc[3][m] max(rpm_m, max_rpm)^2 <= |thrust now| + cap (pred_thr is the
     // factor of the first term, evaluated by the kernel from the motor speeds it has — also speeds the host set beyond max_rpm, or
     // a max_rpm lowered through set_params under running motors), times 1.5 for the re-orthonormalised body z of a not quite
     // orthonormal R
    double cap = 0.0;
    bool   ok  = p.mass > 0 && p.max_rpm >= 0;
    for (int m = 0; m < p.n_motors; m++) {
      const double a = p.allocation_matrix[3 * MRS_MAX_MOTORS + m];
      if (!(a >= 0)) ok = false;
      cap += fabs(a) * p.max_rpm * p.max_rpm;
    }
    t.pred_a0   = ok ? fabs(p.g) + 1.5 * cap / p.mass : INFINITY;
    t.pred_thr  = ok ? 1.5 / p.mass : INFINITY;
    t.pred_drag = ok ? fabs(t.resist_k) / p.mass : INFINITY;
  }
  for (int i = 0; i < 3; i++) {
    t.rate_kp[i] = k.rate.kp * p.J[i * 3 + i];
    t.rate_kd[i] = k.rate.kd * p.J[i * 3 + i];
    t.rate_ki[i] = k.rate.ki * p.J[i * 3 + i];
  }
}

// Eigen::AngleAxisd(angle, UnitZ).toRotationMatrix() (Eigen/src/Geometry/AngleAxis.h), row-major out
static void angle_axis_z(double angle, double R[9]) {
  const double ax[3] = {0, 0, 1};
  const double s = sin(angle), c = cos(angle);
  const double sa[3] = {s * ax[0], s * ax[1], s * ax[2]};
  const double ca[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
  double       tmp;
  tmp  = ca[0] * ax[1];
  R[1] = tmp - sa[2];
  R[3] = tmp + sa[2];
  tmp  = ca[0] * ax[2];
  R[2] = tmp + sa[1];
  R[6] = tmp - sa[1];
  tmp  = ca[1] * ax[2];
  R[5] = tmp - sa[0];
  R[7] = tmp + sa[0];
  R[0] = ca[0] * ax[0] + c;
  R[4] = ca[1] * ax[1] + c;
  R[8] = ca[2] * ax[2] + c;
}

namespace mrs_host {
void track_mode(mrs_swarm* s, int first, int count, int mode) {
  for (int k = 0; k < count; k++) {
    uint8_t& m = s->uav_mode[(size_t)first + k];
    s->n_cascade += (mode >= MRS_CONTROL_GROUP_CMD) - (m >= MRS_CONTROL_GROUP_CMD);
    m = (uint8_t)mode;
  }
}

int check_range(const mrs_swarm* s, int first, int count) {
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (first < 0 || count < 0 || (long long)first + count > s->n) return fail(MRS_ERR_RANGE, "uav range out of bounds");
  return MRS_OK;
}

int intern_type(mrs_swarm* s, const TypeKey& k, int* out) {
  std::string bytes(reinterpret_cast<const char*>(&k), sizeof k);
  auto        it = s->key_index.find(bytes);
  if (it != s->key_index.end()) {
    *out = it->second;
    return MRS_OK;
  }
  if ((int)s->keys.size() >= MRS_MAX_TYPES) return fail(MRS_ERR_TYPES, "type table full (65536 distinct parameter sets)");
  s->keys.push_back(k);
  TypeParams t;
  derive_type(k, s->table_dt > 0 ? s->table_dt : 0.001, t);
  s->tparams.push_back(t);
  *out = (int)s->keys.size() - 1;
  s->key_index.emplace(std::move(bytes), *out);
  s->types_dirty = true;
  return MRS_OK;
}

TypeKey make_key(const mrs_model_params_t* p) {
  TypeKey k;
  memset(&k, 0, sizeof k);  // deterministic padding bytes: the key is compared bytewise
  if (p)
    memcpy(&k.mp, p, sizeof k.mp);
  else
    mrs_model_params_default(&k.mp);
  k.mp.takeoff_patch_enabled = 0;
  k.mp._pad                  = 0;
  for (int r = 0; r < 4; r++)  // unused motor columns must not split types
    for (int m = k.mp.n_motors; m < MRS_MAX_MOTORS; m++) k.mp.allocation_matrix[r * MRS_MAX_MOTORS + m] = 0.0;
  default_controllers(k);
  return k;
}

int upload_blocks(mrs_swarm* s) {
  if (!s->blocks_dirty) return MRS_OK;
  s->nbr_dirty = true;  // some UAV changed its airframe type
  const int nb = s->npad / 64;
  s->block_type.assign((size_t)nb, 0);
  s->mixed_blocks.clear();
  for (int b = 0; b < nb; b++) {
    const int lo = b * 64, hi = (lo + 64 < s->n) ? lo + 64 : s->n;
    uint16_t  t  = lo < s->n ? s->uav_type[(size_t)lo] : 0;
    for (int i = lo + 1; i < hi; i++)
      if (s->uav_type[(size_t)i] != t) {
        t = 0xFFFFu;
        break;
      }
    // (a block beyond the last UAV, or a swarm without UAVs, has no airframe type to look up)
    const bool typed = t != 0xFFFFu && (size_t)t < s->keys.size();
    s->block_type[(size_t)b] = (uint32_t)t | ((typed ? (uint32_t)s->keys[t].mp.n_motors : (uint32_t)MRS_MAX_MOTORS) << 16);
    if (t == 0xFFFFu) s->mixed_blocks.push_back(b);
  }
  HIPCHK(hipStreamSynchronize(s->stream));
  if (!s->dBT) HIPCHK(hipMalloc(&s->dBT, sizeof(uint32_t) * (size_t)nb));
  if (!s->dMB) HIPCHK(hipMalloc(&s->dMB, sizeof(int32_t) * (size_t)nb));
  HIPCHK(hipMemcpyAsync(s->dBT, s->block_type.data(), sizeof(uint32_t) * (size_t)nb, hipMemcpyHostToDevice, s->stream));
  if (!s->mixed_blocks.empty())
    HIPCHK(hipMemcpyAsync(s->dMB, s->mixed_blocks.data(), sizeof(int32_t) * s->mixed_blocks.size(), hipMemcpyHostToDevice, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  s->blocks_dirty = false;
  return MRS_OK;
}

int upload_types(mrs_swarm* s, double dt) {
  {
    int rcb = upload_blocks(s);
    if (rcb) return rcb;
  }
  if (s->types_dirty) s->nbr_dirty = true;  // mass / arm length / propeller radius of the collision records may have changed
  if (!s->types_dirty && dt == s->table_dt) return MRS_OK;
  if (dt != s->table_dt) {
    for (size_t i = 0; i < s->keys.size(); i++) {
      s->tparams[i].filt_c   = exp((-dt) / (s->keys[i].mp.motor_time_constant));  // multirotor_model.hpp:244
      s->tparams[i].filt_1mc = 1.0 - s->tparams[i].filt_c;
    }
    s->table_dt = dt;
  }
  const int need = (int)s->tparams.size();
  if (need > s->dT_cap) {
    // the old table may still be read by launches in flight on the stream
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->dT) HIPCHK(hipFree(s->dT));
    int cap = 16;
    while (cap < need) cap *= 2;
    HIPCHK(hipMalloc(&s->dT, sizeof(TypeParams) * (size_t)cap));
    s->dT_cap = cap;
  }
  HIPCHK(hipMemcpyAsync(s->dT, s->tparams.data(), sizeof(TypeParams) * (size_t)need, hipMemcpyHostToDevice, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));  // tparams (pageable) may change right after we return
  s->types_dirty = false;
  return MRS_OK;
}

// upload one column (count doubles from the staging vector) into field f at [first, first+count)
int put_column(mrs_swarm* s, int f, int first, int count, const double* col) {
  if (f >= F_X && f < F_X + 3) s->nbr_dirty = true;
  HIPCHK(hipMemcpyAsync(s->dS + (size_t)f * s->npad + first, col, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return MRS_OK;
}
int fill_column(mrs_swarm* s, int f, int first, int count, double value) {
  if (value == 0.0) {
    HIPCHK(hipMemsetAsync(s->dS + (size_t)f * s->npad + first, 0, sizeof(double) * (size_t)count, s->stream));
    return MRS_OK;
  }
  s->stage.assign((size_t)count, value);
  return put_column(s, f, first, count, s->stage.data());
}
// host AoS (count x width, element j) -> device column
int put_strided(mrs_swarm* s, int f, int first, int count, const double* src, int width, int j) {
  s->stage.resize((size_t)count);
  for (int k = 0; k < count; k++) s->stage[(size_t)k] = src[(size_t)k * width + j];
  return put_column(s, f, first, count, s->stage.data());
}
int get_strided(mrs_swarm* s, int f, int first, int count, double* dst, int width, int j) {
  s->stage.resize((size_t)count);
  HIPCHK(hipMemcpyAsync(s->stage.data(), s->dS + (size_t)f * s->npad + first, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost,
                        s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  for (int k = 0; k < count; k++) dst[(size_t)k * width + j] = s->stage[(size_t)k];
  return MRS_OK;
}
int flags_update(mrs_swarm* s, int first, int count, uint32_t and_mask, uint32_t or_mask) {
  if (count <= 0) return MRS_OK;
  HIPCHK(mrs_launch_flags_update(s->dF, first, count, and_mask, or_mask, s->stream));
  return MRS_OK;
}

}  // namespace mrs_host

// shared body of the five controller-parameter setters: re-intern the type of every UAV in the range with one
// member of its key replaced, zero that controller's PID columns (pid_field < 0: the mixer has none)
template <class Mutator>
static int set_controller_params(mrs_swarm* s, int first, int count, int pid_field, Mutator mutate) {
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  std::map<int, int> remap;
  int                run_start = first, run_type = -1;
  for (int k = 0; k <= count; k++) {
    int nt = -1;
    if (k < count) {
      const int old = s->uav_type[(size_t)first + k];
      auto      it  = remap.find(old);
      if (it == remap.end()) {
        TypeKey key = s->keys[(size_t)old];
        mutate(key);
        if ((rc = intern_type(s, key, &nt))) return rc;
        remap[old] = nt;
      } else {
        nt = it->second;
      }
      s->uav_type[(size_t)first + k] = (uint16_t)nt;
      s->blocks_dirty = true;
    }
    if (k == count || nt != run_type) {
      if (run_type >= 0 && (rc = flags_update(s, run_start, first + k - run_start, ~(0xFFFFu << FLAG_TYPE_SHIFT), (uint32_t)run_type << FLAG_TYPE_SHIFT)))
        return rc;
      run_start = first + k;
      run_type  = nt;
    }
  }
  if (pid_field >= 0)
    for (int f = pid_field; f < pid_field + 6; f++)
      if ((rc = fill_column(s, f, first, count, 0.0))) return rc;
  return MRS_OK;
}


// callbackSetMass / callbackSetGroundZ: getParams -> modify -> setParams, UAV by UAV (types are interned, so a uniform range
// costs one table entry)
template <class Mutator>
static int modify_params(mrs_swarm* s, int first, int count, Mutator mutate) {
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  std::vector<uint32_t> fl((size_t)count);
  HIPCHK(hipMemcpyAsync(fl.data(), s->dF + first, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  int k = 0;
  while (k < count) {  // runs of UAVs that share (type, take-off flag)
    int e = k + 1;
    while (e < count && s->uav_type[(size_t)first + e] == s->uav_type[(size_t)first + k] && ((fl[(size_t)e] ^ fl[(size_t)k]) & FLAG_TAKEOFF) == 0) e++;
    mrs_model_params_t p   = s->keys[s->uav_type[(size_t)first + k]].mp;
    p.takeoff_patch_enabled = (fl[(size_t)k] & FLAG_TAKEOFF) ? 1 : 0;  // getParams() carries the mutated flag
    mutate(p);
    if ((rc = mrs_swarm_set_params(s, first + k, e - k, &p))) return rc;
    k = e;
  }
  return MRS_OK;
}


// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

const char* mrs_last_error(void) { return g_err.c_str(); }

int mrs_calculate_inertia(mrs_model_params_t* p) {
  if (!p) return fail(MRS_ERR_ARG, "null params");
  memset(p->J, 0, sizeof p->J);
  p->J[0] = p->mass * (3.0 * p->arm_length * p->arm_length + p->body_height * p->body_height) / 12.0;
  p->J[4] = p->mass * (3.0 * p->arm_length * p->arm_length + p->body_height * p->body_height) / 12.0;
  p->J[8] = (p->mass * p->arm_length * p->arm_length) / 2.0;
  return MRS_OK;
}

int mrs_scale_allocation(mrs_model_params_t* p) {
  if (!p) return fail(MRS_ERR_ARG, "null params");
  if (p->n_motors < 1 || p->n_motors > MRS_MAX_MOTORS) return fail(MRS_ERR_ARG, "n_motors must be 1..8");
  for (int m = 0; m < p->n_motors; m++) {
    p->allocation_matrix[0 * MRS_MAX_MOTORS + m] *= p->arm_length * p->kf;
    p->allocation_matrix[1 * MRS_MAX_MOTORS + m] *= p->arm_length * p->kf;
    p->allocation_matrix[2 * MRS_MAX_MOTORS + m] *= p->km * (3.0 * p->prop_radius) * p->kf;
    p->allocation_matrix[3 * MRS_MAX_MOTORS + m] *= p->kf;
  }
  return MRS_OK;
}

int mrs_model_params_default(mrs_model_params_t* p) {
  if (!p) return fail(MRS_ERR_ARG, "null params");
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
  mrs_calculate_inertia(p);
  for (int r = 0; r < 4; r++)
    for (int m = 0; m < 4; m++) p->allocation_matrix[r * MRS_MAX_MOTORS + m] = a[r][m];
  mrs_scale_allocation(p);
  p->ground_enabled        = 0;
  p->ground_z              = 0.0;
  p->takeoff_patch_enabled = 1;
  return MRS_OK;
}

int mrs_swarm_create(int32_t n_uavs, int32_t device_id, mrs_swarm_t** out) {
  if (!out || n_uavs < 0) return fail(MRS_ERR_ARG, "bad arguments");
  *out = nullptr;
  int        ndev = 0;
  hipError_t e0   = hipGetDeviceCount(&ndev);
  if (e0 != hipSuccess || ndev <= 0)
    return fail(MRS_ERR_HIP, std::string("no HIP device available (hipGetDeviceCount: ") + hipGetErrorString(e0) + ", count " +
                                 std::to_string(ndev) + "): libmrs_swarm has no CPU fallback");
  if (device_id < 0) HIPCHK(hipGetDevice(&device_id));
  if (device_id >= ndev) return fail(MRS_ERR_ARG, "device_id out of range");
  HIPCHK(hipSetDevice(device_id));
  mrs_swarm* s = new mrs_swarm();
  // any failure below hands the half-built object (streams, events, device buffers) back through mrs_swarm_destroy
  struct Guard {
    mrs_swarm* p;
    ~Guard() {
      if (p) mrs_swarm_destroy(p);
    }
  } guard{s};
  s->n         = n_uavs;
  s->npad      = ((n_uavs + 63) / 64) * 64;
  if (const char* e = getenv("MRS_NEIGHBOUR_LISTS")) s->use_lists = atoi(e) != 0;
  if (s->npad == 0) s->npad = 64;
  s->device = device_id;
  HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  {
    // The runtime maps streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) round-robin; once other
    // libraries in the process (torch, RCCL) have created theirs, two default-priority streams of ours can end up on the same
    // queue and their launches serialise.  Streams of different priority classes never share a queue.
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (const char* e = getenv("MRS_STREAM2_PRIORITY")) greatest = atoi(e);
    HIPCHK(hipStreamCreateWithPriority(&s->stream2, hipStreamNonBlocking, greatest));
  }
  HIPCHK(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&s->ev_join_b, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&s->ev_copy, hipEventDisableTiming));
  s->cstream = s->stream;
  if (const char* e = getenv("MRS_SPLIT_CU_RESERVE")) s->cu_reserve = atoi(e);
  if (s->cu_reserve > 0) {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device_id));
    const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
    if (s->cu_reserve >= ncu / 2) return fail(MRS_ERR_ARG, "MRS_SPLIT_CU_RESERVE: at most half of the device's compute units");
    std::vector<uint32_t> mb((size_t)words, 0u), mi((size_t)words, 0u);
    for (int b = 0; b < ncu; b++) (b < s->cu_reserve ? mb : mi)[(size_t)(b / 32)] |= 1u << (b % 32);
    HIPCHK(hipExtStreamCreateWithCUMask(&s->stream_b, (uint32_t)words, mb.data()));
    HIPCHK(hipExtStreamCreateWithCUMask(&s->stream_i, (uint32_t)words, mi.data()));
  }
  HIPCHK(hipEventCreate(&s->ev_end2));
  {
    int ncu = 0;
    HIPCHK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id));
    if (ncu > 0) s->resident_waves = ncu * 4 * 2;
  }
  if (const char* e = getenv("MRS_SPLIT_STREAMS")) s->split_steps = atoi(e) != 0;
  if (const char* e = getenv("MRS_FUSED_COLLISIONS")) s->use_fused = atoi(e) != 0;
  if (const char* e = getenv("MRS_FUSED_LEAD")) s->fused_lead = atoi(e) > 0 ? atoi(e) : 1;
  if (const char* e = getenv("MRS_SHARD_SPLIT")) s->shard_split = atoi(e) != 0;
  if (const char* e = getenv("MRS_EARLY_SEARCH")) s->early_search = atoi(e) != 0;
  if (const char* e = getenv("MRS_SHARD_SPLIT_MIN_BLOCKS")) s->split_min_blocks = atoi(e) > 0 ? atoi(e) : 1;
  if (const char* e = getenv("MRS_SHARD_SPLIT_MAX_FRACTION")) s->split_max_fraction = atof(e);
  HIPCHK(hipMalloc(&s->dS, sizeof(double) * (size_t)F_COUNT * s->npad));
  HIPCHK(hipMalloc(&s->dF, sizeof(uint32_t) * (size_t)s->npad));
  HIPCHK(hipMalloc(&s->dDiag, sizeof(unsigned long long) * 4));
  HIPCHK(hipMemsetAsync(s->dS, 0, sizeof(double) * (size_t)F_COUNT * s->npad, s->stream));
  HIPCHK(hipMemsetAsync(s->dF, 0, sizeof(uint32_t) * (size_t)s->npad, s->stream));
  HIPCHK(hipMemsetAsync(s->dDiag, 0, sizeof(unsigned long long) * 4, s->stream));
  s->uav_type.assign((size_t)s->npad, 0);
  s->uav_mode.assign((size_t)s->npad, (uint8_t)MRS_INPUT_UNKNOWN);
  if (n_uavs > 0) {
    int rc = mrs_swarm_construct(s, 0, n_uavs, nullptr, nullptr, nullptr);
    if (rc != MRS_OK) return rc;
  }
  guard.p = nullptr;
  *out    = s;
  return MRS_OK;
}

int mrs_swarm_destroy(mrs_swarm_t* s) {
  if (!s) return MRS_OK;
  (void)hipSetDevice(s->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  if (s->stream2) (void)hipStreamSynchronize(s->stream2);
  if (s->stream_b) (void)hipStreamSynchronize(s->stream_b);
  if (s->stream_i) (void)hipStreamSynchronize(s->stream_i);
  for (auto e : s->ev) (void)hipEventDestroy(e);
  mrs_collide_free(s->cwork);
  if (s->dRec) (void)hipFree(s->dRec);
  if (s->dOut) (void)hipFree(s->dOut);
  if (s->dSt) (void)hipFree(s->dSt);
  if (s->hSt) (void)hipHostFree(s->hSt);
  if (s->hOut) (void)hipHostFree(s->hOut);
  for (hipStream_t st : {s->stream_io, s->stream_up})
    if (st) (void)hipStreamSynchronize(st);  // (a copy may still be in flight into / out of the pinned blocks freed below)
  for (auto& o : s->oslot) {
    if (o.d) (void)hipFree(o.d);
    if (o.h) (void)hipHostFree(o.h);
    if (o.packed) (void)hipEventDestroy(o.packed);
    if (o.done) (void)hipEventDestroy(o.done);
  }
  for (auto& i : s->islot) {
    if (i.d) (void)hipFree(i.d);
    if (i.h) (void)hipHostFree(i.h);
    if (i.copied) (void)hipEventDestroy(i.copied);
    if (i.unpacked) (void)hipEventDestroy(i.unpacked);
  }
  for (hipStream_t st : {s->stream_io, s->stream_up})
    if (st) (void)hipStreamDestroy(st);
  if (s->dT) (void)hipFree(s->dT);
  if (s->dBT) (void)hipFree(s->dBT);
  if (s->dMB) (void)hipFree(s->dMB);
  if (s->dIota) (void)hipFree(s->dIota);
  if (s->dDiag) (void)hipFree(s->dDiag);
  if (s->dF) (void)hipFree(s->dF);
  if (s->dS) (void)hipFree(s->dS);
  if (s->rccl_comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(s->rccl_comm);
  peer_release(s);
  if (s->comm_send) (void)hipFree(s->comm_send);
  if (s->comm_recv) (void)hipFree(s->comm_recv);
  if (s->x_map_send) (void)hipFree(s->x_map_send);
  if (s->x_map_recv) (void)hipFree(s->x_map_recv);
  if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  if (s->ev_end2) (void)hipEventDestroy(s->ev_end2);
  if (s->stream2) (void)hipStreamDestroy(s->stream2);
  if (s->stream_b) (void)hipStreamDestroy(s->stream_b);
  if (s->stream_i) (void)hipStreamDestroy(s->stream_i);
  if (s->ev_join_b) (void)hipEventDestroy(s->ev_join_b);
  if (s->ev_copy) (void)hipEventDestroy(s->ev_copy);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
  return MRS_OK;
}

int mrs_swarm_size(const mrs_swarm_t* s, int32_t* n) {
  MRS_LOCK(s);
  if (!s || !n) return fail(MRS_ERR_ARG, "null argument");
  *n = s->n;
  return MRS_OK;
}

int mrs_swarm_set_arith(mrs_swarm_t* s, int32_t arith) {
  MRS_ENTER(s);
  if (!s || (arith != MRS_ARITH_LITERAL && arith != MRS_ARITH_FAST)) return fail(MRS_ERR_ARG, "bad arith");
  s->arith = arith;
  return MRS_OK;
}

int mrs_swarm_stream(const mrs_swarm_t* s, void** stream) {
  MRS_LOCK(s);
  if (!s || !stream) return fail(MRS_ERR_ARG, "null argument");
  *stream = (void*)s->stream;
  s->stream_exported = true;
  return MRS_OK;
}

int mrs_swarm_synchronize(mrs_swarm_t* s) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  HIPCHK(hipSetDevice(s->device));
  int rc = settle(s);  // ticks queued as no-ops behind a stale-list tick are replayed, the last collision tick is evaluated
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(s->stream));
  s->quiet_seq = s->op_seq;  // (the second stream was joined into this one by whoever used it)
  return MRS_OK;
}

int mrs_swarm_construct(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_model_params_t* params, const double* pos,
                        const double* heading) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (count == 0) return MRS_OK;
  if (params && (params->n_motors < 1 || params->n_motors > MRS_MAX_MOTORS)) return fail(MRS_ERR_ARG, "n_motors must be 1..8");
  HIPCHK(hipSetDevice(s->device));
  TypeKey key = make_key(params);
  int     type;
  if ((rc = intern_type(s, key, &type))) return rc;
  for (int k = 0; k < count; k++) s->uav_type[(size_t)first + k] = (uint16_t)type, s->blocks_dirty = true;
  track_mode(s, first, count, MRS_INPUT_UNKNOWN);
  // MultirotorModel::initializeState (multirotor_model.hpp:183-198): everything zero, R = I
  for (int f = 0; f < F_COUNT; f++) {
    const bool diag = (f == F_R + 0 || f == F_R + 4 || f == F_R + 8);
    if ((rc = fill_column(s, f, first, count, diag ? 1.0 : 0.0))) return rc;
  }
  if (pos) {  // MultirotorModel::setStatePos, multirotor_model.hpp:439-446
    for (int j = 0; j < 3; j++)
      if ((rc = put_strided(s, F_X + j, first, count, pos, 3, j))) return rc;
    if ((rc = put_strided(s, F_INITZ, first, count, pos, 3, 2))) return rc;
    std::vector<double> Rm((size_t)count * 9);
    for (int k = 0; k < count; k++) angle_axis_z(-(heading ? heading[k] : 0.0), &Rm[(size_t)k * 9]);
    for (int j = 0; j < 9; j++)
      if ((rc = put_strided(s, F_R + j, first, count, Rm.data(), 9, j))) return rc;
  }
  const int      takeoff = params ? params->takeoff_patch_enabled : 1;
  const uint32_t flags   = (takeoff ? FLAG_TAKEOFF : 0u) | ((uint32_t)MRS_INPUT_UNKNOWN << FLAG_MODE_SHIFT) | ((uint32_t)type << FLAG_TYPE_SHIFT);
  return flags_update(s, first, count, 0u, flags);
}

int mrs_swarm_set_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_model_params_t* params) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!params || params->n_motors < 1 || params->n_motors > MRS_MAX_MOTORS) return fail(MRS_ERR_ARG, "bad params");
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  TypeKey key = make_key(params);  // default gains: initializeControllers(), uav_system.hpp:404-409
  int     type;
  if ((rc = intern_type(s, key, &type))) return rc;
  for (int k = 0; k < count; k++) s->uav_type[(size_t)first + k] = (uint16_t)type, s->blocks_dirty = true;
  for (int f = F_PID; f < F_PID + 24; f++)
    if ((rc = fill_column(s, f, first, count, 0.0))) return rc;
  return flags_update(s, first, count, ~((0xFFFFu << FLAG_TYPE_SHIFT) | FLAG_TAKEOFF),
                      ((uint32_t)type << FLAG_TYPE_SHIFT) | (params->takeoff_patch_enabled ? FLAG_TAKEOFF : 0u));
}

int mrs_swarm_get_params(mrs_swarm_t* s, int32_t uav, mrs_model_params_t* out) {
  MRS_ENTER(s);
  int rc = check_range(s, uav, 1);
  if (rc) return rc;
  if (!out) return fail(MRS_ERR_ARG, "null out");
  HIPCHK(hipSetDevice(s->device));
  *out = s->keys[s->uav_type[(size_t)uav]].mp;
  uint32_t fl = 0;
  HIPCHK(hipMemcpyAsync(&fl, s->dF + uav, sizeof fl, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  out->takeoff_patch_enabled = (fl & FLAG_TAKEOFF) ? 1 : 0;
  return MRS_OK;
}

int mrs_swarm_set_mixer_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_mixer_params_t* p) {
  MRS_ENTER(s);
  if (!p) return fail(MRS_ERR_ARG, "null params");
  const mrs_mixer_params_t v{p->desaturation ? 1 : 0, 0};
  return set_controller_params(s, first, count, -1, [&](TypeKey& k) { k.mixer = v; });
}
int mrs_swarm_set_position_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_position_params_t* p) {
  MRS_ENTER(s);
  if (!p) return fail(MRS_ERR_ARG, "null params");
  return set_controller_params(s, first, count, F_PID + 0, [&](TypeKey& k) { k.pos = *p; });
}
int mrs_swarm_set_velocity_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_velocity_params_t* p) {
  MRS_ENTER(s);
  if (!p) return fail(MRS_ERR_ARG, "null params");
  return set_controller_params(s, first, count, F_PID + 6, [&](TypeKey& k) { k.vel = *p; });
}
int mrs_swarm_set_attitude_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_attitude_params_t* p) {
  MRS_ENTER(s);
  if (!p) return fail(MRS_ERR_ARG, "null params");
  return set_controller_params(s, first, count, F_PID + 12, [&](TypeKey& k) { k.att = *p; });
}
int mrs_swarm_set_rate_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_rate_params_t* p) {
  MRS_ENTER(s);
  if (!p) return fail(MRS_ERR_ARG, "null params");
  return set_controller_params(s, first, count, F_PID + 18, [&](TypeKey& k) { k.rate = *p; });
}

int mrs_swarm_get_mixer_allocation(mrs_swarm_t* s, int32_t uav, double* out) {
  MRS_ENTER(s);
  int rc = check_range(s, uav, 1);
  if (rc) return rc;
  if (!out) return fail(MRS_ERR_ARG, "null out");
  const TypeParams& t = s->tparams[s->uav_type[(size_t)uav]];
  memcpy(out, t.alloc_inv, sizeof(double) * 4 * (size_t)t.n_motors);
  return MRS_OK;
}

int mrs_swarm_set_input(mrs_swarm_t* s, int32_t first, int32_t count, int32_t mode, const double* payload, int32_t stride) {
  MRS_ENTER_COMMANDS(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (mode < MRS_INPUT_UNKNOWN || mode > MRS_POSITION_CMD) return fail(MRS_ERR_ARG, "bad input mode");
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int width = 0;
  switch (mode) {
    case MRS_INPUT_UNKNOWN: width = 0; break;
    case MRS_ACTUATOR_CMD: width = stride < MRS_MAX_MOTORS ? stride : MRS_MAX_MOTORS; break;
    case MRS_ATTITUDE_CMD: width = 10; break;
    case MRS_TILT_HDG_RATE_CMD: width = 5; break;
    default: width = 4; break;
  }
  if (width > 0 && (!payload || stride < width)) return fail(MRS_ERR_ARG, "payload missing or stride too small for this mode");
  if (mode == MRS_ACTUATOR_CMD) {
    for (int k = 0; k < count; k++)
      if (s->keys[s->uav_type[(size_t)first + k]].mp.n_motors > width)
        return fail(MRS_ERR_ARG, "actuator payload narrower than n_motors");
  }
  for (int j = 0; j < width; j++)
    if ((rc = put_strided(s, F_CMD + j, first, count, payload, stride, j))) return rc;
  track_mode(s, first, count, mode);
  return flags_update(s, first, count, ~FLAG_MODE_MASK, (uint32_t)mode << FLAG_MODE_SHIFT);
}

int mrs_swarm_set_feedforward(mrs_swarm_t* s, int32_t first, int32_t count, int32_t kind, const double* payload, int32_t stride) {
  MRS_ENTER_COMMANDS(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (kind < 0 || kind > 3 || !payload || stride < 4) return fail(MRS_ERR_ARG, "bad feed-forward arguments");
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  for (int j = 0; j < 4; j++)
    if ((rc = put_strided(s, F_FF + 4 * kind + j, first, count, payload, stride, j))) return rc;
  return flags_update(s, first, count, ~0u, (1u << kind) << FLAG_FF_SHIFT);
}

int mrs_swarm_apply_force(mrs_swarm_t* s, int32_t first, int32_t count, const double* force) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!force) return fail(MRS_ERR_ARG, "null force");
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  for (int j = 0; j < 3; j++)
    if ((rc = put_strided(s, F_FEXT + j, first, count, force, 3, j))) return rc;
  s->fext_active = true;
  return MRS_OK;
}

int mrs_swarm_crash(mrs_swarm_t* s, int32_t first, int32_t count) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  return flags_update(s, first, count, ~0u, FLAG_CRASHED);
}

int mrs_swarm_set_hold(mrs_swarm_t* s, int32_t first, int32_t count, int32_t hold) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  return flags_update(s, first, count, ~FLAG_HOLD, hold ? FLAG_HOLD : 0u);
}

int mrs_swarm_has_crashed(mrs_swarm_t* s, int32_t first, int32_t count, int32_t* out) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!out) return fail(MRS_ERR_ARG, "null out");
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  s->stage_u.resize((size_t)count);
  HIPCHK(hipMemcpyAsync(s->stage_u.data(), s->dF + first, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  for (int k = 0; k < count; k++) out[k] = (s->stage_u[(size_t)k] & FLAG_CRASHED) ? 1 : 0;
  return MRS_OK;
}

// ---- state access ----
int mrs_swarm_get_state(mrs_swarm_t* s, int32_t first, int32_t count, double* x, double* v, double* v_prev, double* R, double* omega,
                        double* motor_rpm) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  struct { double* p; int f, w; } items[6] = {{x, F_X, 3}, {v, F_V, 3}, {v_prev, F_VPREV, 3}, {R, F_R, 9}, {omega, F_W, 3}, {motor_rpm, F_RPM, MRS_MAX_MOTORS}};
  for (auto& it : items)
    if (it.p)
      for (int j = 0; j < it.w; j++)
        if ((rc = get_strided(s, it.f + j, first, count, it.p, it.w, j))) return rc;
  if (v_prev) {  // v_prev == v unless the UAV is flagged (see FLAG_VPREV_SPLIT)
    std::vector<uint32_t> fl((size_t)count);
    std::vector<double>   vv((size_t)count * 3);
    HIPCHK(hipMemcpyAsync(fl.data(), s->dF + first, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    for (int j = 0; j < 3; j++)
      if ((rc = get_strided(s, F_V + j, first, count, vv.data(), 3, j))) return rc;
    for (int k = 0; k < count; k++)
      if (!(fl[(size_t)k] & FLAG_VPREV_SPLIT))
        for (int j = 0; j < 3; j++) v_prev[(size_t)k * 3 + j] = vv[(size_t)k * 3 + j];
  }
  return MRS_OK;
}

int mrs_swarm_set_state(mrs_swarm_t* s, int32_t first, int32_t count, const double* x, const double* v, const double* R,
                        const double* omega, const double* motor_rpm) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  struct { const double* p; int f, w; } items[5] = {{x, F_X, 3}, {v, F_V, 3}, {R, F_R, 9}, {omega, F_W, 3}, {motor_rpm, F_RPM, MRS_MAX_MOTORS}};
  if (v) {
    // MultirotorModel::setState leaves v_prev alone: materialise it in its column (it equals the old v unless the flag is
    // already set) before v is overwritten, and mark the UAVs
    std::vector<uint32_t> fl((size_t)count);
    HIPCHK(hipMemcpyAsync(fl.data(), s->dF + first, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    std::vector<double> oldv((size_t)count), vp((size_t)count);
    for (int j = 0; j < 3; j++) {
      HIPCHK(hipMemcpyAsync(oldv.data(), s->dS + (size_t)(F_V + j) * s->npad + first, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipMemcpyAsync(vp.data(), s->dS + (size_t)(F_VPREV + j) * s->npad + first, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s->stream));
      HIPCHK(hipStreamSynchronize(s->stream));
      for (int k = 0; k < count; k++)
        if (!(fl[(size_t)k] & FLAG_VPREV_SPLIT)) vp[(size_t)k] = oldv[(size_t)k];
      if ((rc = put_column(s, F_VPREV + j, first, count, vp.data()))) return rc;
    }
    if ((rc = flags_update(s, first, count, ~0u, FLAG_VPREV_SPLIT))) return rc;
  }
  for (auto& it : items)
    if (it.p)
      for (int j = 0; j < it.w; j++) {
        if (it.f == F_RPM) {  // state_.motor_rpm has n_motors entries: columns beyond a UAV's motor count stay zero
          s->stage.resize((size_t)count);
          for (int k = 0; k < count; k++) {
            const mrs_model_params_t& mp = s->keys[s->uav_type[(size_t)first + k]].mp;
            s->stage[(size_t)k] = j < mp.n_motors ? it.p[(size_t)k * it.w + j] : 0.0;
          }
          if ((rc = put_column(s, it.f + j, first, count, s->stage.data()))) return rc;
        } else if ((rc = put_strided(s, it.f + j, first, count, it.p, it.w, j))) {
          return rc;
        }
      }
  return MRS_OK;
}

int mrs_swarm_set_state_pos(mrs_swarm_t* s, int32_t first, int32_t count, const double* pos, const double* heading) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!pos) return fail(MRS_ERR_ARG, "null position");
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  for (int j = 0; j < 3; j++)
    if ((rc = put_strided(s, F_X + j, first, count, pos, 3, j))) return rc;
  if ((rc = put_strided(s, F_INITZ, first, count, pos, 3, 2))) return rc;
  std::vector<double> Rm((size_t)count * 9);
  for (int k = 0; k < count; k++) angle_axis_z(-(heading ? heading[k] : 0.0), &Rm[(size_t)k * 9]);
  for (int j = 0; j < 9; j++)
    if ((rc = put_strided(s, F_R + j, first, count, Rm.data(), 9, j))) return rc;
  return MRS_OK;
}

int mrs_swarm_set_pid(mrs_swarm_t* s, int32_t first, int32_t count, const double* pid) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!pid) return fail(MRS_ERR_ARG, "null pid");
  HIPCHK(hipSetDevice(s->device));
  for (int j = 0; j < 24 && count > 0; j++)
    if ((rc = put_strided(s, F_PID + j, first, count, pid, 24, j))) return rc;
  return MRS_OK;
}

int mrs_swarm_clone_resized(mrs_swarm_t* s, int32_t n_uavs, mrs_swarm_t** out) {
  MRS_ENTER(s);
  if (!s || !out) return fail(MRS_ERR_ARG, "null argument");
  if (n_uavs < s->n) return fail(MRS_ERR_ARG, "a resized clone holds at least the UAVs of the original");
  *out = nullptr;
  mrs_swarm* c = nullptr;
  int rc = mrs_swarm_create(n_uavs, s->device, &c);  // (UAVs beyond the original's: UavSystem(), type 0 of every table)
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  // (a larger copy: only the original's UAVs travel — the padding lanes of its last block are real UAVs of the copy and keep their own initial values)
  const size_t w = (size_t)(n_uavs == s->n ? s->npad : s->n);
  hipError_t   e = hipMemcpy2DAsync(c->dS, sizeof(double) * (size_t)c->npad, s->dS, sizeof(double) * (size_t)s->npad, sizeof(double) * w, F_COUNT,
                                    hipMemcpyDeviceToDevice, s->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(c->dF, s->dF, sizeof(uint32_t) * w, hipMemcpyDeviceToDevice, s->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(c->dDiag, s->dDiag, sizeof(unsigned long long) * 4, hipMemcpyDeviceToDevice, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess) {
    mrs_swarm_destroy(c);
    return fail(MRS_ERR_HIP, std::string("clone: ") + hipGetErrorString(e));
  }
  c->arith     = s->arith;
  c->keys      = s->keys;  // (index 0 of every table is UavSystem()'s default parameter set: the extra UAVs stay valid)
  c->tparams   = s->tparams;
  c->key_index = s->key_index;
  for (int i = 0; i < s->n; i++) {
    c->uav_type[(size_t)i] = s->uav_type[(size_t)i];
    c->uav_mode[(size_t)i] = s->uav_mode[(size_t)i];
  }
  c->n_cascade    = s->n_cascade;
  c->table_dt     = s->table_dt;
  c->fext_active  = s->fext_active;
  c->use_lists    = s->use_lists;
  c->types_dirty  = true;  // the copy uploads its own type / block tables before its first launch
  c->blocks_dirty = true;
  c->nbr_dirty    = true;
  *out = c;
  return MRS_OK;
}

int mrs_swarm_clone(mrs_swarm_t* s, mrs_swarm_t** out) {
  if (!s) return fail(MRS_ERR_ARG, "null argument");
  return mrs_swarm_clone_resized(s, s->n, out);
}

int mrs_swarm_copy_uavs(mrs_swarm_t* dst, int32_t dst_first, mrs_swarm_t* src, int32_t src_first, int32_t count) {
  if (!dst || !src) return fail(MRS_ERR_ARG, "null swarm");
  // (two swarms: locked in address order, so that two threads copying in opposite directions cannot deadlock)
  std::unique_lock<std::recursive_mutex> l1((dst < src ? dst : src)->mtx), l2;
  if (dst != src) l2 = std::unique_lock<std::recursive_mutex>((dst < src ? src : dst)->mtx);
  dst->op_seq++;
  src->op_seq++;
  int rc;
  if ((rc = settle(src)) || (rc = settle(dst))) return rc;
  if ((rc = check_range(dst, dst_first, count)) || (rc = check_range(src, src_first, count))) return rc;
  if (count == 0) return MRS_OK;
  if (dst->device != src->device) return fail(MRS_ERR_ARG, "copy_uavs: both swarms must live on the same device");
  if (dst == src && dst_first < src_first + count && src_first < dst_first + count) return fail(MRS_ERR_ARG, "copy_uavs: overlapping ranges of one swarm");
  HIPCHK(hipSetDevice(dst->device));
  // the flag words carry indices into the type table: the tables must agree on every index the copied UAVs use (clones of each
  // other do: tables only ever grow at their end); indices the destination does not have yet are appended in the source's order
  for (int k = 0; k < count; k++) {
    const size_t t = src->uav_type[(size_t)src_first + k];
    while (dst->keys.size() <= t) {
      const TypeKey& key = src->keys[dst->keys.size()];
      int            got = -1;
      if ((rc = intern_type(dst, key, &got))) return rc;
      if ((size_t)got + 1 != dst->keys.size()) return fail(MRS_ERR_TYPES, "copy_uavs: the two swarms are not clones of each other (their type tables differ)");
    }
    if (memcmp(&dst->keys[t], &src->keys[t], sizeof(TypeKey)) != 0)
      return fail(MRS_ERR_TYPES, "copy_uavs: the two swarms are not clones of each other (their type tables differ)");
  }
  // No host synchronisation: every write to a swarm's columns is work on its `stream` (settle() above has joined the others), so
  // the copy goes behind the source's work on the SOURCE's stream, after the destination's earlier work (event), and the
  // destination's later work waits for it (event); whatever overwrites the source afterwards is behind the copy in stream order.
  if (dst != src) {
    HIPCHK(hipEventRecord(dst->ev_copy, dst->stream));
    HIPCHK(hipStreamWaitEvent(src->stream, dst->ev_copy, 0));
  }
  HIPCHK(hipMemcpy2DAsync(dst->dS + dst_first, sizeof(double) * (size_t)dst->npad, src->dS + src_first, sizeof(double) * (size_t)src->npad,
                          sizeof(double) * (size_t)count, F_COUNT, hipMemcpyDeviceToDevice, src->stream));
  HIPCHK(hipMemcpyAsync(dst->dF + dst_first, src->dF + src_first, sizeof(uint32_t) * (size_t)count, hipMemcpyDeviceToDevice, src->stream));
  if (dst != src) {
    HIPCHK(hipEventRecord(src->ev_copy, src->stream));
    HIPCHK(hipStreamWaitEvent(dst->stream, src->ev_copy, 0));
  }
  for (int k = 0; k < count; k++) {
    const uint16_t t = src->uav_type[(size_t)src_first + k];
    if (dst->uav_type[(size_t)dst_first + k] != t) {
      dst->uav_type[(size_t)dst_first + k] = t;
      dst->blocks_dirty = true;
    }
  }
  for (int k = 0; k < count; k++) track_mode(dst, dst_first + k, 1, src->uav_mode[(size_t)src_first + k]);
  dst->nbr_dirty = true;  // positions changed under the neighbour lists
  dst->p_valid   = false;
  dst->fext_active = dst->fext_active || src->fext_active;
  return MRS_OK;
}

int mrs_swarm_get_imu(mrs_swarm_t* s, int32_t first, int32_t count, double* imu) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!imu) return fail(MRS_ERR_ARG, "null out");
  HIPCHK(hipSetDevice(s->device));
  for (int j = 0; j < 3 && count > 0; j++)
    if ((rc = get_strided(s, F_IMU + j, first, count, imu, 3, j))) return rc;
  return MRS_OK;
}

int mrs_swarm_get_external_force(mrs_swarm_t* s, int32_t first, int32_t count, double* force) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!force) return fail(MRS_ERR_ARG, "null out");
  HIPCHK(hipSetDevice(s->device));
  for (int j = 0; j < 3 && count > 0; j++)
    if ((rc = get_strided(s, F_FEXT + j, first, count, force, 3, j))) return rc;
  return MRS_OK;
}

int mrs_swarm_get_pid(mrs_swarm_t* s, int32_t first, int32_t count, double* pid) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!pid) return fail(MRS_ERR_ARG, "null out");
  HIPCHK(hipSetDevice(s->device));
  for (int j = 0; j < 24 && count > 0; j++)
    if ((rc = get_strided(s, F_PID + j, first, count, pid, 24, j))) return rc;
  return MRS_OK;
}

int mrs_swarm_get_diag(mrs_swarm_t* s, mrs_diag_t* out) {
  MRS_ENTER(s);
  if (!s || !out) return fail(MRS_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(s->device));
  unsigned long long d[4];
  HIPCHK(hipMemcpyAsync(d, s->dDiag, sizeof d, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  out->hdg_rate_denom_small = d[0];
  out->projected_norm_small = d[1];
  out->yaw_rate_not_finite  = d[2];
  out->nan_rollback         = d[3];
  return MRS_OK;
}

int mrs_swarm_timeout_input(mrs_swarm_t* s, int32_t first, int32_t count) {
  MRS_ENTER_COMMANDS(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(mrs_launch_timeout_input(s->view(), first, count, s->stream));
  return MRS_OK;  // the mode of every UAV is unchanged (a safe command OF THE SAME MODE is substituted)
}

int mrs_swarm_set_mass(mrs_swarm_t* s, int32_t first, int32_t count, double mass) {
  MRS_ENTER(s);
  return modify_params(s, first, count, [&](mrs_model_params_t& p) {  // src/uav_system_ros.cpp:1036-1047
    const double original_mass = p.mass;
    p.mass = mass;
    for (int m = 0; m < p.n_motors; m++) p.allocation_matrix[2 * MRS_MAX_MOTORS + m] = p.mass * (p.allocation_matrix[2 * MRS_MAX_MOTORS + m] / original_mass);
    mrs_calculate_inertia(&p);
  });
}

int mrs_swarm_set_ground_z(mrs_swarm_t* s, int32_t first, int32_t count, double ground_z) {
  MRS_ENTER(s);
  return modify_params(s, first, count, [&](mrs_model_params_t& p) { p.ground_z = ground_z; });  // :1063-1073
}

static int fetch_outputs(mrs_swarm* s, int32_t first, int32_t count) {
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001);
  if (rc) return rc;
  if (count > s->out_cap) {
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->dOut) HIPCHK(hipFree(s->dOut));
    if (s->hOut) HIPCHK(hipHostFree(s->hOut));
    HIPCHK(hipMalloc(&s->dOut, sizeof(mrs_uav_output_t) * (size_t)count));
    HIPCHK(hipHostMalloc(&s->hOut, sizeof(mrs_uav_output_t) * (size_t)count, hipHostMallocDefault));
    s->out_cap = count;
  }
  HIPCHK(mrs_launch_pack_outputs(s->view(), first, count, s->dOut, s->stream));
  HIPCHK(hipMemcpyAsync(s->hOut, s->dOut, sizeof(mrs_uav_output_t) * (size_t)count, hipMemcpyDeviceToHost, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return MRS_OK;
}

int mrs_swarm_get_outputs(mrs_swarm_t* s, int32_t first, int32_t count, mrs_uav_output_t* out) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!out) return fail(MRS_ERR_ARG, "null out");
  if (count == 0) return MRS_OK;
  if ((rc = fetch_outputs(s, first, count))) return rc;
  memcpy(out, s->hOut, sizeof(mrs_uav_output_t) * (size_t)count);
  return MRS_OK;
}

int mrs_swarm_get_outputs_view(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_uav_output_t** view) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!view) return fail(MRS_ERR_ARG, "null view");
  *view = nullptr;
  if (count == 0) return MRS_OK;
  if ((rc = fetch_outputs(s, first, count))) return rc;
  *view = s->hOut;
  return MRS_OK;
}

}  // extern "C"
namespace mrs_host {
static int ensure_io_stream(mrs_swarm* s) {
  if (s->stream_io) return MRS_OK;
  HIPCHK(hipStreamCreateWithFlags(&s->stream_io, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&s->stream_up, hipStreamNonBlocking));
  return MRS_OK;
}
// pack behind everything queued on the step stream, copy on the copy stream (also called by drain() when the launch the pack
// followed is replayed after a stall)
int issue_outputs(mrs_swarm* s, int slot) {
  mrs_swarm::OutSlot& o = s->oslot[slot];
  HIPCHK(hipStreamWaitEvent(s->stream, o.done, 0));  // the copy out of this buffer two tickets ago (a no-op when long complete)
  HIPCHK(mrs_launch_pack_outputs(s->view(), o.first, o.count, o.d, s->stream));
  HIPCHK(hipEventRecord(o.packed, s->stream));
  HIPCHK(hipStreamWaitEvent(s->stream_io, o.packed, 0));
  HIPCHK(hipMemcpyAsync(o.h, o.d, sizeof(mrs_uav_output_t) * (size_t)o.count, hipMemcpyDeviceToHost, s->stream_io));
  HIPCHK(hipEventRecord(o.done, s->stream_io));
  return MRS_OK;
}
}  // namespace mrs_host
extern "C" {

int mrs_swarm_get_outputs_async(mrs_swarm_t* s, int32_t first, int32_t count, int32_t* ticket) {
  MRS_LOCK(s);  // NOT settle(): launches of lazily evaluated collision ticks stay queued; a stall among them is put right at the wait
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!ticket) return fail(MRS_ERR_ARG, "null ticket");
  if (count == 0) return fail(MRS_ERR_ARG, "empty range");
  HIPCHK(hipSetDevice(s->device));
  if ((rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001))) return rc;
  if ((rc = ensure_io_stream(s))) return rc;
  const int32_t        t = s->out_tickets;
  mrs_swarm::OutSlot&  o = s->oslot[t & 1];
  if (!o.packed) {
    HIPCHK(hipEventCreateWithFlags(&o.packed, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&o.done, hipEventDisableTiming));
  }
  if (count > o.cap) {
    HIPCHK(hipEventSynchronize(o.done));  // (nobody copies out of the old buffer any more)
    if (o.d) HIPCHK(hipFree(o.d));
    if (o.h) HIPCHK(hipHostFree(o.h));
    o.d = nullptr; o.h = nullptr; o.cap = 0;
    HIPCHK(hipMalloc(&o.d, sizeof(mrs_uav_output_t) * (size_t)count));
    HIPCHK(hipHostMalloc(&o.h, sizeof(mrs_uav_output_t) * (size_t)count, hipHostMallocDefault));
    o.cap = count;
  }
  o.ticket = t;
  o.first  = first;
  o.count  = count;
  if ((rc = issue_outputs(s, t & 1))) return rc;
  if (!s->log.empty()) s->log.back().out_ticket = t;  // the pack followed this launch: a replay of it repeats the pack
  s->out_tickets++;
  *ticket = t;
  return MRS_OK;
}

int mrs_swarm_outputs_wait(mrs_swarm_t* s, int32_t ticket, const mrs_uav_output_t** view, int32_t* count) {
  MRS_LOCK(s);
  if (!s || !view) return fail(MRS_ERR_ARG, "null argument");
  *view = nullptr;
  if (ticket < 0 || ticket >= s->out_tickets) return fail(MRS_ERR_ARG, "no such ticket");
  mrs_swarm::OutSlot& o = s->oslot[ticket & 1];
  if (o.ticket != ticket) return fail(MRS_ERR_ARG, "this ticket's block has been handed to a newer download (two downloads in flight at most)");
  HIPCHK(hipSetDevice(s->device));
  for (;;) {
    HIPCHK(hipEventSynchronize(o.done));  // that copy only: steps queued behind the pack keep running
    // launches of lazily evaluated collision ticks may have turned into no-ops before the pack ran (a UAV left its skin: DESIGN §4 K1b);
    // drain() repeats the search, replays them — and re-issues every pack that followed a replayed launch
    const volatile unsigned* hw = mrs_collide_host_words(s->cwork);
    if (s->log.empty() || !hw || hw[CTL_STALL] == 0u) break;
    int rc = drain(s);
    if (rc) return rc;
  }
  *view = o.h;
  if (count) *count = o.count;
  return MRS_OK;
}

int mrs_swarm_get_states(mrs_swarm_t* s, int32_t first, int32_t count, mrs_uav_state_t* out) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!out) return fail(MRS_ERR_ARG, "null out");
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  if ((rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001))) return rc;
  if (count > s->st_cap) {
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->dSt) HIPCHK(hipFree(s->dSt));
    if (s->hSt) HIPCHK(hipHostFree(s->hSt));
    s->dSt = nullptr;
    s->hSt = nullptr;
    s->st_cap = 0;
    HIPCHK(hipMalloc(&s->dSt, sizeof(mrs_uav_state_t) * (size_t)count));
    HIPCHK(hipHostMalloc(&s->hSt, sizeof(mrs_uav_state_t) * (size_t)count, hipHostMallocMapped));
    s->st_cap = count;
  }
  // few UAVs (the facade's lone object, a pool's round): the pack kernel stores straight into the pinned block — no copy command
  // behind it (a DMA of a few hundred bytes costs the stream ~8 us, a third of a single object's makeStep + getState)
  static const int direct_max = getenv("MRS_DIRECT_STATE_MAX") ? atoi(getenv("MRS_DIRECT_STATE_MAX")) : 1024;
  mrs_uav_state_t* host_dev = nullptr;
  if (count <= direct_max && hipHostGetDevicePointer((void**)&host_dev, s->hSt, 0) == hipSuccess && host_dev) {
    HIPCHK(mrs_launch_pack_states(s->view(), first, count, host_dev, s->stream));
  } else {
    HIPCHK(mrs_launch_pack_states(s->view(), first, count, s->dSt, s->stream));
    HIPCHK(hipMemcpyAsync(s->hSt, s->dSt, sizeof(mrs_uav_state_t) * (size_t)count, hipMemcpyDeviceToHost, s->stream));
  }
  HIPCHK(hipStreamSynchronize(s->stream));
  memcpy(out, s->hSt, sizeof(mrs_uav_state_t) * (size_t)count);
  return MRS_OK;
}

int mrs_swarm_input_staging(mrs_swarm_t* s, int32_t count, int32_t stride, double** rows) {
  MRS_LOCK(s);
  if (!s || !rows) return fail(MRS_ERR_ARG, "null argument");
  if (count < 0 || count > s->n || stride < 1 || stride > 16) return fail(MRS_ERR_ARG, "bad staging shape");
  HIPCHK(hipSetDevice(s->device));
  int rc = ensure_io_stream(s);
  if (rc) return rc;
  const int64_t need = (int64_t)count * stride;
  s->in_turn ^= 1;  // the other block: the commit of the previous one may still be copying
  mrs_swarm::InSlot& b = s->islot[s->in_turn];
  if (!b.copied) {
    HIPCHK(hipEventCreateWithFlags(&b.copied, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&b.unpacked, hipEventDisableTiming));
  }
  HIPCHK(hipEventSynchronize(b.copied));  // the commit two calls ago has read these rows (that copy only: no stream is waited for)
  if (need > b.cap) {
    HIPCHK(hipEventSynchronize(b.unpacked));
    if (b.h) HIPCHK(hipHostFree(b.h));
    if (b.d) HIPCHK(hipFree(b.d));
    b.h = nullptr; b.d = nullptr; b.cap = 0;
    HIPCHK(hipHostMalloc(&b.h, sizeof(double) * (size_t)need, hipHostMallocDefault));
    HIPCHK(hipMalloc(&b.d, sizeof(double) * (size_t)need));
    b.cap = need;
  }
  *rows = b.h;
  return MRS_OK;
}

int mrs_swarm_commit_input(mrs_swarm_t* s, int32_t first, int32_t count, int32_t mode, int32_t stride) {
  MRS_ENTER_COMMANDS(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (mode < MRS_ACTUATOR_CMD || mode > MRS_POSITION_CMD) return fail(MRS_ERR_ARG, "bad input mode");
  if (count == 0) return MRS_OK;
  mrs_swarm::InSlot& b = s->islot[s->in_turn];
  if (!b.h || (int64_t)count * stride > b.cap) return fail(MRS_ERR_ARG, "no staging rows of this shape (call mrs_swarm_input_staging first)");
  int width = 4;
  if (mode == MRS_ACTUATOR_CMD) width = stride < MRS_MAX_MOTORS ? stride : MRS_MAX_MOTORS;
  if (mode == MRS_ATTITUDE_CMD) width = 10;
  if (mode == MRS_TILT_HDG_RATE_CMD) width = 5;
  if (stride < width) return fail(MRS_ERR_ARG, "stride too small for this mode");
  if (mode == MRS_ACTUATOR_CMD) {
    for (int k = 0; k < count; k++)
      if (s->keys[s->uav_type[(size_t)first + k]].mp.n_motors > width) return fail(MRS_ERR_ARG, "actuator payload narrower than n_motors");
  }
  HIPCHK(hipSetDevice(s->device));
  // the copy runs on the copy stream (beside whatever step is running), the step stream takes it in where this call stands
  HIPCHK(hipStreamWaitEvent(s->stream_up, b.unpacked, 0));  // the unpack kernel of this block's previous commit
  HIPCHK(hipMemcpyAsync(b.d, b.h, sizeof(double) * (size_t)count * (size_t)stride, hipMemcpyHostToDevice, s->stream_up));
  HIPCHK(hipEventRecord(b.copied, s->stream_up));
  HIPCHK(hipStreamWaitEvent(s->stream, b.copied, 0));
  HIPCHK(mrs_launch_unpack_rows(s->view(), b.d, stride, width, F_CMD, first, count, s->stream));
  HIPCHK(hipEventRecord(b.unpacked, s->stream));
  track_mode(s, first, count, mode);
  return flags_update(s, first, count, ~FLAG_MODE_MASK, (uint32_t)mode << FLAG_MODE_SHIFT);
}

int mrs_swarm_get_collision_stats(mrs_swarm_t* s, int64_t* n_ticks, int64_t* n_rebuilds) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  HIPCHK(hipSetDevice(s->device));
  unsigned rb = 0;
  HIPCHK(mrs_collide_rebuilds(s->cwork, s->stream, &rb));
  if (n_ticks) *n_ticks = s->collision_ticks;
  if (n_rebuilds) *n_rebuilds = s->use_lists ? (int64_t)rb : s->collision_ticks;
  return MRS_OK;
}

int mrs_swarm_get_fused_stats(mrs_swarm_t* s, int64_t* fused_launches, int64_t* stalls, int64_t* replayed_launches, int64_t* searches_ahead) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (fused_launches) *fused_launches = s->n_fused;
  if (stalls) *stalls = s->n_stalls;
  if (replayed_launches) *replayed_launches = s->n_noop_launches;
  if (searches_ahead) *searches_ahead = s->n_ahead_searches;
  return MRS_OK;
}

// debugging aid for tools/ (deliberately not declared in include/mrs_swarm.h)
// test hook: the kernels' PID device function over caller-given sequences (see mrs_pid_probe in step_device.inc)
int mrs_debug_pid_sequences(int32_t device_id, int32_t arith, int32_t n_seq, int32_t n_steps, const double* params, const double* err,
                            const double* dt, const double* event, const double* new_sat, double* out) {
  if (n_seq < 0 || n_steps < 0 || !params || !err || !dt || !event || !new_sat || !out) return fail(MRS_ERR_ARG, "bad pid probe arguments");
  if (arith != MRS_ARITH_LITERAL && arith != MRS_ARITH_FAST) return fail(MRS_ERR_ARG, "unknown arithmetic flavour");
  if (n_seq == 0 || n_steps == 0) return MRS_OK;
  if (device_id >= 0) HIPCHK(hipSetDevice(device_id));
  const size_t cells = (size_t)n_seq * (size_t)n_steps;
  double*      d     = nullptr;  // params | err | dt | event | new_sat | out
  HIPCHK(hipMalloc(&d, sizeof(double) * ((size_t)n_seq * 5 + cells * 5)));
  double *dp = d, *de = dp + (size_t)n_seq * 5, *dd = de + cells, *dv = dd + cells, *ds = dv + cells, *dout = ds + cells;
  hipError_t e = hipMemcpy(dp, params, sizeof(double) * (size_t)n_seq * 5, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(de, err, sizeof(double) * cells, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dd, dt, sizeof(double) * cells, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dv, event, sizeof(double) * cells, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ds, new_sat, sizeof(double) * cells, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = arith == MRS_ARITH_FAST ? mrs_launch_pid_probe_fast(dp, de, dd, dv, ds, dout, n_seq, n_steps, nullptr)
                                : mrs_launch_pid_probe_literal(dp, de, dd, dv, ds, dout, n_seq, n_steps, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(double) * cells, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(MRS_ERR_HIP, std::string("pid probe: ") + hipGetErrorString(e));
  return MRS_OK;
}

int mrs_debug_pid_update(int32_t device_id, int32_t arith, int32_t n, const double* params, double* state, const double* err, const double* dt,
                         double* out) {
  if (n < 0 || !params || !state || !err || !dt || !out) return fail(MRS_ERR_ARG, "bad pid update arguments");
  if (arith != MRS_ARITH_LITERAL && arith != MRS_ARITH_FAST) return fail(MRS_ERR_ARG, "unknown arithmetic flavour");
  if (n == 0) return MRS_OK;
  if (device_id >= 0) HIPCHK(hipSetDevice(device_id));
  double* d = nullptr;  // params 5n | state 2n | err n | dt n | out n
  HIPCHK(hipMalloc(&d, sizeof(double) * (size_t)n * 10));
  double *dp = d, *ds = dp + (size_t)n * 5, *de = ds + (size_t)n * 2, *dd = de + n, *dout = dd + n;
  hipError_t e = hipMemcpy(dp, params, sizeof(double) * (size_t)n * 5, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(ds, state, sizeof(double) * (size_t)n * 2, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(de, err, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dd, dt, sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess)
    e = arith == MRS_ARITH_FAST ? mrs_launch_pid_update_probe_fast(dp, ds, de, dd, dout, n, nullptr) : mrs_launch_pid_update_probe_literal(dp, ds, de, dd, dout, n, nullptr);
  if (e == hipSuccess) e = hipMemcpy(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(state, ds, sizeof(double) * (size_t)n * 2, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(MRS_ERR_HIP, std::string("pid update: ") + hipGetErrorString(e));
  return MRS_OK;
}

int mrs_swarm_debug_component(mrs_swarm_t* s, int32_t component, int32_t first, int32_t count, const double* in, int32_t in_stride, double* out,
                              int32_t out_stride, double dt) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  static const int in_w[11]  = {0, 9, 18, 4, 3, 3, 4, 4, 10, 5, 4};
  static const int out_w[11] = {0, 9, 18, 8, 3, 3, 10, 5, 4, 4, 4};
  if (component < MRS_COMP_REORTH || component > MRS_COMP_RATE) return fail(MRS_ERR_ARG, "unknown component");
  if (!in || !out || in_stride < in_w[component] || out_stride < out_w[component] || !(dt > 0)) return fail(MRS_ERR_ARG, "bad component arguments");
  if (count == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  if ((rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001))) return rc;
  double*      d   = nullptr;
  const size_t nin = (size_t)count * in_stride, nout = (size_t)count * out_stride;
  HIPCHK(hipMalloc(&d, sizeof(double) * (nin + nout)));
  hipError_t e = hipMemcpyAsync(d, in, sizeof(double) * nin, hipMemcpyHostToDevice, s->stream);
  if (e == hipSuccess) e = hipMemsetAsync(d + nin, 0, sizeof(double) * nout, s->stream);
  if (e == hipSuccess)
    e = s->arith == MRS_ARITH_FAST ? mrs_launch_component_probe_fast(s->view(), component, first, count, d, in_stride, d + nin, out_stride, dt, s->stream)
                                   : mrs_launch_component_probe_literal(s->view(), component, first, count, d, in_stride, d + nin, out_stride, dt, s->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out, d + nin, sizeof(double) * nout, hipMemcpyDeviceToHost, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(MRS_ERR_HIP, std::string("component probe: ") + hipGetErrorString(e));
  return MRS_OK;
}

int mrs_swarm_debug_collision_words(mrs_swarm_t* s, uint32_t* out8) {
  MRS_ENTER(s);
  if (!s || !out8) return fail(MRS_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(mrs_collide_debug_words(s->cwork, s->stream, out8));
  return MRS_OK;
}

}  // extern "C"
