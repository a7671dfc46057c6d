// swarm_host.hip — host side of libmrs_swarm.so: the C ABI of include/mrs_swarm.h.
//
// Owns the device SoA state of one swarm shard, the interned per-airframe type table, the HIP stream all
// launches go to, and the bookkeeping that turns the reference's per-UAV setters into column uploads.
// No CPU fallback exists: without a usable HIP device mrs_swarm_create fails.
#include <hip/hip_runtime.h>

#include <math.h>
#include <unistd.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <dlfcn.h>
#include <string>
#include <vector>

#include "../../include/mrs_swarm.h"
#include "swarm_layout.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

extern "C" hipError_t mrs_launch_step_literal(SwarmDev sw, double dt, int substeps, int cascade, int blk0, int nblk, int with_mixed, hipStream_t st);
extern "C" hipError_t mrs_launch_step_fast(SwarmDev sw, double dt, int substeps, int cascade, int blk0, int nblk, int with_mixed, hipStream_t st);
extern "C" hipError_t mrs_launch_pid_probe_literal(const double*, const double*, const double*, const double*, const double*, double*, int, int, hipStream_t);
extern "C" hipError_t mrs_launch_pid_probe_fast(const double*, const double*, const double*, const double*, const double*, double*, int, int, hipStream_t);
extern "C" hipError_t mrs_launch_pid_update_probe_literal(const double*, double*, const double*, const double*, double*, int, hipStream_t);
extern "C" hipError_t mrs_launch_pid_update_probe_fast(const double*, double*, const double*, const double*, double*, int, hipStream_t);
extern "C" hipError_t mrs_launch_component_probe_literal(SwarmDev, int, int, int, const double*, int, double*, int, double, hipStream_t);
extern "C" hipError_t mrs_launch_component_probe_fast(SwarmDev, int, int, int, const double*, int, double*, int, double, hipStream_t);
// collide.hip
extern "C" hipError_t mrs_launch_flags_update(uint32_t* F, int first, int count, uint32_t and_mask, uint32_t or_mask, hipStream_t st);
extern "C" hipError_t mrs_launch_pack_positions(SwarmDev sw, PosRecord* out, hipStream_t st);
struct CollideWork;
extern "C" hipError_t mrs_collide_run(SwarmDev sw, CollideWork** work, const PosRecord* rec, long long n_total, long long my_offset,
                                      int crash, double rebounce, int rec_is_local_scratch, hipStream_t st);
extern "C" hipError_t mrs_collide_run_lists(SwarmDev sw, CollideWork** work, int crash, double rebounce, int force_rebuild, unsigned guard_tau,
                                            hipStream_t st);
extern "C" hipError_t mrs_collide_run_lists_gathered(SwarmDev sw, CollideWork** work, const PosRecord* rec, long long n_total, long long my_offset,
                                                     int crash, double rebounce, int force_rebuild, hipStream_t st);
extern "C" hipError_t mrs_collide_export_prepare(SwarmDev sw, CollideWork** work, int world, long long cap, hipStream_t st);
extern "C" long long  mrs_collide_export_capacity(const CollideWork* w);
extern "C" void*      mrs_collide_export_send(const CollideWork* w);
extern "C" void*      mrs_collide_export_recv(const CollideWork* w);
extern "C" hipError_t mrs_collide_export_mark(SwarmDev sw, CollideWork* w, long long n_max, int rank, uint32_t* map_send, hipStream_t st);
extern "C" hipError_t mrs_collide_export_translate(SwarmDev sw, CollideWork* w, long long n_max, int rank, const uint32_t* maps, const PosRecord* rec_all,
                                                   hipStream_t st);
extern "C" hipError_t mrs_collide_export_dev(const SwarmDev* sw, CollideWork* w, long long my_offset, unsigned tau, int eval, int crash, double rebounce,
                                             CollDev* cd);
extern "C" hipError_t mrs_collide_export_eval(SwarmDev sw, CollDev cd, hipStream_t st);
extern "C" hipError_t mrs_collide_export_fold_stall(CollideWork* w, unsigned progress_tau, hipStream_t st);
extern "C" hipError_t mrs_collide_fused_words(const CollideWork* w, hipStream_t st, unsigned* out8);
extern "C" void       mrs_collide_invalidate_gathered(CollideWork* w);
extern "C" void mrs_collide_step_hook(const CollideWork* w, const PosRecord** rec, uint32_t** flag, double* lim2);
extern "C" hipError_t mrs_collide_rebuilds(const CollideWork* w, hipStream_t st, unsigned* out);
extern "C" hipError_t mrs_collide_debug_words(const CollideWork* w, hipStream_t st, unsigned* out8);
extern "C" void mrs_collide_free(CollideWork* w);
extern "C" hipError_t mrs_launch_step_coll_literal(SwarmDev sw, CollDev cd, double dt, int variant, int grid_blocks, hipStream_t st);
extern "C" hipError_t mrs_launch_step_coll_fast(SwarmDev sw, CollDev cd, double dt, int variant, int grid_blocks, hipStream_t st);
extern "C" void       mrs_collide_export_part(CollDev* cd, int part, unsigned n_bnd, double dt, int announce);
extern "C" hipError_t mrs_collide_handoff_init(CollideWork* w, int n, unsigned tau, hipStream_t st);
extern "C" const uint32_t* mrs_collide_ctl_words(const CollideWork* w);
extern "C" hipError_t mrs_collide_heads_to_host(CollideWork* w, const uint32_t* maps, long long stride, int world, const uint32_t** out, hipStream_t st);
extern "C" hipError_t mrs_collide_fused_dev(const SwarmDev* sw, CollideWork* w, unsigned tau, int eval, int crash, double rebounce, CollDev* cd);
extern "C" void mrs_collide_fused_advance(CollideWork* w);
extern "C" const volatile unsigned* mrs_collide_host_words(const CollideWork* w);
extern "C" hipError_t mrs_collide_fused_reset(CollideWork* w, hipStream_t st);
extern "C" hipError_t mrs_collide_latch_force(SwarmDev sw, CollideWork* w, int pin, int crash, double rebounce, hipStream_t st);
extern "C" int        mrs_collide_fused_pin(const CollideWork* w);
// outputs.hip
extern "C" hipError_t mrs_launch_timeout_input(SwarmDev sw, int first, int count, hipStream_t st);
extern "C" hipError_t mrs_launch_unpack_rows(SwarmDev sw, const double* rows, int stride, int width, int base, int first, int count, hipStream_t st);
extern "C" hipError_t mrs_launch_pack_outputs(SwarmDev sw, int first, int count, mrs_uav_output_t* dev_out, hipStream_t st);

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) return fail(MRS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

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

struct TypeKey {  // everything that distinguishes two UavSystem parameterisations
  mrs_model_params_t    mp;  // takeoff_patch_enabled normalised to 0 (it is per-UAV mutable state)
  mrs_mixer_params_t    mixer;
  mrs_rate_params_t     rate;
  mrs_attitude_params_t att;
  mrs_velocity_params_t vel;
  mrs_position_params_t pos;
};

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

// ------------------------------------------------------------------------------------------------
// swarm object
// ------------------------------------------------------------------------------------------------

// RCCL entry points, bound at run time by rccl_load() below
namespace {
struct NcclId { char internal[128]; };  // ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128, passed BY VALUE to ncclCommInitRank
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclId*)                                           = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int)                         = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*)                                             = nullptr;
  int (*CommCount)(void*, int*)                                         = nullptr;
  const char* (*GetErrorString)(int)                                    = nullptr;
};
RcclApi g_rccl;
}  // namespace

struct mrs_swarm {
  // every C-ABI call on a swarm is serialised (the reference's subscriber callbacks run concurrently with timerMain and are
  // serialised by mutex_uav_system_, src/uav_system_ros.cpp:267,702): recursive because entry points call each other
  std::recursive_mutex mtx;
  int32_t  n = 0, npad = 0, device = 0;
  int32_t  arith = MRS_ARITH_LITERAL;
  hipStream_t stream = nullptr;
  // a run of steps without collisions in between is issued as two half-swarm launches per step on two streams: the halves are
  // independent, so the drain of one launch overlaps the ramp of the other (tools/two_streams.py: +11 % at 100 k, +19 % at 200 k)
  hipStream_t stream2 = nullptr;
  // Split sharded ticks with reserved compute units (MRS_SPLIT_CU_RESERVE = R > 0; tools/cu_mask_probe.hip): the boundary launch and the
  // collective run on `stream_b`, whose queue may use R CUs only (mask bits 0..R-1: bit i is CU i / 8 of XCD i % 8), the interior
  // launch on `stream_i`, whose queue uses all the others — the lone waves of the boundary chain no longer share SIMDs with the
  // streaming interior waves.  `cstream`: where collectives and boundary launches go right now (`stream` outside split segments).
  hipStream_t stream_b = nullptr, stream_i = nullptr, cstream = nullptr;
  hipEvent_t  ev_join_b = nullptr;
  int         cu_reserve = 0;
  hipEvent_t  ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t  ev_end2 = nullptr;   // profiling: end of the second stream's part of a split run (recorded before the join)
  bool        prof_split = false;  // the end events of the running profile region have been recorded by the split run itself
  // every C-ABI call bumps op_seq (MRS_LOCK); mrs_swarm_synchronize notes the value it leaves behind: a run of steps that is the very
  // next call finds both streams idle and starts its second stream without the fork event
  uint64_t    op_seq = 0, quiet_seq = ~0ull;
  mutable bool stream_exported = false;  // mrs_swarm_stream has handed the stream to the caller, who may enqueue work behind the library's back
  bool        split_steps = true;  // tuning: MRS_SPLIT_STREAMS=0
  // native multi-GPU collision exchange (mrs_swarm_comm_init): RCCL all-gather issued on `stream`
  void*      rccl_comm = nullptr;
  int        comm_world = 0, comm_rank = 0;   // comm_world > 0: a communicator of some kind is bound
  int64_t    comm_n_total = 0, comm_n_max = 0;
  PosRecord* comm_send = nullptr;
  PosRecord* comm_recv = nullptr;
  // collective backend other than RCCL: a caller-supplied all-gather (mrs_swarm_comm_init_custom) or an in-process group of
  // swarms driven by one host thread each (mrs_swarm_comm_init_loopback)
  mrs_allgather_fn      comm_fn = nullptr;
  void*                 comm_user = nullptr;
  mrs_loopback_group_t* comm_group = nullptr;
  // measurement stand-in (mrs_swarm_comm_init_standin): ONE rank of `world` alone on the device; its neighbours in the slab order
  // are images of itself one slab width away, and every collective costs a fixed latency
  bool   comm_standin = false;
  double standin_delay_us = 0.0, standin_width = 0.0;
  // peer-window exchange (mrs_swarm_peer_window_create / mrs_swarm_comm_init_peer; collide.hip k_peer_allgather): ranks write their
  // blocks straight into each other's device memory, one kernel per collective on the swarm's stream, no collective library
  bool               comm_peer = false;
  void*              peer_window = nullptr;       // this rank's window (fine-grained device memory)
  size_t             peer_slot_bytes = 0, peer_window_bytes = 0;
  int                peer_world = 0, peer_rank = 0;
  int64_t            peer_n_total = 0;
  MrsPeerWindows     peer_windows{};              // every rank's window as this process addresses it
  std::vector<void*> peer_opened;                 // the ones mapped here through an IPC handle (closed by comm_destroy)
  unsigned           peer_seq = 0, peer_tickets = 0;
  unsigned*          peer_ticket = nullptr;       // device words, one per peer: blocks of all exchange kernels so far that have pushed their share for it
  unsigned*          peer_err = nullptr;          // pinned host word: an exchange kernel waited in vain for a peer
  // export-set exchange: between two searches only boundary UAVs travel
  int       exchange = MRS_EXCHANGE_EXPORT_SETS;
  bool      x_ok = false;           // export lists are live: the next tick can be a fused launch + export-set all-gather
  int       x_fallback_left = 0;    // ticks to stay on the full exchange after an incomplete (overflowing) search
  uint32_t* x_map_send = nullptr;   // [2 + n_max] slot map of this rank
  uint32_t* x_map_recv = nullptr;   // [world][2 + n_max]
  int64_t   x_export_count = 0;
  std::vector<unsigned> x_last_overflow;
  int64_t   x_searches = 0, x_ticks = 0, x_noop_ticks = 0;
  // split sharded ticks: between two searches the blocks that hold a boundary UAV are stepped by a small launch on
  // `stream`, followed there by the collective, while the interior launch runs on `stream2` and never waits for a collective
  bool      shard_split = true;     // tuning: MRS_SHARD_SPLIT=0 keeps every tick in the serial form (fused launch, then the collective)
  int       split_min_blocks = 512; // tuning / tests: MRS_SHARD_SPLIT_MIN_BLOCKS
  double    split_max_fraction = 0.25;  // ... and MRS_SHARD_SPLIT_MAX_FRACTION: the boundary launch may cover at most this share of the blocks
  uint32_t  x_nbnd = 0;             // boundary blocks of this rank as of the last search
  double    x_dt = -1.0;            // dt of the previous call: the announcements of its last launches assumed it
  int       resident_waves = 2048;  // wave slots of the device at the interior kernel's occupancy (2 per SIMD): see split_ok()
  int64_t   x_split_ticks = 0;
  // test hook (mrs_swarm_debug_chaos): this rank's host sleeps a random time before every launch of a sharded tick and, half of
  // the time, decides on the stall / warning words as it read them one launch earlier (still within what the protocol guarantees)
  int       chaos_max_us = 0;
  uint64_t  chaos_state = 0x9E3779B97F4A7C15ull;
  unsigned  chaos_T = 0, chaos_W = 0;
  double*   dS = nullptr;
  uint32_t* dF = nullptr;
  TypeParams* dT = nullptr;
  int32_t   dT_cap = 0;
  unsigned long long* dDiag = nullptr;
  std::vector<TypeKey>    keys;
  std::vector<TypeParams> tparams;
  std::map<std::string, int> key_index;
  std::vector<uint16_t>   uav_type;
  std::vector<uint8_t>    uav_mode;   // host mirror of active_input_: picks the kernel variant
  int64_t n_cascade = 0;              // UAVs whose mode needs the controller cascade
  bool   types_dirty = true;
  double table_dt    = -1.0;
  // publisher payloads: device pack buffer + pinned host staging
  mrs_uav_output_t* dOut = nullptr;
  mrs_uav_output_t* hOut = nullptr;
  int32_t           out_cap = 0;
  // staged command upload: pinned host rows + device copy
  double* hIn = nullptr;
  double* dIn = nullptr;
  int64_t in_cap = 0;  // doubles
  // collision scratch
  PosRecord*   dRec = nullptr;
  CollideWork* cwork = nullptr;
  bool         use_lists = true;   // single-GPU collision ticks reuse neighbour lists between rebuilds (tuning: MRS_NEIGHBOUR_LISTS=0)
  bool         nbr_dirty = true;   // the host wrote positions or airframe constants since the last collision tick
  int64_t      collision_ticks = 0;
  // Lazily evaluated collision ticks (single-GPU neighbour lists).  handleCollisions is not launched when it is called: it is
  // evaluated by the NEXT makeStep launch, whose prologue forms the forces from the neighbour lists (step_device.inc *_coll), or
  // by settle() when the host looks at the swarm first.  `log` holds the launches the device has not confirmed yet: when a UAV
  // leaves its skin during step T the launches after T turn into no-ops, and the host repeats the search and replays them.
  struct Collide { bool on = false; int enabled = 0, crash = 0; double rebounce = 0.0; };
  // one fused launch: the collision tick it evaluates first (searched: a search queued right before it has done that), then makeStep(dt)
  struct TickRec { double dt; Collide eval; bool searched; int pin; };  // pin: which position buffer the launch read
  Collide              pend;                        // requested after the most recent step, not evaluated yet
  // A fused launch consumes the force it evaluates from registers and does not write the F_ext columns (24 B per UAV and tick).
  // While f_lazy.on those columns are stale: the latched force is "collision tick f_lazy on the position records f_lazy_pin",
  // re-derived by settle() (or overwritten by the next search) before anything reads the columns.
  Collide              f_lazy;
  int                  f_lazy_pin = 0;
  bool                 collide_since_step = false;  // ... or evaluated already: either way the next step keeps the fused form
  bool                 p_valid = false;             // the position records hold the positions after the most recent step
  bool                 fk_ok   = false;             // the lists are complete (no UAV over the list capacity) and in local mode
  unsigned             last_overflow = 0;
  std::vector<TickRec> log;
  uint32_t             tau = 0;                     // tick index of the last fused launch since the stream was last drained
  bool                 use_fused = true;            // tuning: MRS_FUSED_COLLISIONS=0 launches every collision tick on its own
  int                  fused_lead = 3;              // launches the host may run ahead of the device (MRS_FUSED_LEAD; 2-3 measured best, 8: stalls)
  uint32_t             search_mark = 0;             // tick index behind which the last ahead-of-time search was queued
  int64_t              n_ahead_searches = 0;
  int64_t              n_stalls = 0, n_noop_launches = 0, n_fused = 0;
  // profiling
  int  profiling = 0;  // 0 off, 1 one event pair around the whole step_n/tick_n region, 2 one pair per step launch
  std::vector<hipEvent_t> ev;
  int  ev_used = 0;
  int  region_launches = 0;
  double last_ms = 0.0;
  int    last_launches = 0;
  std::vector<double>   stage;  // host staging column
  std::vector<uint32_t> stage_u;

  // per-64-block airframe type (0xFFFF = mixed) and the list of mixed blocks
  std::vector<uint32_t> block_type;
  std::vector<int32_t>  mixed_blocks;
  uint32_t* dBT = nullptr;
  int32_t*  dMB = nullptr;
  bool      blocks_dirty = true;

  bool      fext_active = false;  // apply_force / collisions were used at least once

  SwarmDev view() const {
    SwarmDev v{dS, dF, dT, dDiag, dBT, dMB, n, npad, (int32_t)mixed_blocks.size(), fext_active ? 1u : 0u, nullptr, nullptr, 0.0, 0, arith == MRS_ARITH_FAST ? 1 : 0};
    mrs_collide_step_hook(cwork, &v.vl_rec, &v.vl_flag, &v.vl_lim2);
    return v;
  }
};

static void track_mode(mrs_swarm* s, int first, int count, int mode) {
  for (int k = 0; k < count; k++) {
    uint8_t& m = s->uav_mode[(size_t)first + k];
    s->n_cascade += (mode >= MRS_CONTROL_GROUP_CMD) - (m >= MRS_CONTROL_GROUP_CMD);
    m = (uint8_t)mode;
  }
}

#define MRS_LOCK(s)                                                                      \
  std::unique_lock<std::recursive_mutex> _lk;                                            \
  if (s) {                                                                               \
    _lk = std::unique_lock<std::recursive_mutex>(const_cast<mrs_swarm*>(s)->mtx);        \
    const_cast<mrs_swarm*>(s)->op_seq++;                                                 \
  }

static int settle(mrs_swarm* s);
// entry of every call that reads or writes swarm state: collision ticks still pending on the device side are evaluated first
#define MRS_ENTER(s)                                          \
  MRS_LOCK(s);                                                \
  if (s) {                                                    \
    int _src = settle(const_cast<mrs_swarm*>(s));             \
    if (_src) return _src;                                    \
  }

static int check_range(const mrs_swarm* s, int first, int count) {
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (first < 0 || count < 0 || (long long)first + count > s->n) return fail(MRS_ERR_RANGE, "uav range out of bounds");
  return MRS_OK;
}

static int intern_type(mrs_swarm* s, const TypeKey& k, int* out) {
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

static TypeKey make_key(const mrs_model_params_t* p) {
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

static int upload_blocks(mrs_swarm* s) {
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

static int upload_types(mrs_swarm* s, double dt) {
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
static int put_column(mrs_swarm* s, int f, int first, int count, const double* col) {
  if (f >= F_X && f < F_X + 3) s->nbr_dirty = true;
  HIPCHK(hipMemcpyAsync(s->dS + (size_t)f * s->npad + first, col, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  return MRS_OK;
}
static int fill_column(mrs_swarm* s, int f, int first, int count, double value) {
  if (value == 0.0) {
    HIPCHK(hipMemsetAsync(s->dS + (size_t)f * s->npad + first, 0, sizeof(double) * (size_t)count, s->stream));
    return MRS_OK;
  }
  s->stage.assign((size_t)count, value);
  return put_column(s, f, first, count, s->stage.data());
}
// host AoS (count x width, element j) -> device column
static int put_strided(mrs_swarm* s, int f, int first, int count, const double* src, int width, int j) {
  s->stage.resize((size_t)count);
  for (int k = 0; k < count; k++) s->stage[(size_t)k] = src[(size_t)k * width + j];
  return put_column(s, f, first, count, s->stage.data());
}
static int get_strided(mrs_swarm* s, int f, int first, int count, double* dst, int width, int j) {
  s->stage.resize((size_t)count);
  HIPCHK(hipMemcpyAsync(s->stage.data(), s->dS + (size_t)f * s->npad + first, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost,
                        s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  for (int k = 0; k < count; k++) dst[(size_t)k * width + j] = s->stage[(size_t)k];
  return MRS_OK;
}
static int flags_update(mrs_swarm* s, int first, int count, uint32_t and_mask, uint32_t or_mask) {
  if (count <= 0) return MRS_OK;
  HIPCHK(mrs_launch_flags_update(s->dF, first, count, and_mask, or_mask, s->stream));
  return MRS_OK;
}

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

namespace {
void peer_release(mrs_swarm* s);
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
  if (s->hOut) (void)hipHostFree(s->hOut);
  if (s->hIn) (void)hipHostFree(s->hIn);
  if (s->dIn) (void)hipFree(s->dIn);
  if (s->dT) (void)hipFree(s->dT);
  if (s->dBT) (void)hipFree(s->dBT);
  if (s->dMB) (void)hipFree(s->dMB);
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
  MRS_ENTER(s);
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
  MRS_ENTER(s);
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

// ---- hot path ----
static int launch_part(mrs_swarm* s, double dt, int substeps, int blk0, int nblk, int with_mixed, hipStream_t st) {
  const int variant = s->n_cascade > 0 ? 0 : 1;  // 0 all input modes | 1 model only
  if (s->arith == MRS_ARITH_FAST)
    HIPCHK(mrs_launch_step_fast(s->view(), dt, substeps, variant, blk0, nblk, with_mixed, st));
  else
    HIPCHK(mrs_launch_step_literal(s->view(), dt, substeps, variant, blk0, nblk, with_mixed, st));
  return MRS_OK;
}

static int launch_step(mrs_swarm* s, double dt, int substeps) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  s->region_launches++;
  if (s->profiling == 2) {
    while ((int)s->ev.size() < s->ev_used + 2) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      s->ev.push_back(e);
    }
    e0 = s->ev[(size_t)s->ev_used];
    e1 = s->ev[(size_t)s->ev_used + 1];
    s->ev_used += 2;
    HIPCHK(hipEventRecord(e0, s->stream));
  }
  int rc = launch_part(s, dt, substeps, 0, (s->n + 63) / 64, 1, s->stream);
  if (rc) return rc;
  if (s->profiling == 2) HIPCHK(hipEventRecord(e1, s->stream));
  return MRS_OK;
}

// one step as two independent half-swarm launches, one per stream (between fork_streams and join_streams)
static int launch_step_split(mrs_swarm* s, double dt, int substeps) {
  s->region_launches++;
  const int nb = (s->n + 63) / 64, half = nb / 2;
  int rc = launch_part(s, dt, substeps, 0, half, 1, s->stream);
  if (rc) return rc;
  return launch_part(s, dt, substeps, half, nb - half, 0, s->stream2);
}
static int fork_streams(mrs_swarm* s) {
  HIPCHK(hipEventRecord(s->ev_fork, s->stream));
  HIPCHK(hipStreamWaitEvent(s->stream2, s->ev_fork, 0));
  return MRS_OK;
}
static int join_streams(mrs_swarm* s) {
  HIPCHK(hipEventRecord(s->ev_join, s->stream2));
  HIPCHK(hipStreamWaitEvent(s->stream, s->ev_join, 0));
  return MRS_OK;
}

static int begin_profile(mrs_swarm* s) {
  s->ev_used          = 0;
  s->region_launches  = 0;
  s->prof_split       = false;
  if (s->profiling == 1) {
    while (s->ev.size() < 2) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      s->ev.push_back(e);
    }
    HIPCHK(hipEventRecord(s->ev[0], s->stream));
  }
  return MRS_OK;
}

static int finish_profile(mrs_swarm* s) {
  if (s->profiling) {  // the timed region ends when every tick of it has really run (replays and the last collision tick included)
    int rc = settle(s);
    if (rc) return rc;
  }
  if (s->profiling == 1) {
    // a split run has recorded its own end events, one per stream, before joining the streams: the region ends when the later of
    // the two halves has finished its last step (the join is stream bookkeeping, not part of the steps)
    if (!s->prof_split) HIPCHK(hipEventRecord(s->ev[1], s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
    if (s->prof_split) {
      float ms2 = 0;
      HIPCHK(hipEventElapsedTime(&ms2, s->ev[0], s->ev_end2));
      if (ms2 > ms) ms = ms2;
      s->prof_split = false;
    }
    s->last_launches = s->region_launches;
    s->last_ms       = s->region_launches ? (double)ms / s->region_launches : 0.0;
    return MRS_OK;
  }
  if (!s->profiling || s->ev_used == 0) return MRS_OK;
  HIPCHK(hipStreamSynchronize(s->stream));
  double total = 0;
  for (int k = 0; k < s->ev_used; k += 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev[(size_t)k], s->ev[(size_t)k + 1]));
    total += ms;
  }
  s->last_launches = s->ev_used / 2;
  s->last_ms       = total / s->last_launches;
  s->ev_used       = 0;
  return MRS_OK;
}

// ------------------------------------------------------------------------------------------------
// lazily evaluated collision ticks (see the `pend` / `log` members of mrs_swarm)
// ------------------------------------------------------------------------------------------------
// handleCollisions launched on its own: pack + insert / list evaluation, then the query (collide.hip) — the device decides
// whether the search has to be repeated, unless `force` says so
static int collide_now(mrs_swarm* s, const mrs_swarm::Collide& c, bool force) {
  int rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001);
  if (rc) return rc;
  HIPCHK(mrs_collide_run_lists(s->view(), &s->cwork, c.crash, c.rebounce, (force || s->nbr_dirty) ? 1 : 0, 0u, s->stream));
  s->nbr_dirty = false;
  s->f_lazy.on = false;  // the pass latched this tick's force
  s->p_valid = true;  // the pass refreshed the position records
  if (!s->use_fused) return MRS_OK;  // (every tick on its own: nobody needs to know, the call stays asynchronous)
  // the host must know whether the lists are complete before a step kernel may evaluate a tick from them
  unsigned w[8];
  HIPCHK(mrs_collide_debug_words(s->cwork, s->stream, w));  // (synchronises the stream)
  s->fk_ok         = w[6] == s->last_overflow;  // no UAV over the list capacity in this pass
  s->last_overflow = w[6];
  return MRS_OK;
}

static bool fused_usable(const mrs_swarm* s) {
  return s->use_lists && s->use_fused && s->fk_ok && s->p_valid && !s->nbr_dirty && !s->blocks_dirty && !s->types_dirty && s->cwork != nullptr;
}

// The host runs at most `lead` launches ahead of its device: it waits (spinning on the pinned progress word, no synchronisation)
// until launch `index - lead` has started or some launch has reported stale lists.  A device that makes no progress for
// MRS_PROGRESS_TIMEOUT_S seconds (default 30; a wedged kernel, a collective whose peer is gone) is an error, returned with the words
// the host last saw — after it the swarm's stream, and for a sharded swarm its communicator, must be considered dead: destroy the
// swarm from a fresh process (never re-exec a process that has touched the GPU).
// the stall / warning index the host knows of: each chain of a split tick keeps mirrors of its own (one writer per word)
static inline unsigned min_nonzero(unsigned a, unsigned b) { return a == 0u ? b : (b == 0u ? a : (a < b ? a : b)); }
static inline unsigned stall_word(const volatile unsigned* hw) { return min_nonzero(hw[CTL_STALL], hw[CTL_STALL2]); }
static inline unsigned warn_word(const volatile unsigned* hw) { return min_nonzero(hw[CTL_WARN], hw[CTL_WARN2]); }

static int wait_for_progress(mrs_swarm* s, const volatile unsigned* hw, unsigned index, int lead) {
  if (!hw) return MRS_OK;
  // (launches after a stall index T are no-ops and report no progress: once launch T has started nothing more will come.  A sharded
  //  swarm ANNOUNCES stall indices ahead of time, so a known T does not end the waiting by itself — an earlier one may still turn up,
  //  and the host must not outrun what it has seen)
  auto behind = [&]() {
    const unsigned T = stall_word(hw), P = hw[CTL_PROGRESS];
    return (int)(index - P) > lead && !(T != 0u && P >= T);
  };
  if (!behind()) return MRS_OK;
  static const double limit_s = getenv("MRS_PROGRESS_TIMEOUT_S") ? atof(getenv("MRS_PROGRESS_TIMEOUT_S")) : 30.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long spins = 1; behind(); spins++) {
    __builtin_ia32_pause();
    if ((spins & 0xFFFFul) != 0) continue;
    // a launch that failed asynchronously never writes its words — on whichever stream of a split tick it ran
    for (hipStream_t st : {s->stream, s->stream2, s->stream_i, s->stream_b}) {
      if (!st) continue;
      const hipError_t q = hipStreamQuery(st);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(MRS_ERR_HIP, std::string("fused launches: ") + hipGetErrorString(q));
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
      return fail(MRS_ERR_HIP, "the device made no progress for " + std::to_string((int)limit_s) + " s: waiting for launch " + std::to_string(index - (unsigned)lead) +
                                   ", progress word " + std::to_string(hw[CTL_PROGRESS]) + ", stall word " + std::to_string(stall_word(hw)) + ", warning word " +
                                   std::to_string(warn_word(hw)) + " (the stream" + (s->comm_world > 1 ? " and the communicator are" : " is") + " dead: use a fresh process)");
  }
  return MRS_OK;
}

// one fused launch: evaluate collision tick `e.eval` (if any) from the lists, then makeStep(e.dt)
static int launch_fused(mrs_swarm* s, const mrs_swarm::TickRec& e) {
  const volatile unsigned* hw = mrs_collide_host_words(s->cwork);
  // do not run further ahead of the device than a few launches: when the lists go stale at tick T everything queued behind T is wasted
  if (int rcw = wait_for_progress(s, hw, s->tau, s->fused_lead)) return rcw;
  if (e.dt != s->table_dt) {  // (a replayed tick of another dt: the motor-filter constants of the type table follow)
    int rc = upload_types(s, e.dt);
    if (rc) return rc;
  }
  CollDev cd;
  SwarmDev v = s->view();
  HIPCHK(mrs_collide_fused_dev(&v, s->cwork, s->tau + 1, (e.eval.on && !e.searched) ? 1 : 0, e.eval.crash, e.eval.rebounce, &cd));
  const int variant = s->n_cascade > 0 ? 0 : 1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  s->region_launches++;
  if (s->profiling == 2) {
    while ((int)s->ev.size() < s->ev_used + 2) {
      hipEvent_t ev;
      HIPCHK(hipEventCreate(&ev));
      s->ev.push_back(ev);
    }
    e0 = s->ev[(size_t)s->ev_used];
    e1 = s->ev[(size_t)s->ev_used + 1];
    s->ev_used += 2;
    HIPCHK(hipEventRecord(e0, s->stream));
  }
  if (s->arith == MRS_ARITH_FAST)
    HIPCHK(mrs_launch_step_coll_fast(v, cd, e.dt, variant, 0, s->stream));
  else
    HIPCHK(mrs_launch_step_coll_literal(v, cd, e.dt, variant, 0, s->stream));
  if (s->profiling == 2) HIPCHK(hipEventRecord(e1, s->stream));
  mrs_swarm::TickRec rec = e;
  rec.pin = mrs_collide_fused_pin(s->cwork);
  if (e.eval.on) {
    s->f_lazy.on = !e.searched;  // (a search queued right before the launch latched the force itself)
    if (!e.searched) {
      s->f_lazy     = e.eval;
      s->f_lazy_pin = rec.pin;
    }
  }
  mrs_collide_fused_advance(s->cwork);
  s->tau++;
  s->n_fused++;
  s->log.push_back(rec);
  if (e.eval.on) s->fext_active = true;
  return MRS_OK;
}

// Wait for the device and make good for launches that turned into no-ops: if the lists went stale during step T (a UAV left its
// skin), repeat the search on the state after step T — which also evaluates the collision tick that followed step T — and issue
// the ticks after T again.  Returns with an empty log.
static int drain(mrs_swarm* s) {
  if (s->log.empty()) return MRS_OK;
  const volatile unsigned* hw = mrs_collide_host_words(s->cwork);
  for (;;) {
    HIPCHK(hipStreamSynchronize(s->stream));
    const unsigned T = hw ? hw[CTL_STALL] : 0u;
    if (T == 0u || T > s->log.size()) {
      s->log.clear();
      s->tau = 0;
      s->search_mark = 0;
      if (T) return fail(MRS_ERR_HIP, "collision lists: stall index beyond the launch log");
      HIPCHK(mrs_collide_fused_reset(s->cwork, s->stream));  // (progress word back to 0 with tau)
      return MRS_OK;
    }
    s->n_stalls++;
    s->n_noop_launches += (int64_t)s->log.size() - T;
    {  // launch T was the last one that ran.  If it evaluated a collision tick, that force is the latched one (not written: see
       // f_lazy); a launch without evaluation only ever follows a pass that wrote the columns
      const mrs_swarm::TickRec& last = s->log[T - 1];
      s->f_lazy.on = last.eval.on && !last.searched;
      if (s->f_lazy.on) {
        s->f_lazy     = last.eval;
        s->f_lazy_pin = last.pin;
      }
    }
    std::vector<mrs_swarm::TickRec> tail(s->log.begin() + T, s->log.end());
    for (auto& e : tail) e.searched = false;  // (a search queued ahead of time behind the stalled launch did nothing)
    s->log.clear();
    s->tau = 0;
    s->search_mark = 0;
    HIPCHK(mrs_collide_fused_reset(s->cwork, s->stream));
    // the collision tick that followed step T: the first replayed launch was going to evaluate it, or it is the pending one
    mrs_swarm::Collide& c = tail.empty() ? s->pend : tail[0].eval;
    if (c.on) {
      int rc = collide_now(s, c, /*force=*/true);
      if (rc) return rc;
      c.on = false;
    } else {
      s->fk_ok = false;  // nobody needs the lists right now: the next collision tick starts with a search
      s->nbr_dirty = true;
    }
    for (const auto& e : tail) {
      int rc;
      if (fused_usable(s)) {
        if ((rc = launch_fused(s, e))) return rc;
      } else {  // (lists incomplete: dense neighbourhoods) every tick on its own
        if (e.eval.on && (rc = collide_now(s, e.eval, false))) return rc;
        s->p_valid = false;
        s->region_launches++;
        if ((rc = launch_part(s, e.dt, 1, 0, (s->n + 63) / 64, 1, s->stream))) return rc;
      }
    }
    if (s->log.empty()) return MRS_OK;
  }
}

// everything the caller asked for so far has happened on the device (asynchronously at most the plain launches)
static int settle(mrs_swarm* s) {
  if (s->log.empty() && !s->pend.on && !s->f_lazy.on) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int rc = drain(s);
  if (rc) return rc;
  if (s->pend.on) {
    const mrs_swarm::Collide c = s->pend;
    s->pend.on = false;
    if ((rc = collide_now(s, c, false))) return rc;
  } else if (s->f_lazy.on) {  // nothing newer overwrites the force the last fused launch evaluated: write it out now
    if ((rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001))) return rc;
    HIPCHK(mrs_collide_latch_force(s->view(), s->cwork, s->f_lazy_pin, s->f_lazy.crash, s->f_lazy.rebounce, s->stream));
    s->f_lazy.on = false;
  }
  return MRS_OK;
}

// one makeStep of every UAV; the collision tick requested since the previous step (if any) is evaluated by the same launch
static int step_one(mrs_swarm* s, double dt) {
  int rc;
  if (s->collide_since_step && s->use_lists && s->use_fused) {
    const volatile unsigned* hw = mrs_collide_host_words(s->cwork);
    if (hw && hw[CTL_STALL] != 0u && (rc = drain(s))) return rc;  // seen without synchronising: stop feeding no-ops
    if (s->pend.on && !fused_usable(s) && (rc = settle(s))) return rc;  // first tick / after host writes: the pass on its own
    if (fused_usable(s)) {
      mrs_swarm::TickRec e{dt, s->pend, false};
      if (s->pend.on && hw && hw[CTL_WARN] > s->search_mark) {
        // some UAV has used up most of its skin: repeat the search NOW, in stream order — it evaluates the pending collision tick
        // itself — instead of running into the stall a few ticks on (no synchronisation, nothing to replay)
        if ((rc = upload_types(s, dt))) return rc;
        HIPCHK(mrs_collide_run_lists(s->view(), &s->cwork, s->pend.crash, s->pend.rebounce, 1, s->tau, s->stream));  // (tau >= 1: the warning came from a launch of this log)
        e.searched     = true;
        s->search_mark = s->tau;
        s->n_ahead_searches++;
      }
      s->pend.on            = false;
      s->collide_since_step = false;
      return launch_fused(s, e);
    }
  }
  if ((rc = settle(s))) return rc;
  s->collide_since_step = false;
  s->p_valid            = false;  // a plain step kernel does not refresh the position records
  return launch_step(s, dt, 1);
}

int mrs_swarm_step_n(mrs_swarm_t* s, double dt, int32_t n_steps, int32_t substeps_per_launch) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (!(dt > 0) || n_steps < 0 || substeps_per_launch < 1) return fail(MRS_ERR_ARG, "bad step arguments");
  if (s->n == 0 || n_steps == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, dt);
  if (rc) return rc;
  // (a caller holding the handle of mrs_swarm_stream() may have enqueued work of its own without an ABI call: then the stream is
  //  asked — BEFORE the profile's start event goes onto it.  Only then: the query itself puts a marker on the stream, which costs
  //  the run that follows ~12 us — 0.6 us per step of a 20-step region, measured)
  const bool idle_at_entry = s->quiet_seq + 1 == s->op_seq && (!s->stream_exported || hipStreamQuery(s->stream) == hipSuccess);
  if ((rc = begin_profile(s))) return rc;
  bool enqueued = false;  // something of this call is already on the stream
  if (s->collide_since_step && substeps_per_launch == 1) {  // the first step of the run may carry a collision tick
    if ((rc = step_one(s, dt))) return rc;
    n_steps--;
    enqueued = true;
  }
  if (n_steps > 0) {
    enqueued = enqueued || !s->log.empty() || s->pend.on || s->f_lazy.on;  // (what settle is about to issue)
    if ((rc = settle(s))) return rc;
    s->p_valid = false;
    // enough launches to overlap, enough blocks for two useful halves, and no per-launch events to keep in order
    static const int split_min_blocks = getenv("MRS_SPLIT_MIN_BLOCKS") ? atoi(getenv("MRS_SPLIT_MIN_BLOCKS")) : 1024;  // tuning aid
    const bool split = s->split_steps && s->profiling != 2 && (s->n + 63) / 64 >= split_min_blocks && (n_steps + substeps_per_launch - 1) / substeps_per_launch >= 4;
    // (the call right before this one was mrs_swarm_synchronize and nothing has been enqueued since — not even by the lines above:
    //  both streams are idle (upload_types synchronises when it copies), the second one needs no event to wait for)
    const bool quiet = idle_at_entry && !enqueued;
    if (split && !quiet && (rc = fork_streams(s))) return rc;
    int left = n_steps;
    while (left > 0 && rc == MRS_OK) {
      const int sub = left < substeps_per_launch ? left : substeps_per_launch;
      rc = split ? launch_step_split(s, dt, sub) : launch_step(s, dt, sub);
      left -= sub;
    }
    if (split) {  // also on a failed launch: nothing else may touch the state before the second stream has been joined
      if (rc == MRS_OK && s->profiling == 1 && hipEventRecord(s->ev[1], s->stream) == hipSuccess && hipEventRecord(s->ev_end2, s->stream2) == hipSuccess)
        s->prof_split = true;
      const int rcj = join_streams(s);
      if (rc == MRS_OK) rc = rcj;
    }
    if (rc) return rc;
  }
  return finish_profile(s);
}

int mrs_swarm_step(mrs_swarm_t* s, double dt) {
  MRS_LOCK(s); return mrs_swarm_step_n(s, dt, 1, 1); }

int mrs_swarm_pack_positions(mrs_swarm_t* s, void** dev_ptr, int64_t* n_bytes) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001);
  if (rc) return rc;
  if (!s->dRec) HIPCHK(hipMalloc(&s->dRec, sizeof(PosRecord) * (size_t)s->npad));
  HIPCHK(mrs_launch_pack_positions(s->view(), s->dRec, s->stream));
  if (dev_ptr) *dev_ptr = s->dRec;
  if (n_bytes) *n_bytes = (int64_t)sizeof(PosRecord) * s->n;
  return MRS_OK;
}

int mrs_swarm_pack_positions_to(mrs_swarm_t* s, void* dev_dst) {
  MRS_ENTER(s);
  if (!s || !dev_dst) return fail(MRS_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001);
  if (rc) return rc;
  HIPCHK(mrs_launch_pack_positions(s->view(), (PosRecord*)dev_dst, s->stream));
  return MRS_OK;
}

int mrs_swarm_handle_collisions_gathered(mrs_swarm_t* s, const void* dev_records, int64_t n_total, int64_t my_offset, int32_t enabled,
                                         int32_t crash, double rebounce) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (!(crash || enabled)) return MRS_OK;  // src/multirotor_simulator.cpp:299-301
  if (!dev_records || n_total < s->n || my_offset < 0 || my_offset + s->n > n_total) return fail(MRS_ERR_ARG, "bad gathered-record arguments");
  if (s->n == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  s->fext_active = true;
  s->collision_ticks++;
  s->nbr_dirty = true;  // a later single-GPU tick starts from a rebuild
  if (s->use_lists)
    HIPCHK(mrs_collide_run_lists_gathered(s->view(), &s->cwork, (const PosRecord*)dev_records, n_total, my_offset, crash, rebounce, 0, s->stream));
  else
    HIPCHK(mrs_collide_run(s->view(), &s->cwork, (const PosRecord*)dev_records, n_total, my_offset, crash, rebounce, 0, s->stream));
  return MRS_OK;
}

int mrs_swarm_handle_collisions(mrs_swarm_t* s, int32_t enabled, int32_t crash, double rebounce) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (!(crash || enabled)) return MRS_OK;  // src/multirotor_simulator.cpp:299-301
  if (s->comm_world > 1) return fail(MRS_ERR_ARG, "this swarm is one shard of a sharded swarm: use mrs_swarm_tick_sharded_n (the collision pass is collective)");
  if (s->n == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int rc;
  s->collision_ticks++;
  if (s->use_lists) {
    // two collision ticks without a step in between: the earlier one is evaluated now (its crash flags stay, its forces are overwritten)
    if (s->pend.on && (rc = settle(s))) return rc;
    s->fext_active        = true;
    s->pend               = mrs_swarm::Collide{true, enabled, crash, rebounce};
    s->collide_since_step = true;
    if (!s->use_fused) return settle(s);
    return MRS_OK;  // evaluated by the next step launch, or by settle() when the host looks at the swarm first
  }
  if ((rc = settle(s))) return rc;
  if ((rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001))) return rc;
  s->fext_active = true;
  if (!s->dRec) HIPCHK(hipMalloc(&s->dRec, sizeof(PosRecord) * (size_t)s->npad));
  HIPCHK(mrs_collide_run(s->view(), &s->cwork, s->dRec, s->n, 0, crash, rebounce, /*rec_is_local_scratch=*/1, s->stream));
  return MRS_OK;
}

// ------------------------------------------------------------------------------------------------
// RCCL, bound at run time: the process must use ONE HIP runtime, so the library named by the caller is loaded (PyTorch-ROCm
// ships its own librccl.so next to its own libamdhip64; a plain C++ host passes NULL for the system one)
// ------------------------------------------------------------------------------------------------
namespace {

int rccl_load(const char* path) {
  static std::mutex load_mtx;  // swarms of different host threads may initialise their communicators concurrently
  std::lock_guard<std::mutex> lk(load_mtx);
  if (g_rccl.lib) return MRS_OK;
  void* lib = dlopen(path && *path ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return fail(MRS_ERR_HIP, std::string("cannot load RCCL: ") + dlerror());
  g_rccl.GetUniqueId    = (int (*)(NcclId*))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommDestroy    = (int (*)(void*))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char* (*)(int))dlsym(lib, "ncclGetErrorString");
  g_rccl.AllGather      = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(lib, "ncclAllGather");
  g_rccl.CommInitRank   = (int (*)(void**, int, NcclId, int))dlsym(lib, "ncclCommInitRank");
  g_rccl.CommCount      = (int (*)(void*, int*))dlsym(lib, "ncclCommCount");
  if (!g_rccl.GetUniqueId || !g_rccl.CommDestroy || !g_rccl.AllGather || !g_rccl.CommInitRank) {
    g_rccl = RcclApi();
    dlclose(lib);
    return fail(MRS_ERR_HIP, "the RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclAllGather / ncclCommDestroy");
  }
  g_rccl.lib = lib;  // last: the unlocked readers (the communicator calls) only run after a successful load
  return MRS_OK;
}
int rccl_check(int rc, const char* what) {
  if (rc == 0) return MRS_OK;
  return fail(MRS_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}
}  // namespace

int mrs_rccl_unique_id(const char* librccl_path, uint8_t* id128) {
  if (!id128) return fail(MRS_ERR_ARG, "null id");
  int rc = rccl_load(librccl_path);
  if (rc) return rc;
  NcclId id;
  if ((rc = rccl_check(g_rccl.GetUniqueId(&id), "ncclGetUniqueId"))) return rc;
  memcpy(id128, id.internal, 128);
  return MRS_OK;
}

// ---- in-process collective: `world` swarms of one process, one host thread each -------------------------------------------------
struct mrs_loopback_group {
  int                      world = 0;
  std::atomic<int>         arrived{0};
  std::atomic<unsigned>    generation{0};
  std::vector<const void*> send;
  std::vector<hipEvent_t>  ev_ready, ev_copied;
  std::vector<int>         device;
  std::atomic<int>         failed{0};
  // rendezvous mode (mrs_loopback_group_set_rendezvous): no barrier — a rank stages its block, publishes the index of the collective
  // and only waits until every peer has published the same index (it cannot enqueue copies of data a peer has not enqueued yet);
  // staging buffers and events alternate between two sets, so a fast rank never waits for a slow one to have COPIED
  int                                     rendezvous = 0;
  std::unique_ptr<std::atomic<unsigned>[]> seq;           // seq[q] = collectives rank q has published
  std::vector<void*>                      stage[2];
  std::vector<size_t>                     stage_cap[2];
  std::vector<hipEvent_t>                 ev_ready2[2], ev_copied2[2];
  void barrier() {  // sense-reversing spin barrier (at most a handful of threads, all inside the same library call)
    const unsigned gen = generation.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == world) {
      arrived.store(0, std::memory_order_relaxed);
      generation.store(gen + 1, std::memory_order_release);
    } else {
      long spins = 0;
      const auto t0 = std::chrono::steady_clock::now();
      while (generation.load(std::memory_order_acquire) == gen) {
        if (++spins > 2000) std::this_thread::yield();
        // a peer that failed (or whose driver thread died) never arrives: give up instead of hanging the process
        if ((spins & 0xFFFF) == 0 && (failed.load() || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))) {
          failed.store(1);
          return;
        }
      }
    }
  }
};

namespace {
// all-gather among the swarms of a loopback group: every rank copies every rank's send buffer into its own receive buffer, device
// to device on its own stream; events order the copies behind the producers and the producers' next writes behind the copies
int loopback_allgather_rendezvous(mrs_loopback_group* g, int rank, const void* send, void* recv, size_t bytes, hipStream_t st) {
  const size_t   r = (size_t)rank;
  const unsigned k = g->seq[r].load(std::memory_order_relaxed);  // index of this collective (only this thread writes seq[rank])
  const int      par = (int)(k & 1u);
  hipError_t     e = hipSuccess;
  // the staging buffer of this parity was last read by the peers in collective k - 2: their "copied" events were recorded before they
  // published k - 1, which this rank waited for in collective k - 1
  for (int q = 0; q < g->world && e == hipSuccess && k >= 2u; q++) e = hipStreamWaitEvent(st, g->ev_copied2[par][(size_t)q], 0);
  if (e == hipSuccess && bytes > g->stage_cap[par][r]) {
    e = hipStreamSynchronize(st);
    if (g->stage[par][r]) (void)hipFree(g->stage[par][r]);
    g->stage[par][r] = nullptr;
    if (e == hipSuccess) e = hipMalloc(&g->stage[par][r], bytes);
    g->stage_cap[par][r] = e == hipSuccess ? bytes : 0;
  }
  if (e == hipSuccess && bytes) e = hipMemcpyAsync(g->stage[par][r], send, bytes, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipEventRecord(g->ev_ready2[par][r], st);
  if (e != hipSuccess) g->failed.store(1);
  g->seq[r].store(k + 1u, std::memory_order_release);
  const auto t0 = std::chrono::steady_clock::now();
  for (int q = 0; q < g->world; q++) {
    unsigned long spins = 0;
    while (g->seq[(size_t)q].load(std::memory_order_acquire) < k + 1u) {
      if (++spins > 2000) std::this_thread::yield();
      if ((spins & 0xFFFF) == 0 && (g->failed.load() || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))) {
        g->failed.store(1);
        return fail(MRS_ERR_HIP, "loopback all-gather (rendezvous): a peer never arrived at collective " + std::to_string(k));
      }
    }
  }
  for (int q = 0; q < g->world && e == hipSuccess; q++) {
    e = hipStreamWaitEvent(st, g->ev_ready2[par][(size_t)q], 0);
    if (e == hipSuccess && bytes) e = hipMemcpyAsync((char*)recv + (size_t)q * bytes, g->stage[par][(size_t)q], bytes, hipMemcpyDeviceToDevice, st);
  }
  if (e == hipSuccess) e = hipEventRecord(g->ev_copied2[par][r], st);
  if (e != hipSuccess || g->failed.load()) {
    g->failed.store(1);
    return fail(MRS_ERR_HIP, std::string("loopback all-gather (rendezvous): ") + hipGetErrorString(e));
  }
  return MRS_OK;
}

int loopback_allgather(mrs_loopback_group* g, int rank, const void* send, void* recv, size_t bytes, hipStream_t st) {
  if (g->rendezvous) return loopback_allgather_rendezvous(g, rank, send, recv, bytes, st);
  const char* what = "record";
  hipError_t  e = hipGetLastError();  // (an error left behind by an earlier asynchronous launch belongs to that launch, not to this collective)
  if (e != hipSuccess) what = "an earlier launch on this thread";
  if (e == hipSuccess) e = hipEventRecord(g->ev_ready[(size_t)rank], st);
  g->send[(size_t)rank] = send;
  if (e != hipSuccess) g->failed.store(1);
  g->barrier();  // every rank has published its buffer and recorded "my send data is complete"
  for (int q = 0; q < g->world && e == hipSuccess; q++) {
    what = "wait for a peer's data";
    e = hipStreamWaitEvent(st, g->ev_ready[(size_t)q], 0);
    if (e == hipSuccess && bytes) {
      what = "copy";
      e = hipMemcpyAsync((char*)recv + (size_t)q * bytes, g->send[(size_t)q], bytes, hipMemcpyDeviceToDevice, st);
    }
  }
  if (e == hipSuccess) e = hipEventRecord(g->ev_copied[(size_t)rank], st);
  if (e != hipSuccess) g->failed.store(1);
  g->barrier();  // every rank has enqueued its copies
  for (int q = 0; q < g->world && e == hipSuccess; q++) e = hipStreamWaitEvent(st, g->ev_copied[(size_t)q], 0);  // nobody overwrites a buffer a peer still reads
  if (e != hipSuccess) return fail(MRS_ERR_HIP, std::string("loopback all-gather (") + what + ", " + std::to_string(bytes) + " bytes per rank): " + hipGetErrorString(e));
  if (g->failed.load()) return fail(MRS_ERR_HIP, "loopback all-gather: a peer of the group has failed");
  return MRS_OK;
}

extern "C" hipError_t mrs_launch_peer_allgather(const MrsPeerWindows* pw, const void* send, void* recv, size_t bytes, int rank, int world, unsigned seq,
                                                size_t slot_bytes, unsigned* tickets, unsigned ticket_total, unsigned* err_host, unsigned* bpp_out,
                                                hipStream_t st);
extern "C" hipError_t mrs_launch_standin_gather(const void* send, void* recv, size_t bytes, int rank, int world, double latency_us, int records, double width,
                                                hipStream_t st);
// The collective of the measurement stand-in: ONE kernel that takes `standin_delay_us` of stream time (a collective's latency) and
// leaves the rank's own block in its own place and in the places of its two neighbours in the slab order — records (recognised by
// their size) moved one slab width to either side, slot maps and export blocks as they are: the neighbours are periodic images of
// this rank, so the export sets mirror each other as they do between real neighbours (positions of foreign partners are the
// images' only at a search; between searches they read as far away — fine for a time measurement, meaningless as a simulation).
int standin_allgather(mrs_swarm* s, const void* send, void* recv, size_t bytes) {
  const bool records = bytes == sizeof(PosRecord) * (size_t)s->comm_n_max;
  if (bytes % 16 != 0) {  // (the slot maps: 4 * (n_max + 2) bytes) plain copies, no latency worth modelling on a search tick
    HIPCHK(hipMemsetAsync(recv, 0, bytes * (size_t)s->comm_world, s->cstream));
    for (int d = -1; d <= 1; d++)
      if (s->comm_rank + d >= 0 && s->comm_rank + d < s->comm_world)
        HIPCHK(hipMemcpyAsync((char*)recv + (size_t)(s->comm_rank + d) * bytes, send, bytes, hipMemcpyDeviceToDevice, s->cstream));
    return MRS_OK;
  }
  HIPCHK(mrs_launch_standin_gather(send, recv, bytes, s->comm_rank, s->comm_world, s->standin_delay_us, records ? 1 : 0, s->standin_width, s->cstream));
  return MRS_OK;
}

// the peer-window exchange: every rank issues the same collectives in the same order, so the sequence number is the same everywhere
int peer_allgather(mrs_swarm* s, const void* send, void* recv, size_t bytes) {
  if (bytes > s->peer_slot_bytes) return fail(MRS_ERR_ARG, "peer-window exchange: a block of " + std::to_string(bytes) + " bytes does not fit the window's slots (" + std::to_string(s->peer_slot_bytes) + ")");
  unsigned taken = 0;
  HIPCHK(mrs_launch_peer_allgather(&s->peer_windows, send, recv, bytes, s->comm_rank, s->comm_world, ++s->peer_seq, s->peer_slot_bytes, s->peer_ticket,
                                   s->peer_tickets, s->peer_err, &taken, s->cstream));
  s->peer_tickets += taken;
  return MRS_OK;
}

int peer_failed(mrs_swarm* s) {
  volatile unsigned* e = s->peer_err;
  return fail(MRS_ERR_HIP, "peer-window exchange: rank " + std::to_string(s->comm_rank) + " waited in vain for the block of rank " + std::to_string(e[2]) +
                               " in collective " + std::to_string(e[1]) + " (that rank's flag says " + std::to_string(e[3]) + "; this rank has issued " +
                               std::to_string(s->peer_seq) + " collectives, " + std::to_string(s->x_ticks) + " ticks, " + std::to_string(s->x_searches) +
                               " searches) — the results of this call are not valid and the windows are dead: use fresh processes");
}

int comm_allgather(mrs_swarm* s, const void* send, void* recv, size_t bytes) {
  if (s->comm_peer) return peer_allgather(s, send, recv, bytes);
  if (s->comm_standin) return standin_allgather(s, send, recv, bytes);
  if (s->rccl_comm) return rccl_check(g_rccl.AllGather(send, recv, bytes, /*ncclInt8*/ 0, s->rccl_comm, s->cstream), "ncclAllGather");
  if (s->comm_group) return loopback_allgather(s->comm_group, s->comm_rank, send, recv, bytes, s->cstream);
  if (s->comm_fn) {
    const int rc = s->comm_fn(s->comm_user, send, recv, (uint64_t)bytes, (void*)s->cstream);
    return rc == 0 ? MRS_OK : fail(MRS_ERR_HIP, "the caller's all-gather failed with code " + std::to_string(rc));
  }
  return fail(MRS_ERR_ARG, "no communicator");
}

int comm_setup(mrs_swarm* s, int world, int rank, int64_t n_total) {
  if (world < 1 || rank < 0 || rank >= world || n_total < s->n) return fail(MRS_ERR_ARG, "bad communicator shape");
  if (s->comm_world > 0) return fail(MRS_ERR_ARG, "communicator already initialised");
  const int64_t base = n_total / world, rem = n_total % world;
  const int64_t mine = base + (rank < rem ? 1 : 0);  // equal-count shards, sizes differ by at most one
  if (mine != s->n) return fail(MRS_ERR_ARG, "this swarm does not hold the shard of its rank (n_total / world UAVs, the first n_total % world ranks one more)");
  // (the position records of a shard are addressed through a buffer descriptor with 32-bit byte offsets: step_device.inc store_pos_sc1)
  if ((long long)s->n >= (1ll << 27)) return fail(MRS_ERR_ARG, "a shard of a sharded swarm holds at most 2^27 - 1 UAVs (use more ranks)");
  return MRS_OK;
}

int comm_buffers(mrs_swarm* s, int world, int rank, int64_t n_total) {
  s->comm_world   = world;
  s->comm_rank    = rank;
  s->comm_n_total = n_total;
  s->comm_n_max   = (n_total + world - 1) / world;
  if (s->comm_n_max < 1) s->comm_n_max = 1;
  s->x_ok         = false;
  s->x_last_overflow.assign((size_t)world, 0u);
  if (const char* e = getenv("MRS_EXCHANGE")) s->exchange = atoi(e) == 1 ? MRS_EXCHANGE_FULL_GATHER : MRS_EXCHANGE_EXPORT_SETS;
  HIPCHK(hipMalloc(&s->comm_send, sizeof(PosRecord) * (size_t)s->comm_n_max));
  HIPCHK(hipMalloc(&s->comm_recv, sizeof(PosRecord) * (size_t)s->comm_n_max * (size_t)world));
  HIPCHK(hipMalloc(&s->x_map_send, sizeof(uint32_t) * (size_t)(s->comm_n_max + 2)));
  HIPCHK(hipMalloc(&s->x_map_recv, sizeof(uint32_t) * (size_t)(s->comm_n_max + 2) * (size_t)world));
  HIPCHK(hipMemsetAsync(s->comm_send, 0xFF, sizeof(PosRecord) * (size_t)s->comm_n_max, s->stream));  // NaN padding records never collide
  return MRS_OK;
}
}  // namespace

int mrs_swarm_comm_init(mrs_swarm_t* s, const char* librccl_path, int32_t world, int32_t rank, const uint8_t* id128, int64_t n_total) {
  MRS_ENTER(s);
  if (!s || !id128) return fail(MRS_ERR_ARG, "null argument");
  int rc = comm_setup(s, world, rank, n_total);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  if ((rc = rccl_load(librccl_path))) return rc;
  NcclId id;
  memcpy(id.internal, id128, 128);
  if ((rc = rccl_check(g_rccl.CommInitRank(&s->rccl_comm, world, id, rank), "ncclCommInitRank"))) return rc;
  return comm_buffers(s, world, rank, n_total);
}

int mrs_swarm_comm_init_custom(mrs_swarm_t* s, int32_t world, int32_t rank, int64_t n_total, mrs_allgather_fn fn, void* user) {
  MRS_ENTER(s);
  if (!s || !fn) return fail(MRS_ERR_ARG, "null argument");
  int rc = comm_setup(s, world, rank, n_total);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  s->comm_fn   = fn;
  s->comm_user = user;
  return comm_buffers(s, world, rank, n_total);
}

namespace {
void peer_release(mrs_swarm* s) {
  for (void* p : s->peer_opened) (void)hipIpcCloseMemHandle(p);
  s->peer_opened.clear();
  if (s->peer_window) (void)hipFree(s->peer_window);
  if (s->peer_ticket) (void)hipFree(s->peer_ticket);
  if (s->peer_err) (void)hipHostFree(s->peer_err);
  s->peer_window = nullptr;
  s->peer_ticket = s->peer_err = nullptr;
  s->peer_world  = 0;
  s->comm_peer   = false;
}
}  // namespace

int mrs_swarm_peer_window_create(mrs_swarm_t* s, int32_t world, int32_t rank, int64_t n_total, void** window, uint8_t* ipc_handle64) {
  MRS_ENTER(s);
  if (!s || world > MRS_MAX_PEERS) return fail(MRS_ERR_ARG, "bad peer-window arguments (at most 64 ranks)");
  int rc = comm_setup(s, world, rank, n_total);
  if (rc) return rc;
  if (s->peer_window) return fail(MRS_ERR_ARG, "this swarm already has a peer window");
  HIPCHK(hipSetDevice(s->device));
  // the largest block any collective of the sharded tick sends: the full gather of a search (one record per UAV of the largest shard)
  const int64_t n_max = (n_total + world - 1) / world > 0 ? (n_total + world - 1) / world : 1;
  size_t slot = sizeof(PosRecord) * (size_t)n_max;
  if (slot < sizeof(uint32_t) * (size_t)(n_max + 2)) slot = sizeof(uint32_t) * (size_t)(n_max + 2);
  // (an export block is header + capacity records of 32 B, the capacity up to 1.5 x the largest export set + 127: export_search)
  if (slot < sizeof(Pos4) * ((size_t)n_max + (size_t)n_max / 2 + 129)) slot = sizeof(Pos4) * ((size_t)n_max + (size_t)n_max / 2 + 129);
  slot = (slot + 255) / 256 * 256;
  s->peer_slot_bytes   = slot;
  s->peer_window_bytes = 4096 + 2 * (size_t)world * slot;
  // Written by other devices WHILE kernels of this one poll and read it: uncached device memory ("extended-scope fine-grained" — on
  // this GPU family plain fine-grained memory is only guaranteed coherent across devices at kernel boundaries, and the flags are
  // polled inside a kernel; collective libraries allocate their flag and staging memory the same way).  MRS_PEER_WINDOW_MEMORY =
  // finegrained | coarse for runtimes that cannot export an uncached allocation (every access of the exchange kernel is
  // system-scope either way).
  struct Release {  // any failure below gives the window, the ticket words and the pinned error word back
    mrs_swarm* p;
    ~Release() {
      if (p) peer_release(p);
    }
  } release{s};
  const char* kind = getenv("MRS_PEER_WINDOW_MEMORY");
  if (kind && strcmp(kind, "coarse") == 0)
    HIPCHK(hipMalloc(&s->peer_window, s->peer_window_bytes));
  else
    HIPCHK(hipExtMallocWithFlags(&s->peer_window, s->peer_window_bytes, kind && strcmp(kind, "finegrained") == 0 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
  HIPCHK(hipMalloc((void**)&s->peer_ticket, sizeof(unsigned) * (MRS_MAX_PEERS + 1)));  // (+ the give-up mark the exchange kernels read)
  HIPCHK(hipHostMalloc((void**)&s->peer_err, 64, hipHostMallocMapped));
  *s->peer_err = 0u;
  HIPCHK(hipMemsetAsync(s->peer_window, 0, 4096, s->stream));  // flags: no collective has happened
  HIPCHK(hipMemsetAsync(s->peer_ticket, 0, sizeof(unsigned) * (MRS_MAX_PEERS + 1), s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));  // ... before any peer can learn the address
  if (ipc_handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI carries IPC handles as 64 bytes");
    hipIpcMemHandle_t h;
    const hipError_t  e = hipIpcGetMemHandle(&h, s->peer_window);
    if (e != hipSuccess) return fail(MRS_ERR_HIP, std::string("peer window: hipIpcGetMemHandle: ") + hipGetErrorString(e));
    memcpy(ipc_handle64, &h, 64);
  }
  release.p = nullptr;
  if (window) *window = s->peer_window;
  s->peer_world   = world;
  s->peer_rank    = rank;
  s->peer_n_total = n_total;
  return MRS_OK;
}

int mrs_swarm_comm_init_peer(mrs_swarm_t* s, void* const* windows, const uint8_t* ipc_handles) {
  MRS_ENTER(s);
  if (!s || s->peer_world == 0) return fail(MRS_ERR_ARG, "mrs_swarm_peer_window_create has not been called");
  if (!windows && !ipc_handles) return fail(MRS_ERR_ARG, "the peers' windows are needed as pointers or as IPC handles");
  if (s->comm_world > 0) return fail(MRS_ERR_ARG, "communicator already initialised");
  HIPCHK(hipSetDevice(s->device));
  for (int q = 0; q < s->peer_world; q++) {
    void* p = nullptr;
    if (q == s->peer_rank) {
      p = s->peer_window;
    } else if (windows && windows[q]) {
      p = windows[q];
    } else if (ipc_handles) {
      hipIpcMemHandle_t h;
      memcpy(&h, ipc_handles + 64 * (size_t)q, 64);
      const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) return fail(MRS_ERR_HIP, "peer window of rank " + std::to_string(q) + ": hipIpcOpenMemHandle: " + hipGetErrorString(e));
      s->peer_opened.push_back(p);
    } else {
      return fail(MRS_ERR_ARG, "no window given for rank " + std::to_string(q));
    }
    s->peer_windows.win[q] = p;
  }
  s->comm_peer = true;
  s->peer_seq = s->peer_tickets = 0u;
  const int rc = comm_buffers(s, s->peer_world, s->peer_rank, s->peer_n_total);
  if (rc) s->comm_peer = false;
  return rc;
}

int mrs_swarm_comm_init_standin(mrs_swarm_t* s, int32_t world, int32_t rank, int64_t n_total, double collective_latency_us, double slab_width) {
  MRS_ENTER(s);
  if (!s || !(collective_latency_us >= 0) || !(slab_width > 0)) return fail(MRS_ERR_ARG, "bad stand-in arguments");
  int rc = comm_setup(s, world, rank, n_total);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  s->comm_standin     = true;
  s->standin_delay_us = collective_latency_us;
  s->standin_width    = slab_width;
  return comm_buffers(s, world, rank, n_total);
}

int mrs_loopback_group_create(int32_t world, mrs_loopback_group_t** out) {
  if (!out || world < 1 || world > 64) return fail(MRS_ERR_ARG, "bad loopback group size");
  mrs_loopback_group* g = new mrs_loopback_group();
  g->world = world;
  g->send.assign((size_t)world, nullptr);
  g->ev_ready.assign((size_t)world, nullptr);
  g->ev_copied.assign((size_t)world, nullptr);
  g->device.assign((size_t)world, -1);
  g->seq.reset(new std::atomic<unsigned>[(size_t)world]);
  for (int q = 0; q < world; q++) g->seq[(size_t)q].store(0u);
  for (int par = 0; par < 2; par++) {
    g->stage[par].assign((size_t)world, nullptr);
    g->stage_cap[par].assign((size_t)world, 0);
    g->ev_ready2[par].assign((size_t)world, nullptr);
    g->ev_copied2[par].assign((size_t)world, nullptr);
  }
  *out = g;
  return MRS_OK;
}

int mrs_loopback_group_set_rendezvous(mrs_loopback_group_t* g, int32_t on) {
  if (!g) return fail(MRS_ERR_ARG, "null group");
  for (int q = 0; q < g->world; q++)
    if (g->seq[(size_t)q].load() != 0u) return fail(MRS_ERR_ARG, "the mode of a loopback group is chosen before its first collective");
  g->rendezvous = on ? 1 : 0;
  return MRS_OK;
}

int mrs_loopback_group_destroy(mrs_loopback_group_t* g) {
  if (!g) return MRS_OK;
  for (int q = 0; q < g->world; q++) {
    if (g->device[(size_t)q] >= 0) (void)hipSetDevice(g->device[(size_t)q]);
    if (g->ev_ready[(size_t)q]) (void)hipEventDestroy(g->ev_ready[(size_t)q]);
    if (g->ev_copied[(size_t)q]) (void)hipEventDestroy(g->ev_copied[(size_t)q]);
    for (int par = 0; par < 2; par++) {
      if (g->ev_ready2[par][(size_t)q]) (void)hipEventDestroy(g->ev_ready2[par][(size_t)q]);
      if (g->ev_copied2[par][(size_t)q]) (void)hipEventDestroy(g->ev_copied2[par][(size_t)q]);
      if (g->stage[par][(size_t)q]) (void)hipFree(g->stage[par][(size_t)q]);
    }
  }
  delete g;
  return MRS_OK;
}

int mrs_swarm_comm_init_loopback(mrs_swarm_t* s, mrs_loopback_group_t* g, int32_t rank, int64_t n_total) {
  MRS_ENTER(s);
  if (!s || !g) return fail(MRS_ERR_ARG, "null argument");
  int rc = comm_setup(s, g->world, rank, n_total);
  if (rc) return rc;
  if (g->ev_ready[(size_t)rank]) return fail(MRS_ERR_ARG, "this rank of the loopback group is taken");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipEventCreateWithFlags(&g->ev_ready[(size_t)rank], hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&g->ev_copied[(size_t)rank], hipEventDisableTiming));
  for (int par = 0; par < 2; par++) {
    HIPCHK(hipEventCreateWithFlags(&g->ev_ready2[par][(size_t)rank], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&g->ev_copied2[par][(size_t)rank], hipEventDisableTiming));
    HIPCHK(hipEventRecord(g->ev_copied2[par][(size_t)rank], s->stream));  // (an event that was never recorded must not be waited for)
  }
  g->device[(size_t)rank] = s->device;
  s->comm_group           = g;
  return comm_buffers(s, g->world, rank, n_total);
}

int mrs_swarm_get_split_stats(mrs_swarm_t* s, int64_t* split_ticks, int64_t* boundary_blocks) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (split_ticks) *split_ticks = s->x_split_ticks;
  if (boundary_blocks) *boundary_blocks = (int64_t)s->x_nbnd;
  return MRS_OK;
}

extern "C" hipError_t mrs_launch_stream_delay(hipStream_t st, double microseconds);
int mrs_debug_stream_delay(void* stream, double microseconds) {
  if (!(microseconds >= 0) || microseconds > 1e6) return fail(MRS_ERR_ARG, "bad delay");
  HIPCHK(mrs_launch_stream_delay((hipStream_t)stream, microseconds));
  return MRS_OK;
}

int mrs_swarm_debug_chaos(mrs_swarm_t* s, int32_t max_sleep_us, uint64_t seed) {
  MRS_LOCK(s);
  if (!s || max_sleep_us < 0) return fail(MRS_ERR_ARG, "bad chaos arguments");
  s->chaos_max_us = max_sleep_us;
  s->chaos_state  = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  if (s->chaos_state == 0) s->chaos_state = 1;
  return MRS_OK;
}

int mrs_swarm_set_exchange(mrs_swarm_t* s, int32_t exchange) {
  MRS_ENTER(s);
  if (!s || (exchange != MRS_EXCHANGE_FULL_GATHER && exchange != MRS_EXCHANGE_EXPORT_SETS)) return fail(MRS_ERR_ARG, "bad exchange");
  if (exchange != s->exchange) {
    s->x_ok = false;
    mrs_collide_invalidate_gathered(s->cwork);  // the two exchanges keep their neighbour lists in different forms
  }
  s->exchange = exchange;
  return MRS_OK;
}

int mrs_swarm_comm_destroy(mrs_swarm_t* s) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (s->comm_world == 0) {
    if (s->peer_window) peer_release(s);  // a window that never became a communicator
    return MRS_OK;
  }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  int rc = MRS_OK;
  if (s->rccl_comm) rc = rccl_check(g_rccl.CommDestroy(s->rccl_comm), "ncclCommDestroy");
  s->rccl_comm  = nullptr;
  s->comm_fn    = nullptr;
  s->comm_user  = nullptr;
  s->comm_group = nullptr;
  s->comm_standin = false;
  peer_release(s);
  s->comm_world = 0;
  s->x_ok       = false;
  mrs_collide_invalidate_gathered(s->cwork);
  if (s->comm_send) (void)hipFree(s->comm_send);
  if (s->comm_recv) (void)hipFree(s->comm_recv);
  if (s->x_map_send) (void)hipFree(s->x_map_send);
  if (s->x_map_recv) (void)hipFree(s->x_map_recv);
  s->comm_send = s->comm_recv = nullptr;
  s->x_map_send = s->x_map_recv = nullptr;
  return rc;
}

int mrs_swarm_comm_info(mrs_swarm_t* s, mrs_comm_info_t* out) {
  MRS_ENTER(s);
  if (!s || !out) return fail(MRS_ERR_ARG, "null argument");
  memset(out, 0, sizeof *out);
  if (s->comm_world == 0) return MRS_OK;
  out->world    = s->comm_world;
  out->rank     = s->comm_rank;
  out->n_total  = s->comm_n_total;
  out->exchange = s->exchange;
  const int64_t full = (int64_t)sizeof(PosRecord) * s->comm_n_max;
  if (s->exchange == MRS_EXCHANGE_EXPORT_SETS) {
    out->export_capacity   = mrs_collide_export_capacity(s->cwork);
    out->export_count      = s->x_export_count;
    out->bytes_per_tick    = (int64_t)sizeof(Pos4) * (1 + out->export_capacity);
    out->bytes_per_rebuild = full + (int64_t)sizeof(uint32_t) * (s->comm_n_max + 2);
  } else {
    out->bytes_per_tick = out->bytes_per_rebuild = full;
  }
  out->ticks      = s->x_ticks;
  out->searches   = s->x_searches;
  out->noop_ticks = s->x_noop_ticks;
  if (s->rccl_comm && g_rccl.CommCount) {
    int c = 0;
    int rc = rccl_check(g_rccl.CommCount(s->rccl_comm, &c), "ncclCommCount");
    if (rc) return rc;
    out->rccl_ranks = c;
  }
  return MRS_OK;
}

// UAVs sorted by x (ties: by public index), cut into equal-count slabs by the caller
int mrs_slab_partition(const double* pos_xyz, int64_t n_total, int32_t world, int64_t* order) {
  if (!pos_xyz || !order || n_total < 0 || world < 1) return fail(MRS_ERR_ARG, "bad partition arguments");
  for (int64_t k = 0; k < n_total; k++) order[k] = k;
  std::stable_sort(order, order + n_total, [&](int64_t a, int64_t b) {
    const double xa = pos_xyz[3 * a], xb = pos_xyz[3 * b];
    if (xa != xa || xb != xb) return (xa == xa) && (xb != xb);  // NaN positions last
    return xa < xb;
  });
  return MRS_OK;
}

// ---- sharded ticks ----------------------------------------------------------------------------------------------------------------
namespace {

// every tick gathers all records (MRS_EXCHANGE_FULL_GATHER, and the fallback of the export-set exchange when some UAV has more
// neighbours than its list holds): step, pack, ONE all-gather of the 48-B records, collision pass against the gathered records
int full_gather_ticks(mrs_swarm* s, double dt, int n_ticks, const mrs_swarm::Collide& c) {
  const int64_t n_rec = s->comm_n_max * s->comm_world;
  int           rc;
  s->x_ok    = false;
  s->p_valid = false;
  s->fk_ok   = false;
  for (int k = 0; k < n_ticks; k++) {
    if (s->n > 0 && (rc = launch_step(s, dt, 1))) return rc;
    if (s->n > 0) HIPCHK(mrs_launch_pack_positions(s->view(), s->comm_send, s->stream));
    if ((rc = comm_allgather(s, s->comm_send, s->comm_recv, sizeof(PosRecord) * (size_t)s->comm_n_max))) return rc;
    s->x_ticks++;
    if (s->n == 0) continue;
    s->fext_active = true;
    s->collision_ticks++;
    s->nbr_dirty = true;
    if (s->use_lists)
      HIPCHK(mrs_collide_run_lists_gathered(s->view(), &s->cwork, s->comm_recv, n_rec, (int64_t)s->comm_rank * s->comm_n_max, c.crash, c.rebounce, 0, s->stream));
    else
      HIPCHK(mrs_collide_run(s->view(), &s->cwork, s->comm_recv, n_rec, (int64_t)s->comm_rank * s->comm_n_max, c.crash, c.rebounce, 0, s->stream));
  }
  return MRS_OK;
}

// The tick after the most recent step on the SEARCH path of the export-set exchange: gather all records, search (which evaluates
// this tick's handleCollisions), then derive the export sets and rewrite the lists.  Collective.  Returns 1 when the lists came out
// incomplete (a UAV with more neighbours than its list holds, on any rank): the caller stays on the full exchange for a while.
int export_search(mrs_swarm* s, const mrs_swarm::Collide& c, int* incomplete) {
  const int     world = s->comm_world, rank = s->comm_rank;
  const int64_t n_max = s->comm_n_max, n_rec = n_max * world, stride = n_max + 2;
  int           rc;
  *incomplete = 0;
  s->x_ok     = false;
  s->x_searches++;
  if (s->n > 0) HIPCHK(mrs_launch_pack_positions(s->view(), s->comm_send, s->stream));
  if ((rc = comm_allgather(s, s->comm_send, s->comm_recv, sizeof(PosRecord) * (size_t)n_max))) return rc;
  s->fext_active = true;
  s->nbr_dirty   = true;  // (a later single-GPU tick starts from a search of its own)
  if (s->n > 0)
    HIPCHK(mrs_collide_run_lists_gathered(s->view(), &s->cwork, s->comm_recv, n_rec, (int64_t)rank * n_max, c.crash, c.rebounce, /*force=*/1, s->stream));
  long long cap = mrs_collide_export_capacity(s->cwork);
  HIPCHK(mrs_collide_export_prepare(s->view(), &s->cwork, world, cap > 0 ? cap : 64, s->stream));
  HIPCHK(mrs_collide_export_mark(s->view(), s->cwork, n_max, rank, s->x_map_send, s->stream));
  if ((rc = comm_allgather(s, s->x_map_send, s->x_map_recv, sizeof(uint32_t) * (size_t)stride))) return rc;
  // the heads of all ranks' maps: export count, lanes over the list capacity so far — the same numbers on every rank
  const uint32_t* heads = nullptr;  // (pinned host words, written by one small launch)
  HIPCHK(mrs_collide_heads_to_host(s->cwork, s->x_map_recv, stride, world, &heads, s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  s->x_nbnd = s->n > 0 ? heads[2 * world] : 0u;
  long long need = 0;
  for (int q = 0; q < world; q++) {
    if ((long long)heads[(size_t)q * 2] > need) need = heads[(size_t)q * 2];
    if (heads[(size_t)q * 2 + 1] != s->x_last_overflow[(size_t)q]) *incomplete = 1;
    s->x_last_overflow[(size_t)q] = heads[(size_t)q * 2 + 1];
  }
  s->x_export_count = heads[(size_t)rank * 2];
  if (*incomplete) {
    mrs_collide_invalidate_gathered(s->cwork);
    return MRS_OK;
  }
  if (need > cap || cap == 0) {  // grow with headroom: the sets change from search to search
    long long ncap = ((need + need / 2 + 64 + 63) / 64) * 64;
    HIPCHK(mrs_collide_export_prepare(s->view(), &s->cwork, world, ncap, s->stream));
  }
  HIPCHK(mrs_collide_export_translate(s->view(), s->cwork, n_max, rank, s->x_map_recv, s->comm_recv, s->stream));
  // (the lists are in export form now; collide.hip remembers that, and the full exchange would start with a search of its own)
  s->x_ok = true;
  s->tau  = 0;
  return MRS_OK;
}

int launch_fused_export(mrs_swarm* s, double dt, const mrs_swarm::Collide& eval) {
  if (s->n > 0) {
    CollDev  cd;
    SwarmDev v = s->view();
    HIPCHK(mrs_collide_export_dev(&v, s->cwork, (int64_t)s->comm_rank * s->comm_n_max, s->tau + 1, eval.on ? 1 : 0, eval.crash, eval.rebounce, &cd));
    mrs_collide_export_part(&cd, MRS_PART_FULL, 0u, dt, s->shard_split ? 1 : 0);  // (MRS_SHARD_SPLIT=0: round 2's protocol, nothing announced)
    s->region_launches++;
    const int variant = s->n_cascade > 0 ? 0 : 1;
    if (s->arith == MRS_ARITH_FAST)
      HIPCHK(mrs_launch_step_coll_fast(v, cd, dt, variant, 0, s->stream));
    else
      HIPCHK(mrs_launch_step_coll_literal(v, cd, dt, variant, 0, s->stream));
    mrs_collide_fused_advance(s->cwork);
  } else {
    HIPCHK(mrs_collide_export_fold_stall(s->cwork, s->tau + 1, s->stream));  // a rank without UAVs still watches the headers
  }
  s->tau++;
  const size_t bytes = sizeof(Pos4) * (size_t)(mrs_collide_export_capacity(s->cwork) + 1);
  return comm_allgather(s, mrs_collide_export_send(s->cwork), mrs_collide_export_recv(s->cwork), bytes);
}

// One tick of the export-set exchange in the SPLIT form: the boundary launch (the blocks that hold a UAV with a foreign partner; it
// evaluates, steps, writes this rank's export block) and the collective on `stream`, the interior launch on `stream2`.  The two
// chains meet inside the kernels (per-block epoch words: step_device.inc), never on the host and never through an event, so the
// interior launch of tick t+1 runs beside the collective of tick t.  `split_base`: launch index behind which this run of split
// ticks started (mrs_collide_handoff_init).
int launch_split_export(mrs_swarm* s, double dt, const mrs_swarm::Collide& eval, unsigned split_base) {
  CollDev  cd, part;
  SwarmDev v = s->view();
  HIPCHK(mrs_collide_export_dev(&v, s->cwork, (int64_t)s->comm_rank * s->comm_n_max, s->tau + 1, eval.on ? 1 : 0, eval.crash, eval.rebounce, &cd));
  s->region_launches++;
  const int      variant = s->n_cascade > 0 ? 0 : 1, bound_ok = 1;
  const unsigned grid_b  = s->x_nbnd > 0 ? s->x_nbnd : 1u;
  auto launch = [&](const CollDev& c, int grid, hipStream_t st) {
    return s->arith == MRS_ARITH_FAST ? mrs_launch_step_coll_fast(v, c, dt, variant, grid, st) : mrs_launch_step_coll_literal(v, c, dt, variant, grid, st);
  };
  part = cd;
  mrs_collide_export_part(&part, MRS_PART_BOUNDARY, s->x_nbnd, dt, bound_ok);
  HIPCHK(launch(part, (int)grid_b, s->cstream));
  const size_t bytes = sizeof(Pos4) * (size_t)(mrs_collide_export_capacity(s->cwork) + 1);
  int rc = comm_allgather(s, mrs_collide_export_send(s->cwork), mrs_collide_export_recv(s->cwork), bytes);
  if (rc) return rc;
  part = cd;
  mrs_collide_export_part(&part, MRS_PART_INTERIOR, s->x_nbnd, dt, bound_ok);
  HIPCHK(launch(part, 0, s->stream_i ? s->stream_i : s->stream2));
  mrs_collide_fused_advance(s->cwork);
  s->tau++;
  s->x_split_ticks++;
  return MRS_OK;
}

// Ticks of the export-set exchange.  All ranks must issue the same launches and collectives in the same order, yet nobody may wait
// for anybody on the host.  What keeps them in step: every decision is taken from words that reach all ranks with the positions
// themselves (headers of the export collective, folded by each fused launch into pinned host words), at a launch index that is a
// function of those words alone:
//   * warning word W (tick in which some UAV of some rank had used 75 % of its skin): the search is done before launch W + D;
//   * stall word T (some UAV left its skin during step T; launches > T are no-ops everywhere): the segment ends with launch T + L + 1;
//   * L = launches a host may run ahead of its device (progress word), D = L + 3: a host deciding on launch W + D, or on T + L + 1,
//     has provably seen W, or T (the launches that report them have completed on its device by then).
// A segment ends with fold + synchronise (the only host wait), then the search where one is due.
int export_ticks(mrs_swarm* s, double dt, int n_ticks, const mrs_swarm::Collide& c) {
  int  rc, done = 0;
  bool pending = false;  // the collision tick after the most recent step has not been evaluated yet
  s->p_valid = false;    // (single-GPU lazies do not mix with this path)
  s->fk_ok   = false;
  // Split ticks (see launch_split_export; the protocol and its proof obligations: DESIGN §5).  In the split form a launch does not
  // see the reports the collective of the previous tick carries, so a stall index T must be ANNOUNCED MRS_PRED_HORIZON launches
  // ahead (displacement bound, step_device.inc) — and the ticks the bound cannot vouch for run in the serial form, whose launches
  // test exactly and hear of each other's reports through the collective in stream order: the first MRS_PRED_HORIZON ticks of
  // every call (the host may have written positions, velocities or airframe constants since the last one) and after every search.
  const bool protocol_split = s->shard_split;  // (the same on every rank: it sets how long a report takes to reach everybody)
  const unsigned lead = (unsigned)(s->fused_lead > 0 ? s->fused_lead : 1), search_ahead = lead + (protocol_split ? 6u : 3u);
  int serial_left = (int)MRS_PRED_HORIZON;
  s->cstream = s->stream;  // (a call that failed inside a split segment may have left it elsewhere)
  if (dt != s->x_dt) s->x_ok = false;  // the announcements of the last call's final launches assumed its dt: start from a search
  s->x_dt = dt;
  const int nb = (s->n + 63) / 64;
  const volatile unsigned* hw = nullptr;  // pinned host mirror of the control words (exists once a search has run)
  auto split_ok = [&]() {
    // (a communicator of one rank has no boundary and announces nothing: its exact reports need the serial form)
    if (!(protocol_split && s->comm_world > 1 && s->n > 0 && nb >= s->split_min_blocks && (double)s->x_nbnd <= s->split_max_fraction * nb && s->mixed_blocks.empty() && s->stream2 != nullptr))
      return false;
    // Residency: the waves that SPIN inside a split tick — block 0 and the layer-1 blocks of an interior launch waiting for
    // the boundary launch of the previous tick, the boundary blocks waiting for an interior launch — hold their wave slots while they
    // wait.  "Producers are enqueued before consumers" covers the hardware queues, not SIMD and register slots: if spinning interior
    // waves could fill the device, a boundary launch queued behind a late collective would find no slot and the tick would end in the
    // 10-s give-up.  So a rank stays in the serial form unless the spinners leave at least half of the wave slots (at the interior
    // kernel's two waves per SIMD) to everybody else — unless the boundary chain owns compute units of its own (MRS_SPLIT_CU_RESERVE).
    // The count comes from the search (CTL_NL1, mirrored to the host words by its last launch); unknown yet: serial.
    const unsigned nl1 = hw ? hw[CTL_NL1] : 0xFFFFFFFFu;
    if (nl1 == 0xFFFFFFFFu) return false;
    if (s->cu_reserve > 0) return true;
    return (long long)nl1 + 1 <= s->resident_waves / 2 && (long long)s->x_nbnd <= s->resident_waves / 4;
  };
  while (done < n_ticks) {
    if (s->x_fallback_left > 0) {
      const int k = n_ticks - done < s->x_fallback_left ? n_ticks - done : s->x_fallback_left;
      if ((rc = full_gather_ticks(s, dt, k, c))) return rc;
      s->x_fallback_left -= k;
      done += k;
      continue;
    }
    if (!s->x_ok) {  // no usable export lists (first tick, lists gone stale): this tick on the search path
      if (s->n > 0 && (rc = launch_step(s, dt, 1))) return rc;
      int incomplete = 0;
      if ((rc = export_search(s, c, &incomplete))) return rc;
      s->collision_ticks++;
      s->x_ticks++;
      done++;
      pending = false;
      serial_left = (int)MRS_PRED_HORIZON;
      if (incomplete) s->x_fallback_left = 64;
      continue;
    }
    // ---- a segment of fused ticks ----
    hw = mrs_collide_host_words(s->cwork);
    if (!hw) return fail(MRS_ERR_HIP, "export-set exchange: the control words of the fused launches do not exist");
    const unsigned first = s->tau + 1;                                 // launch indices run on from the last search
    unsigned       last  = s->tau + (unsigned)(n_ticks - done);       // ... to the end of the call, unless a word says otherwise
    mrs_swarm::Collide off;
    bool     in_split   = false;
    unsigned split_base = 0;
    s->chaos_T = s->chaos_W = 0u;
    while (s->tau < last) {
      const unsigned next = s->tau + 1;
      // (a device that makes no progress is an ERROR here, never a reason to launch anyway: lock-step of the ranks rests on every host
      //  having seen the words of launch next - lead - 1 before it issues launch `next`)
      if ((rc = wait_for_progress(s, hw, next, (int)lead))) return rc;
      unsigned T = stall_word(hw), W = warn_word(hw);
      if (s->chaos_max_us > 0) {
        s->chaos_state ^= s->chaos_state << 13; s->chaos_state ^= s->chaos_state >> 7; s->chaos_state ^= s->chaos_state << 17;
        const unsigned Tf = T, Wf = W;
        if (s->chaos_state & 0x100u) { T = s->chaos_T; W = s->chaos_W; }
        s->chaos_T = Tf; s->chaos_W = Wf;
        usleep((useconds_t)((s->chaos_state >> 16) % (uint64_t)(s->chaos_max_us + 1)));
      }
      if (T != 0u && T + lead + 1 < last) last = T + lead + 1;
      if (W != 0u && W + search_ahead - 1 < last) last = W + search_ahead - 1;
      if (next > last) break;
      if (serial_left == 0 && !in_split && split_ok() && last - s->tau >= 4u) {
        // from the serial form to the split one: what the last collective carried is folded into the control words (the first
        // interior launch reads nothing else), every block counts as finished by launch tau, and the second stream starts behind all that
        HIPCHK(mrs_collide_export_fold_stall(s->cwork, 0u, s->stream));
        HIPCHK(mrs_collide_handoff_init(s->cwork, s->n, s->tau, s->stream));
        HIPCHK(hipEventRecord(s->ev_fork, s->stream));
        HIPCHK(hipStreamWaitEvent(s->stream_i ? s->stream_i : s->stream2, s->ev_fork, 0));
        if (s->stream_b) {
          HIPCHK(hipStreamWaitEvent(s->stream_b, s->ev_fork, 0));
          s->cstream = s->stream_b;
        }
        in_split   = true;
        split_base = s->tau;
      }
      if (in_split) {
        if ((rc = launch_split_export(s, dt, pending ? c : off, split_base))) return rc;
      } else {
        if ((rc = launch_fused_export(s, dt, pending ? c : off))) return rc;
        if (serial_left > 0) serial_left--;
      }
      pending = true;
    }
    if (in_split) {  // back to one stream: everything that follows (fold, search, the caller's work) comes behind the interior launches too
      HIPCHK(hipEventRecord(s->ev_join, s->stream_i ? s->stream_i : s->stream2));
      HIPCHK(hipStreamWaitEvent(s->stream, s->ev_join, 0));
      if (s->stream_b) {
        HIPCHK(hipEventRecord(s->ev_join_b, s->stream_b));
        HIPCHK(hipStreamWaitEvent(s->stream, s->ev_join_b, 0));
        s->cstream = s->stream;
      }
    }
    if (protocol_split) {
      // What a rank's interior launches reported in the last ticks of the segment sits in the header of its export block but has
      // not travelled yet: one more exchange of the export blocks, so that every rank ends the segment with the same words.
      // (All ranks do this, whichever form their own ticks took.)
      const size_t bytes = sizeof(Pos4) * (size_t)(mrs_collide_export_capacity(s->cwork) + 1);
      if ((rc = comm_allgather(s, mrs_collide_export_send(s->cwork), mrs_collide_export_recv(s->cwork), bytes))) return rc;
    }
    HIPCHK(mrs_collide_export_fold_stall(s->cwork, 0u, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    const unsigned T = stall_word(hw), W = warn_word(hw);  // identical on every rank
    const unsigned launched = s->tau + 1 - first;
    const unsigned ran = (T != 0u && T + 1 >= first && T + 1 - first < launched) ? T + 1 - first : launched;
    done += (int)ran;
    s->x_ticks += ran;
    s->collision_ticks += ran;
    s->x_noop_ticks += launched - ran;
    if (T != 0u || (W != 0u && done < n_ticks)) {
      // the lists are stale after step T / about to be: all ranks search on the state they have now, which also evaluates the
      // collision tick that followed the last step that ran
      int incomplete = 0;
      if ((rc = export_search(s, c, &incomplete))) return rc;
      pending = false;
      serial_left = (int)MRS_PRED_HORIZON;
      if (incomplete) s->x_fallback_left = 64;
    }
  }
  if (pending && s->n > 0) {  // the last tick's handleCollisions: the export buffer holds the positions after the last step
    CollDev  cd;
    SwarmDev v = s->view();
    HIPCHK(mrs_collide_export_dev(&v, s->cwork, (int64_t)s->comm_rank * s->comm_n_max, 1u, 1, c.crash, c.rebounce, &cd));
    cd.p_out = nullptr;
    HIPCHK(mrs_collide_export_eval(v, cd, s->stream));
  }
  unsigned w[CTL_WORDS];
  HIPCHK(mrs_collide_fused_words(s->cwork, s->stream, w));
  if (w[CTL_BADSLOT]) return fail(MRS_ERR_HIP, "export-set exchange: a listed foreign UAV is not in its owner's export set (" + std::to_string(w[CTL_BADSLOT]) + " entries)");
  if (s->peer_err && *s->peer_err) return peer_failed(s);
  if (w[CTL_ERROR] & 1u) return fail(MRS_ERR_HIP, "split sharded tick: a launch waited in vain for the launch on the other stream (the results of this call are not valid; the stream and the communicator are dead: use a fresh process)");
  if (w[CTL_ERROR] & 2u) return fail(MRS_ERR_HIP, "split sharded tick: a UAV left its skin without the displacement bound announcing it (DESIGN §5) — the results of this call are not valid; run with MRS_SHARD_SPLIT=0 on every rank and report the case");
  if (w[CTL_ERROR] & 0x300u)  // (any rank's error invalidates every rank's results: the ranks that only HEARD of it would otherwise return MRS_OK with a wrong state)
    return fail(MRS_ERR_HIP, std::string("sharded tick: another rank of the swarm reported ") + ((w[CTL_ERROR] & 0x200u) ? "an unannounced skin exit (displacement bound violated)" : "a wait that ran out") +
                                 " — the results of this call are not valid on ANY rank");
  return MRS_OK;
}
}  // namespace

// timerMain on every rank of a sharded swarm (no host synchronisation inside a batch of ticks, everything on the swarm's stream)
int mrs_swarm_tick_sharded_n(mrs_swarm_t* s, double dt, int32_t n_ticks, int32_t enabled, int32_t crash, double rebounce) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (s->comm_world == 0) return fail(MRS_ERR_ARG, "mrs_swarm_comm_init has not been called");
  if (!(dt > 0) || n_ticks < 0) return fail(MRS_ERR_ARG, "bad tick arguments");
  if (n_ticks == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  s->cstream = s->stream;  // (a call that failed inside a split segment may have left it on the boundary chain's stream)
  int rc = upload_types(s, dt);
  if (rc) return rc;
  // Host writes since the last sharded tick (set_state, set_mass, ... possibly on this rank only) do not touch x_ok: all ranks must
  // take the same path.  The first fused launch notices them on the device — a UAV away from its recorded position or with other
  // airframe constants than its record raises the stall word, which travels to every rank in the collective's headers.
  if ((rc = begin_profile(s))) return rc;
  if (!(crash || enabled)) {  // src/multirotor_simulator.cpp:299-301: no collision pass, no exchange
    for (int k = 0; k < n_ticks; k++)
      if (s->n > 0 && (rc = launch_step(s, dt, 1))) return rc;
    return finish_profile(s);
  }
  const mrs_swarm::Collide c{true, enabled, crash, rebounce};
  if (s->exchange == MRS_EXCHANGE_EXPORT_SETS && s->use_lists && s->use_fused)
    rc = export_ticks(s, dt, n_ticks, c);
  else
    rc = full_gather_ticks(s, dt, n_ticks, c);
  // (a peer that never answered is the CAUSE of whatever else went wrong behind it — a search over blocks that never arrived, say)
  if (s->peer_err && *s->peer_err) return peer_failed(s);
  if (rc) return rc;
  s->nbr_dirty = false;
  return finish_profile(s);
}

int mrs_swarm_tick_n(mrs_swarm_t* s, double dt, int32_t n_ticks, int32_t enabled, int32_t crash, double rebounce) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (!(dt > 0) || n_ticks < 0) return fail(MRS_ERR_ARG, "bad tick arguments");
  if (s->n == 0 || n_ticks == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, dt);
  if (rc) return rc;
  if ((rc = begin_profile(s))) return rc;
  for (int k = 0; k < n_ticks; k++) {
    if ((rc = step_one(s, dt))) return rc;
    if ((rc = mrs_swarm_handle_collisions(s, enabled, crash, rebounce))) return rc;
  }
  return finish_profile(s);
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

int mrs_swarm_clone(mrs_swarm_t* s, mrs_swarm_t** out) {
  MRS_ENTER(s);
  if (!s || !out) return fail(MRS_ERR_ARG, "null argument");
  *out = nullptr;
  mrs_swarm* c = nullptr;
  int rc = mrs_swarm_create(s->n, s->device, &c);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(c->stream));
  hipError_t e = hipMemcpyAsync(c->dS, s->dS, sizeof(double) * (size_t)F_COUNT * s->npad, hipMemcpyDeviceToDevice, s->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(c->dF, s->dF, sizeof(uint32_t) * (size_t)s->npad, hipMemcpyDeviceToDevice, s->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(c->dDiag, s->dDiag, sizeof(unsigned long long) * 4, hipMemcpyDeviceToDevice, s->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
  if (e != hipSuccess) {
    mrs_swarm_destroy(c);
    return fail(MRS_ERR_HIP, std::string("clone: ") + hipGetErrorString(e));
  }
  c->arith        = s->arith;
  c->keys         = s->keys;
  c->tparams      = s->tparams;
  c->key_index    = s->key_index;
  c->uav_type     = s->uav_type;
  c->uav_mode     = s->uav_mode;
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
  MRS_ENTER(s);
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

int mrs_swarm_input_staging(mrs_swarm_t* s, int32_t count, int32_t stride, double** rows) {
  MRS_ENTER(s);
  if (!s || !rows) return fail(MRS_ERR_ARG, "null argument");
  if (count < 0 || count > s->n || stride < 1 || stride > 16) return fail(MRS_ERR_ARG, "bad staging shape");
  HIPCHK(hipSetDevice(s->device));
  const int64_t need = (int64_t)count * stride;
  HIPCHK(hipStreamSynchronize(s->stream));  // an earlier commit may still be reading the rows
  if (need > s->in_cap) {
    if (s->hIn) HIPCHK(hipHostFree(s->hIn));
    if (s->dIn) HIPCHK(hipFree(s->dIn));
    s->hIn = nullptr;
    s->dIn = nullptr;
    HIPCHK(hipHostMalloc(&s->hIn, sizeof(double) * (size_t)need, hipHostMallocDefault));
    HIPCHK(hipMalloc(&s->dIn, sizeof(double) * (size_t)need));
    s->in_cap = need;
  }
  *rows = s->hIn;
  return MRS_OK;
}

int mrs_swarm_commit_input(mrs_swarm_t* s, int32_t first, int32_t count, int32_t mode, int32_t stride) {
  MRS_ENTER(s);
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (mode < MRS_ACTUATOR_CMD || mode > MRS_POSITION_CMD) return fail(MRS_ERR_ARG, "bad input mode");
  if (count == 0) return MRS_OK;
  if (!s->hIn || (int64_t)count * stride > s->in_cap) return fail(MRS_ERR_ARG, "no staging rows of this shape (call mrs_swarm_input_staging first)");
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
  HIPCHK(hipMemcpyAsync(s->dIn, s->hIn, sizeof(double) * (size_t)count * (size_t)stride, hipMemcpyHostToDevice, s->stream));
  HIPCHK(mrs_launch_unpack_rows(s->view(), s->dIn, stride, width, F_CMD, first, count, s->stream));
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

// measurement hook (bench.py roofline_collision): `reps` neighbour searches of the single-GPU collision pass back to back on the
// swarm's stream — pack + insert, then the list-building query, exactly what a tick that repeats the search launches — between
// two hipEvents.  The forces / crash flags latched are those of handleCollisions(enabled, crash, rebounce) on the current positions.
int mrs_swarm_debug_search_ms(mrs_swarm_t* s, int32_t reps, int32_t crash, double rebounce, double* avg_ms) {
  MRS_ENTER(s);
  if (!s || reps < 1 || !avg_ms) return fail(MRS_ERR_ARG, "bad search-timing arguments");
  if (s->comm_world > 1 || !s->use_lists) return fail(MRS_ERR_ARG, "search timing: single-GPU swarms with neighbour lists only");
  if (s->n == 0) { *avg_ms = 0.0; return MRS_OK; }
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001);
  if (rc) return rc;
  const mrs_swarm::Collide c{true, 1, crash, rebounce};
  if ((rc = collide_now(s, c, /*force=*/true))) return rc;  // buffers exist, tables are in their steady state
  while (s->ev.size() < 2) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    s->ev.push_back(e);
  }
  HIPCHK(hipEventRecord(s->ev[0], s->stream));
  for (int k = 0; k < reps; k++) HIPCHK(mrs_collide_run_lists(s->view(), &s->cwork, crash, rebounce, 1, 0u, s->stream));
  HIPCHK(hipEventRecord(s->ev[1], s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
  *avg_ms = (double)ms / reps;
  s->fext_active = true;
  return collide_now(s, c, /*force=*/true);  // host bookkeeping (list completeness, lazies) as after any stand-alone pass
}

int mrs_swarm_set_profiling(mrs_swarm_t* s, int32_t enabled) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  s->profiling = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
  return MRS_OK;
}

int mrs_swarm_last_step_kernel_ms(mrs_swarm_t* s, double* avg_ms, int32_t* n_launches) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (avg_ms) *avg_ms = s->last_ms;
  if (n_launches) *n_launches = s->last_launches;
  return MRS_OK;
}

}  // extern "C"
