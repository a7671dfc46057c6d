/*
 * mrs_swarm.h — C ABI of the MI355X-native multi-UAV stepper (libmrs_swarm.so).
 *
 * Drop-in boundary for the reference's UavSystem::makeStep() hot path: a whole swarm of
 * mrs_multirotor_simulator::UavSystem objects lives on one GPU as SoA FP64 state; every entry point
 * below replaces the per-UAV C++ call named next to it (paths relative to /root/reference).
 * Plain C types, caller-owned host buffers, int return codes (MRS_OK == 0) — the reference itself
 * signals no errors on this path.  The header-only C++ facade
 * include/mrs_multirotor_simulator/uav_system/uav_system.hpp re-exports the reference's class
 * names on top of this ABI.
 *
 * There is NO CPU fallback: every compute entry point fails with MRS_ERR_HIP when no gfx950 device
 * is usable.
 */
#ifndef MRS_SWARM_H
#define MRS_SWARM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MRS_MAX_MOTORS 8

enum { MRS_OK = 0, MRS_ERR_ARG = 1, MRS_ERR_HIP = 2, MRS_ERR_RANGE = 3, MRS_ERR_TYPES = 4 };

/* UavSystem::INPUT_MODE — include/mrs_multirotor_simulator/uav_system/uav_system.hpp:19-32 */
enum {
  MRS_INPUT_UNKNOWN = 0,
  MRS_ACTUATOR_CMD,
  MRS_CONTROL_GROUP_CMD,
  MRS_ATTITUDE_RATE_CMD,
  MRS_ATTITUDE_CMD,
  MRS_TILT_HDG_RATE_CMD,
  MRS_ACCELERATION_HDG_RATE_CMD,
  MRS_ACCELERATION_HDG_CMD,
  MRS_VELOCITY_HDG_RATE_CMD,
  MRS_VELOCITY_HDG_CMD,
  MRS_POSITION_CMD
};

/* the four std::optional feed-forward slots — uav_system.hpp:112-115 */
enum { MRS_FF_VELOCITY_HDG_RATE = 0, MRS_FF_VELOCITY_HDG, MRS_FF_ACCELERATION_HDG_RATE, MRS_FF_ACCELERATION_HDG };

/* multi-GPU collision exchange (mrs_swarm_set_exchange) */
enum { MRS_EXCHANGE_NONE = 0, MRS_EXCHANGE_FULL_GATHER = 1, MRS_EXCHANGE_EXPORT_SETS = 2 };

/* arithmetic flavour of the step kernel */
enum {
  MRS_ARITH_LITERAL = 0, /* reference operation order, no FMA contraction: bit-comparable with a scalar CPU restatement */
  MRS_ARITH_FAST    = 1  /* FMA contraction + reciprocal/triangular simplifications; same results to ~1e-12 relative */
};

/* MultirotorModel::ModelParams — uav_system/multirotor_model.hpp:24-88 (matrices row-major) */
typedef struct {
  int32_t n_motors;
  int32_t ground_enabled;
  int32_t takeoff_patch_enabled;
  int32_t _pad;
  double  g, mass, kf, km, prop_radius, arm_length, body_height, motor_time_constant;
  double  max_rpm, min_rpm, air_resistance_coeff, ground_z;
  double  J[9];
  double  allocation_matrix[4 * MRS_MAX_MOTORS]; /* row r, motor m at [r*MRS_MAX_MOTORS + m] */
} mrs_model_params_t;

typedef struct { int32_t desaturation; int32_t _pad; } mrs_mixer_params_t;                       /* controllers/mixer.hpp:14-17 */
typedef struct { double kp, kd, ki; } mrs_rate_params_t;                                        /* controllers/rate_controller.hpp:14-19 */
typedef struct { double kp, kd, ki, max_rate_roll_pitch, max_rate_yaw; } mrs_attitude_params_t; /* controllers/attitude_controller.hpp:14-21 */
typedef struct { double kp, kd, ki, max_acceleration; } mrs_velocity_params_t;                  /* controllers/velocity_controller.hpp:14-20 */
typedef struct { double kp, kd, ki, max_velocity; } mrs_position_params_t;                      /* controllers/position_controller.hpp:14-20 */

/* per-swarm event counters replacing the std::cout warnings of controllers/attitude_controller.hpp:196,236,245
 * and counting the NaN rollbacks of multirotor_model.hpp:228-233 */
typedef struct {
  uint64_t hdg_rate_denom_small;
  uint64_t projected_norm_small;
  uint64_t yaw_rate_not_finite;
  uint64_t nan_rollback;
} mrs_diag_t;

/* what UavSystemRos publishes per UAV and tick (src/uav_system_ros.cpp:342-466) and MultirotorSimulator::publishPoses
 * (src/multirotor_simulator.cpp:365-389), derived on the device and downloaded as one packed array */
typedef struct {
  double position[3];            /* odom.pose.pose.position                                  :352-354 */
  double orientation[4];         /* x, y, z, w of mrs_lib::AttitudeConverter(state.R) == Eigen::Quaterniond(R)  :350 */
  double velocity_body[3];       /* odom.twist.twist.linear = R^T v                          :356-360 */
  double angular_velocity[3];    /* odom/imu angular velocity = omega                         :362-364,380-382 */
  double linear_acceleration[3]; /* imu.linear_acceleration = getImuAcceleration()            :384-388 */
  double range;                  /* rangefinder: (z - ground_z)/cos(tilt) + 0.01, >40 -> 41, body_z.z <= 0 -> 41  :403-419 */
} mrs_uav_output_t;

/* MultirotorModel::State (+ what UavSystemRos::makeStep reads right after it: IMU acceleration, crash flag) of one UAV, packed for
 * ONE device-to-host copy of a whole range — multirotor_model.hpp:90-98, src/uav_system_ros.cpp:270-282 */
typedef struct {
  double  x[3], v[3], v_prev[3];
  double  R[9];                  /* row-major */
  double  omega[3];
  double  motor_rpm[MRS_MAX_MOTORS];
  double  imu_acceleration[3];   /* UavSystem::getImuAcceleration — uav_system.hpp:424 */
  int32_t crashed;               /* UavSystem::hasCrashed — uav_system.hpp:286 */
  int32_t n_motors;
} mrs_uav_state_t;

typedef struct mrs_swarm mrs_swarm_t;

/* ---- parameter helpers (host only) ---- */
/* ModelParams::ModelParams() x500 defaults — multirotor_model.hpp:26-66 (ground_z := 0; uninitialised there) */
int mrs_model_params_default(mrs_model_params_t* p);
/* UavSystemRos::calculateInertia — src/uav_system_ros.cpp:664-671 */
int mrs_calculate_inertia(mrs_model_params_t* p);
/* allocation-matrix row scaling applied to the YAML matrix — src/uav_system_ros.cpp:100-103 */
int mrs_scale_allocation(mrs_model_params_t* p);

/* ---- lifetime ---- */
/* std::vector<std::unique_ptr<UavSystemRos>> uavs_ — src/multirotor_simulator.cpp:70,150-157.
 * All UAVs start as UavSystem() (default ctor). device_id < 0 -> current device. */
int mrs_swarm_create(int32_t n_uavs, int32_t device_id, mrs_swarm_t** out);
int mrs_swarm_destroy(mrs_swarm_t* s);
int mrs_swarm_size(const mrs_swarm_t* s, int32_t* n_uavs);
int mrs_swarm_set_arith(mrs_swarm_t* s, int32_t arith);
/* the HIP stream every launch of this swarm goes to (hipStream_t as void*) */
int mrs_swarm_stream(const mrs_swarm_t* s, void** stream);
int mrs_swarm_synchronize(mrs_swarm_t* s);
const char* mrs_last_error(void);

/* ---- construction / parameters ---- */
/* UavSystem ctors — uav_system.hpp:127-153.  params==NULL: UavSystem(void); pos==NULL: UavSystem(params)
 * (no setStatePos); else UavSystem(params, spawn_pos, spawn_heading).  pos: count x 3, heading: count. */
int mrs_swarm_construct(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_model_params_t* params,
                        const double* pos, const double* heading);
/* UavSystem::setParams — uav_system.hpp:404-409 (re-creates all controllers with DEFAULT gains, fresh PIDs) */
int mrs_swarm_set_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_model_params_t* params);
/* UavSystem::getParams — uav_system.hpp:395 (takeoff_patch_enabled reflects the flag the step may have cleared) */
int mrs_swarm_get_params(mrs_swarm_t* s, int32_t uav, mrs_model_params_t* out);
/* UavSystem::set{Mixer,RateController,AttitudeController,VelocityController,PositionController}Params —
 * uav_system.hpp:433-451; each resets the PIDs of that controller */
int mrs_swarm_set_mixer_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_mixer_params_t* p);
int mrs_swarm_set_rate_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_rate_params_t* p);
int mrs_swarm_set_attitude_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_attitude_params_t* p);
int mrs_swarm_set_velocity_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_velocity_params_t* p);
int mrs_swarm_set_position_params(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_position_params_t* p);
/* UavSystem::getMixerAllocation — uav_system.hpp:415 (n_motors x 4, row-major) */
int mrs_swarm_get_mixer_allocation(mrs_swarm_t* s, int32_t uav, double* out);

/* ---- commands ---- */
/* UavSystem::setInput(...) x11 — uav_system.hpp:175-248.  payload: count x stride doubles per UAV:
 *   ACTUATOR: motors[n_motors] | CONTROL_GROUP: roll,pitch,yaw,throttle | ATTITUDE_RATE: rx,ry,rz,throttle
 *   ATTITUDE: R[9] row-major, throttle | TILT_HDG_RATE: tilt[3], heading_rate, throttle
 *   ACCELERATION_HDG(_RATE) / VELOCITY_HDG(_RATE) / POSITION: vec[3], heading(_rate) | INPUT_UNKNOWN: none */
int mrs_swarm_set_input(mrs_swarm_t* s, int32_t first, int32_t count, int32_t mode, const double* payload, int32_t stride);
/* Staged form of mrs_swarm_set_input for hosts that refresh every command each tick (the subscriber callbacks of
 * src/uav_system_ros.cpp:679-1022, batched): mrs_swarm_input_staging hands out pinned host memory for count rows of `stride`
 * doubles (row k = the setInput payload of UAV first+k, layouts as above); the caller fills it and mrs_swarm_commit_input sends it
 * with one asynchronous copy (on a copy stream, beside the running step) + one unpack kernel on the swarm's stream — no pageable
 * staging, no per-column copies.  Two row blocks are handed out in turn: a block returned by mrs_swarm_input_staging belongs to
 * the caller until the commit that follows; the call waits only for the COPY of the commit two calls back, never for a step. */
int mrs_swarm_input_staging(mrs_swarm_t* s, int32_t count, int32_t stride, double** rows);
int mrs_swarm_commit_input(mrs_swarm_t* s, int32_t first, int32_t count, int32_t mode, int32_t stride);

/* UavSystem::setFeedforward(...) x4 — uav_system.hpp:254-272.  payload: vec[3], heading(_rate) */
int mrs_swarm_set_feedforward(mrs_swarm_t* s, int32_t first, int32_t count, int32_t kind, const double* payload, int32_t stride);
/* UavSystem::applyForce — uav_system.hpp:295; force: count x 3 */
int mrs_swarm_apply_force(mrs_swarm_t* s, int32_t first, int32_t count, const double* force);
/* UavSystem::crash / hasCrashed — uav_system.hpp:278,286 */
int mrs_swarm_crash(mrs_swarm_t* s, int32_t first, int32_t count);
/* UavSystemRos::makeStep iterates the model only `if (_iterate_without_input_ || time_last_input_ > 0)` — src/uav_system_ros.cpp:265.
 * hold != 0 excludes the UAVs from mrs_swarm_step* / tick (state, PIDs and IMU stay as they are); collisions still see them. */
int mrs_swarm_set_hold(mrs_swarm_t* s, int32_t first, int32_t count, int32_t hold);
int mrs_swarm_has_crashed(mrs_swarm_t* s, int32_t first, int32_t count, int32_t* out);

/* ---- UavSystemRos semantics that touch device state ---- */
/* UavSystemRos::timeoutInput — src/uav_system_ros.cpp:474-647: the command of every UAV in [first, first+count) is replaced
 * by the safe command of its current input mode (hold position / zero velocity / level attitude / zero rates / zero
 * motors), computed on the device from the current state; heading = mrs_lib::AttitudeConverter(R).getHeading() */
int mrs_swarm_timeout_input(mrs_swarm_t* s, int32_t first, int32_t count);
/* UavSystemRos::callbackSetMass — :1028-1053 (allocation row 2 rescaled by m_new/m_old, inertia recomputed, setParams:
 * controllers fall back to DEFAULT gains with fresh PIDs) */
int mrs_swarm_set_mass(mrs_swarm_t* s, int32_t first, int32_t count, double mass);
/* UavSystemRos::callbackSetGroundZ — :1055-1080 (also through setParams: gains reset) */
int mrs_swarm_set_ground_z(mrs_swarm_t* s, int32_t first, int32_t count, double ground_z);

/* ---- the hot path ---- */
/* for (i) uavs_[i]->makeStep(dt) — src/multirotor_simulator.cpp:211-213 -> UavSystem::makeStep, uav_system.hpp:304-380.
 * Asynchronous on the swarm's stream. */
int mrs_swarm_step(mrs_swarm_t* s, double dt);
/* uavs_[i]->makeStep(dt) for the UAVs [first, first + count) ONLY — src/multirotor_simulator.cpp:212 outside a whole-swarm round
 * (a UAV whose inputs changed after the round's launch, a host that steps one UAV on its own).  Same results as a whole-swarm step
 * of those UAVs; the neighbour lists of the collision pass are rebuilt at the next collision tick. */
int mrs_swarm_step_range(mrs_swarm_t* s, int32_t first, int32_t count, double dt);
/* n_steps consecutive makeStep(dt) rounds; substeps_per_launch > 1 keeps the state in registers across that many
 * steps inside one launch (legal while commands are constant and collisions are off; results identical). */
int mrs_swarm_step_n(mrs_swarm_t* s, double dt, int32_t n_steps, int32_t substeps_per_launch);
/* MultirotorSimulator::handleCollisions — src/multirotor_simulator.cpp:295-359 (kd-tree replaced by a spatial hash) */
int mrs_swarm_handle_collisions(mrs_swarm_t* s, int32_t enabled, int32_t crash, double rebounce);
/* n_ticks of the timerMain order: makeStep for all, then handleCollisions — src/multirotor_simulator.cpp:211-217 */
int mrs_swarm_tick_n(mrs_swarm_t* s, double dt, int32_t n_ticks, int32_t enabled, int32_t crash, double rebounce);

/* ---- state access ---- */
/* UavSystem::getState — uav_system.hpp:386 / MultirotorModel::State multirotor_model.hpp:90-98.  Any pointer may be
 * NULL.  x,v,v_prev,omega: count x 3; R: count x 9 row-major; motor_rpm: count x MRS_MAX_MOTORS. */
int mrs_swarm_get_state(mrs_swarm_t* s, int32_t first, int32_t count, double* x, double* v, double* v_prev, double* R,
                        double* omega, double* motor_rpm);
/* the same for a whole range as packed records: one pack kernel, one device-to-host copy (what a per-UAV loop of getState() calls
 * over a pool of UavSystem objects is served from: uav_system.hpp UavPool) */
int mrs_swarm_get_states(mrs_swarm_t* s, int32_t first, int32_t count, mrs_uav_state_t* out);
/* MultirotorModel::setState — multirotor_model.hpp:424-433 (v_prev untouched, like the reference) */
int mrs_swarm_set_state(mrs_swarm_t* s, int32_t first, int32_t count, const double* x, const double* v, const double* R,
                        const double* omega, const double* motor_rpm);
/* MultirotorModel::setStatePos — multirotor_model.hpp:439-446: x, R = AngleAxis(-heading, z) and _initial_pos_; everything else stays */
int mrs_swarm_set_state_pos(mrs_swarm_t* s, int32_t first, int32_t count, const double* pos, const double* heading);
/* the controllers' PID state (layout of mrs_swarm_get_pid): the reference has no accessor for it — needed to copy a UavSystem */
int mrs_swarm_set_pid(mrs_swarm_t* s, int32_t first, int32_t count, const double* pid);
/* an independent copy of the whole swarm on the same device: state, commands, feed-forwards, PIDs, parameters, flags.  The
 * reference's UavSystem is a copy-assignable value (src/uav_system_ros.cpp:105); collision bookkeeping starts afresh in the copy. */
int mrs_swarm_clone(mrs_swarm_t* s, mrs_swarm_t** out);
/* the same with room for more UAVs: the first mrs_swarm_size(s) UAVs are copies, the others UavSystem() — how a pool of
 * UavSystem objects grows (include/mrs_multirotor_simulator/uav_system/uav_system.hpp UavPool) */
int mrs_swarm_clone_resized(mrs_swarm_t* s, int32_t n_uavs, mrs_swarm_t** out);
/* UavSystem copy-assignment between batches: UAVs [src_first, src_first + count) of `src` replace [dst_first, ...) of `dst` — state,
 * command, feed-forwards, PIDs, flags and parameter set.  The swarms must be clones of each other (mrs_swarm_clone[_resized]: their
 * parameter tables agree on every index in use; MRS_ERR_TYPES otherwise) on the same device; ranges of one swarm must not overlap. */
int mrs_swarm_copy_uavs(mrs_swarm_t* dst, int32_t dst_first, mrs_swarm_t* src, int32_t src_first, int32_t count);
/* UavSystem::getImuAcceleration — uav_system.hpp:424 */
int mrs_swarm_get_imu(mrs_swarm_t* s, int32_t first, int32_t count, double* imu);
/* MultirotorModel::getExternalForce — multirotor_model.hpp:452 */
int mrs_swarm_get_external_force(mrs_swarm_t* s, int32_t first, int32_t count, double* force);
/* PID internals for parity checks: count x 24 = {position,velocity,attitude,rate} x {x,y,z} x {last_error, integral} */
int mrs_swarm_get_pid(mrs_swarm_t* s, int32_t first, int32_t count, double* pid);
int mrs_swarm_get_diag(mrs_swarm_t* s, mrs_diag_t* out);
/* publishOdometry + publishIMU + publishRangefinder + publishPoses payloads of UAVs [first, first+count): one pack kernel,
 * one device-to-host copy (src/uav_system_ros.cpp:342-431, src/multirotor_simulator.cpp:365-389) */
int mrs_swarm_get_outputs(mrs_swarm_t* s, int32_t first, int32_t count, mrs_uav_output_t* out);
/* the same payloads without the final host copy: *view points into the library's pinned staging buffer and stays valid until the
 * next mrs_swarm_get_outputs* call on this swarm (publishers fill their messages straight from it) */
int mrs_swarm_get_outputs_view(mrs_swarm_t* s, int32_t first, int32_t count, const mrs_uav_output_t** view);
/* The same payloads PIPELINED with the steps — what a loop that publishes every UAV's odometry / IMU / range every step
 * (src/uav_system_ros.cpp:278-282) and the pose array every tick (src/multirotor_simulator.cpp:215,365-389) should call.
 * mrs_swarm_get_outputs_async returns at once: the pack kernel is queued on the swarm's stream behind every step queued so far, the
 * device-to-host copy runs on a copy stream into one of two pinned blocks.  mrs_swarm_outputs_wait blocks until THAT copy has
 * landed (an event, not a stream synchronisation: steps queued after the _async call keep running — the download of tick t overlaps
 * step t + 1) and hands out the block; it stays valid until the second _async call after this ticket's.  At most two tickets are in
 * flight.  Collision ticks evaluated lazily by the next step launch (mrs_swarm_tick_n) stay lazy: if a launch before the pack turned
 * out to be a no-op (stale neighbour lists), the wait repeats search, launches and pack before it returns — the payload is always
 * the state after the tick the caller packed behind. */
int mrs_swarm_get_outputs_async(mrs_swarm_t* s, int32_t first, int32_t count, int32_t* ticket);
int mrs_swarm_outputs_wait(mrs_swarm_t* s, int32_t ticket, const mrs_uav_output_t** view, int32_t* count);

/* ---- multi-GPU collision exchange (one swarm shard per process/GPU) ---- */
/* device pointer + byte size of this shard's packed {x,y,z,mass,arm_length,prop_radius} records (48 B/UAV), refreshed by
 * mrs_swarm_pack_positions; the caller all-gathers them (RCCL) into a buffer of n_total records */
int mrs_swarm_pack_positions(mrs_swarm_t* s, void** dev_ptr, int64_t* n_bytes);
/* same records written to caller-owned device memory (e.g. the send buffer of an RCCL all-gather): n_uavs x 48 B */
int mrs_swarm_pack_positions_to(mrs_swarm_t* s, void* dev_dst);
/* handleCollisions for this shard against ALL gathered records (device pointer, n_total x 48 B);
 * my_offset = index of this shard's first UAV in the gathered order */
int mrs_swarm_handle_collisions_gathered(mrs_swarm_t* s, const void* dev_records, int64_t n_total, int64_t my_offset,
                                         int32_t enabled, int32_t crash, double rebounce);

/* The same exchange done by the library itself, for hosts that do not want to drive the collective: it is issued on the swarm's own
 * stream between the kernels, so a whole run of ticks is one call.  Two exchanges exist (mrs_swarm_set_exchange):
 *   MRS_EXCHANGE_EXPORT_SETS (default) — SURVEY 8e v2, "all-gather of boundary-UAV positions": a tick that repeats the neighbour
 *       search gathers the records another rank can list (those inside its bounding box of the last search; 64 B each — the first
 *       search, and any search whose halo turns out too small, gathers all 48-B records: mrs_swarm_get_search_stats); every tick until
 *       the next search gathers only the UAVs some other rank lists (32 B each, padded to the largest export set), and the collision
 *       tick is evaluated by the next step kernel (as on one GPU);
 *   MRS_EXCHANGE_FULL_GATHER — all 48-B records on every tick.
 * Results are identical.  Shards are equal-count index ranges of the caller's (spatially sorted, see mrs_slab_partition) order.
 * Collective backends:
 *   RCCL, bound at run time from `librccl_path` (NULL = "librccl.so" from the loader path; a process that already holds a HIP
 *       runtime — PyTorch-ROCm ships its own — must name the librccl.so that belongs to THAT runtime):
 *         mrs_rccl_unique_id   : rank 0 creates the 128-byte id and hands it to the other ranks by any host channel
 *         mrs_swarm_comm_init  : collective; this swarm must hold the shard of `rank` (n_total / world UAVs, the first
 *                                n_total % world ranks one more)
 *   a caller-supplied all-gather (mrs_swarm_comm_init_custom): `fn` must enqueue, on `stream`, the all-gather of `bytes_per_rank`
 *       bytes from `send` into `recv` (rank-major) and return 0; every rank calls it the same number of times in the same order
 *   an in-process group (mrs_loopback_group_*): `world` swarms of ONE process, each driven by its own host thread, on one device
 *       ("virtual shards", what the tests use on the one-GPU box) or on several devices of a node without RCCL
 *   mrs_swarm_tick_sharded_n : n_ticks of timerMain on every rank — makeStep, then handleCollisions over ALL n_total UAVs
 *                          (src/multirotor_simulator.cpp:211-217, 295-359); collective; returns with every tick evaluated
 *   mrs_swarm_comm_destroy   : collective */
typedef int (*mrs_allgather_fn)(void* user, const void* send, void* recv, uint64_t bytes_per_rank, void* stream);
typedef struct mrs_loopback_group mrs_loopback_group_t;
int mrs_rccl_unique_id(const char* librccl_path, uint8_t* id128);
int mrs_swarm_comm_init(mrs_swarm_t* s, const char* librccl_path, int32_t world, int32_t rank, const uint8_t* id128, int64_t n_total);
int mrs_swarm_comm_init_custom(mrs_swarm_t* s, int32_t world, int32_t rank, int64_t n_total, mrs_allgather_fn fn, void* user);
int mrs_loopback_group_create(int32_t world, mrs_loopback_group_t** out);
int mrs_loopback_group_destroy(mrs_loopback_group_t* g);
int mrs_swarm_comm_init_loopback(mrs_swarm_t* s, mrs_loopback_group_t* g, int32_t rank, int64_t n_total);
/* test hooks of the sharded tick (tests/test_sharded_chaos_gpu.py):
 *   mrs_loopback_group_set_rendezvous : the group's all-gather without its two host barriers — a rank waits only until every peer
 *       has ARRIVED at the same collective (before the group's first collective);
 *   mrs_swarm_debug_chaos : this rank's host sleeps a random 0..max_sleep_us before every launch and, at random, decides on the
 *       stall / warning words as it read them one launch earlier — host skew the protocol must tolerate (0 switches it off);
 *   mrs_swarm_get_split_stats : ticks this rank ran in the split form (interior and boundary launches on two streams) and the
 *       64-UAV blocks its boundary launch covers since the last search */
int mrs_loopback_group_set_rendezvous(mrs_loopback_group_t* g, int32_t on);
int mrs_swarm_debug_chaos(mrs_swarm_t* s, int32_t max_sleep_us, uint64_t seed);
int mrs_swarm_get_split_stats(mrs_swarm_t* s, int64_t* split_ticks, int64_t* boundary_blocks);
/* Neighbour searches of the export-set exchange on this rank: all of them; the ones that ran on a HALO exchange — each rank sends the
 * records that lie inside another rank's box of the last search (plus the distance a UAV may have moved), 64 B each, instead of
 * gathering all 48-B records (MRS_SEARCH_HALO=0 switches that off); the halo searches that had to be repeated on all records (a UAV
 * further from its rank's old hull than the margin, or more entries than the block held); the entries per rank the next one sends.
 * Results do not depend on which exchange a search used. */
int mrs_swarm_get_search_stats(mrs_swarm_t* s, int64_t* searches, int64_t* halo_searches, int64_t* halo_repeats, int64_t* halo_capacity);
/* test / measurement hook: a kernel that keeps `stream` (a hipStream_t of this process) busy for `microseconds` — stands in for the
 * latency of a collective in tools/sharded_rank_cost.py */
int mrs_debug_stream_delay(void* stream, double microseconds);
/* measurement stand-in for ONE rank of a `world`-rank sharded swarm alone on a device (tools/sharded_rank_cost.py): every collective
 * takes `collective_latency_us` of stream time, and the rank's neighbours in the slab order are periodic images of itself
 * `slab_width` metres away — boundary sets, launches and buffer sizes of the real run, none of its physics across the slab faces.
 * Not a simulation backend. */
int mrs_swarm_comm_init_standin(mrs_swarm_t* s, int32_t world, int32_t rank, int64_t n_total, double collective_latency_us, double slab_width);
/* Peer-window exchange — the collectives of the sharded tick as direct writes into the peers' device memory over xGMI, no
 * collective library and no host in the tick (one small kernel per collective on the swarm's stream: push into every peer's
 * window, signal, wait for every peer's signal, pull — csrc/collide.hip k_peer_allgather).  xGMI is point to point: a rank's block
 * reaches every peer in ONE hop, where a ring all-gather pays 2 (world - 1) hops behind its own kernel launch.
 *   mrs_swarm_peer_window_create : allocates this rank's window (4096 + 2 * world * slot bytes, slot = the largest shard's full
 *       gather) and returns its address (`window`, for peers in the same process) and / or its 64-byte IPC handle (`ipc_handle64`,
 *       for peers in other processes: hipIpcMemHandle_t) — either may be NULL;
 *   mrs_swarm_comm_init_peer     : binds the communicator once the caller has carried the addresses / handles to every rank by any
 *       host channel: `windows[q]` (if given and not NULL) is rank q's window as THIS process addresses it (same process, or a
 *       device with peer access enabled), otherwise `ipc_handles + 64 q` is opened.  The entries of the own rank are ignored.
 * Afterwards mrs_swarm_tick_sharded_n / mrs_swarm_comm_destroy as with any other backend (destroy only after every rank's last
 * tick call has returned: peers write into the window until then).  A rank that waits 10 s for a peer's block gives up and the
 * call returns MRS_ERR_HIP.  Ranks of one process must sit on DIFFERENT devices (peer access enabled by the caller): on one device the
 * kernels of different ranks wait for each other, and any runtime call of one rank's host that waits for the whole device (hipFree in a
 * search that grows a buffer) then waits for a peer's kernel that waits for this rank. */
int mrs_swarm_peer_window_create(mrs_swarm_t* s, int32_t world, int32_t rank, int64_t n_total, void** window, uint8_t* ipc_handle64);
int mrs_swarm_comm_init_peer(mrs_swarm_t* s, void* const* windows, const uint8_t* ipc_handles);
int mrs_swarm_set_exchange(mrs_swarm_t* s, int32_t exchange);
int mrs_swarm_tick_sharded_n(mrs_swarm_t* s, double dt, int32_t n_ticks, int32_t enabled, int32_t crash, double rebounce);
int mrs_swarm_comm_destroy(mrs_swarm_t* s);
/* Spatially coherent shards for a swarm addressed by a fixed public index (the reference's uavs_[i]): UAVs sorted by x and cut into
 * `world` equal-count slabs (n_total / world each, the first n_total % world one more).  order[k] = public index of the UAV at
 * position k of the sorted order (rank r holds order[lo_r .. hi_r)); host only. */
int mrs_slab_partition(const double* pos_xyz, int64_t n_total, int32_t world, int64_t* order);
/* A spawn order that follows space, for callers that are free to choose which UAV gets which index (the reference numbers its UAVs in
 * the order of config/uavs.yaml, src/multirotor_simulator.cpp:136-157): order[k] = index, in the caller's numbering, of the UAV that
 * should be spawned k-th, by a Morton key of the neighbour-list cells (edge `cell` metres; <= 0: the library's 2.25 m).  Listed partners
 * and hash buckets of neighbours then share cache lines: -2 % on a collision tick, -7 % on a neighbour search at 100 000 UAVs
 * (profiles/r05_overlapped_ticks_on_ordered_slots.log).  Results do not depend on the order; host only. */
int mrs_cell_order(const double* pos_xyz, int64_t n_total, double cell, int64_t* order);
/* what the communicator of this swarm looks like: ranks as mrs_swarm_comm_init was told and as RCCL itself counts them
 * (ncclCommCount), the exchange in use and the bytes every rank contributes to the per-tick collective */
typedef struct {
  int32_t world, rank;
  int32_t rccl_ranks;       /* ncclCommCount of the communicator (0: no RCCL communicator, e.g. an in-process loopback group) */
  int32_t exchange;         /* MRS_EXCHANGE_* */
  int64_t n_total;
  int64_t bytes_per_tick;   /* bytes this rank sends into the collision collective of an ordinary tick */
  int64_t bytes_per_rebuild; /* bytes it sends on a tick that repeats the neighbour search (export-set exchange only) */
  int64_t export_count, export_capacity; /* export-set exchange: own UAVs some other rank lists / slots of the padded collective */
  int64_t ticks, searches, noop_ticks;   /* sharded ticks so far, how many repeated the search, launches replayed after a stale-list tick */
} mrs_comm_info_t;
int mrs_swarm_comm_info(mrs_swarm_t* s, mrs_comm_info_t* out);

/* collision-pass statistics of mrs_swarm_handle_collisions / mrs_swarm_tick_n: ticks that ran the pass, and how many of them had to
 * repeat the neighbour search (the others reused the neighbour lists of an earlier tick — same results as the reference's per-tick
 * kd-tree, src/multirotor_simulator.cpp:303-317, which this replaces) */
int mrs_swarm_get_collision_stats(mrs_swarm_t* s, int64_t* n_ticks, int64_t* n_rebuilds);

/* how the collision ticks of mrs_swarm_tick_n / step + handle_collisions were evaluated: by the following step launch (fused), how often
 * a launch found the neighbour lists stale (a UAV had left its skin), how many queued launches had to be issued again after it, and
 * how many searches the host queued ahead of time (a UAV close to the edge of its skin) so that no launch found them stale */
int mrs_swarm_get_fused_stats(mrs_swarm_t* s, int64_t* fused_launches, int64_t* stalls, int64_t* replayed_launches, int64_t* searches_ahead);

/* diagnostic hook: the eight control words of the collision pass's neighbour-list state machine (skin flags of the two tick
 * parities, search counter, table-dirty flags, list-overflow counter — collide.hip); synchronises the stream.  tools/collision_words.py */
int mrs_swarm_debug_collision_words(mrs_swarm_t* s, uint32_t* out8);
/* test hook: runs the cascade kernels' own PID device function (PIDController::update, controllers/pid.hpp:67-96) over
 * caller-given sequences on the GPU, one lane per sequence — row-major [n_seq][n_steps] arrays; params = n_seq x
 * {kp, kd, ki, saturation, antiwindup}; event 1 = reset() before the update, 2 = setSaturation(new_sat) before it.
 * tests/test_parity_gpu.py feeds it the golden vectors recorded from the reference's own class. */
int mrs_debug_pid_sequences(int32_t device_id, int32_t arith, int32_t n_seq, int32_t n_steps, const double* params, const double* err,
                            const double* dt, const double* event, const double* new_sat, double* out);

/* one PIDController::update (controllers/pid.hpp:67-96) for each of n independent controllers, on the GPU, with caller-held state:
 * params = n x {kp, kd, ki, saturation, antiwindup}, state = n x {last_error, integral} (updated in place), err / dt / out = n.
 * Backs the stand-alone PIDController class of the header facade. */
int mrs_debug_pid_update(int32_t device_id, int32_t arith, int32_t n, const double* params, double* state, const double* err, const double* dt,
                         double* out);
/* ONE component of the path for the UAVs [first, first + count), on each UAV's own state (x, v, R, omega, motor_rpm, external force),
 * airframe / controller constants and PID state — the device functions the step kernels are made of, run on their own.  Row k of `in`
 * (in_stride doubles) is the input for UAV first + k, row k of `out` receives the result; the PID-bearing controllers update the
 * UAV's PID state like getControlSignal() mutates the reference's controller objects.  Backs the stand-alone L0 classes of the
 * header facade (MultirotorModel, the controllers) and the per-component parity tests.  Matrices row-major.
 *   component                        in                               out                             reference
 *   MRS_COMP_REORTH                  R[9]                             R * L^-1 [9]                    multirotor_model.hpp:249-253, :314-316
 *   MRS_COMP_MODEL_RHS               x[3] v[3] R[9] omega[3]          derivative, same order [18]     MultirotorModel::operator() :301-366
 *   MRS_COMP_MIXER                   roll pitch yaw throttle          motors[8]                       Mixer::getControlSignal mixer.hpp:107-144
 *   MRS_COMP_POSITION                position ref[3]                  velocity[3]                     position_controller.hpp:73-86
 *   MRS_COMP_VELOCITY                velocity ref[3]                  acceleration[3]                 velocity_controller.hpp:68-102
 *   MRS_COMP_ACCELERATION_HDG        acceleration[3] heading          Rd[9] throttle                  acceleration_controller.hpp:44-97
 *   MRS_COMP_ACCELERATION_HDG_RATE   acceleration[3] heading_rate     tilt[3] heading_rate throttle   acceleration_controller.hpp:103-122
 *   MRS_COMP_ATTITUDE                Rd[9] throttle                   rate[3] throttle                attitude_controller.hpp:79-100
 *   MRS_COMP_TILT_HDG_RATE           tilt[3] heading_rate throttle    rate[3] throttle                attitude_controller.hpp:106-145
 *   MRS_COMP_RATE                    rate[3] throttle                 roll pitch yaw throttle         rate_controller.hpp:67-81 */
enum {
  MRS_COMP_REORTH = 1, MRS_COMP_MODEL_RHS, MRS_COMP_MIXER, MRS_COMP_POSITION, MRS_COMP_VELOCITY, MRS_COMP_ACCELERATION_HDG,
  MRS_COMP_ACCELERATION_HDG_RATE, MRS_COMP_ATTITUDE, MRS_COMP_TILT_HDG_RATE, MRS_COMP_RATE
};
int mrs_swarm_debug_component(mrs_swarm_t* s, int32_t component, int32_t first, int32_t count, const double* in, int32_t in_stride, double* out,
                              int32_t out_stride, double dt);

/* measurement hook: average device time (ms) of ONE neighbour search of the single-GPU collision pass (pack + insert, then the
 * list-building query — what replaces nanoflann's per-tick kd-tree build + radius searches, src/multirotor_simulator.cpp:303-326),
 * `reps` searches back to back between two hipEvents on the swarm's stream.  Latches the forces / crash flags of
 * handleCollisions(true, crash, rebounce) on the current positions. */
int mrs_swarm_debug_search_ms(mrs_swarm_t* s, int32_t reps, int32_t crash, double rebounce, double* avg_ms);

/* test hook: ONE forced neighbour search on the current positions (it latches the forces / crash flags of
 * handleCollisions(true, crash, rebounce)), then the lists it built: count[i] = listed neighbours of UAV i (0 for a UAV with
 * more neighbours than a list holds), nbr[r * n + i] = the r-th of them in ascending index, r < min(count[i], *list_cap); rows >= count[i]
 * hold stale values.  `nbr` holds list_cap_in rows of n entries; *list_cap returns the library's list capacity.  What the lists must
 * hold (a superset of nanoflann's radiusSearch(3.0) result, src/multirotor_simulator.cpp:326): every UAV closer than sqrt(3) + skin. */
int mrs_swarm_debug_neighbour_lists(mrs_swarm_t* s, int32_t crash, double rebounce, uint32_t* count, uint32_t* nbr, int32_t list_cap_in, int32_t* list_cap,
                                    double* list_radius);

/* timing helper: average device time (ms) per step-kernel launch of the last mrs_swarm_step_n / mrs_swarm_tick_n call,
 * measured with hipEvents on the swarm's stream.  mode 1: one event pair around the whole region (elapsed / launches,
 * inter-launch gaps included, no perturbation); mode 2: one pair around every launch (perturbs the region); 0: off */
int mrs_swarm_last_step_kernel_ms(mrs_swarm_t* s, double* avg_ms, int32_t* n_launches);
int mrs_swarm_set_profiling(mrs_swarm_t* s, int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* MRS_SWARM_H */
