/*
 * uav_oracle.h — CPU oracle for the UavSystem::makeStep() hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product path
 * (mrs_multirotor_simulator_amd/, include/) never links or calls it.
 *
 * What it is: a plain-C99, scalar FP64, AoS restatement of the reference's
 * header-only path (citations relative to /root/reference/include/mrs_multirotor_simulator/uav_system):
 *   multirotor_model.hpp   (MultirotorModel: ODE RHS, RK4 step, post-processing)
 *   uav_system.hpp         (UavSystem: input-mode cascade, feed-forward, crash)
 *   controllers/{pid,mixer,...}.hpp (PID, Mixer, Rate/Attitude/Acceleration/Velocity/Position)
 *   ode/boost/numeric/odeint/stepper/runge_kutta4.hpp + detail/generic_rk_*.hpp (RK4 order)
 *   /root/reference/src/multirotor_simulator.cpp:295-359 (handleCollisions)
 *   /root/reference/src/uav_system_ros.cpp:96-105,223-232,664-671 (init sequence helpers)
 * Eigen/Boost arithmetic (absent from this image) is restated operation by
 * operation (Cholesky, cofactor inverse, products left-to-right, pairwise
 * dynamic-vector sums as Eigen's SSE2 redux does them).
 *
 * PARITY PINNING: the dynamics/cascade part is **parity unpinned** — the
 * reference holds no tests, golden vectors or fixtures for this path and its
 * headers cannot be compiled here (Eigen3/Boost missing).  It is pinned only
 * by analytic known-answer tests derived from the reference source
 * (tests/test_oracle_kat.py).  The collision neighbour *set* IS pinned: the
 * reference's own nanoflann kd-tree is compiled from /root/reference into
 * oracle/_ref/ and compared with orc_handle_collisions' predicate.
 */
#ifndef UAV_ORACLE_H
#define UAV_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_MOTORS 8

/* UavSystem::INPUT_MODE, uav_system.hpp:19-32 */
enum {
  ORC_INPUT_UNKNOWN = 0,
  ORC_ACTUATOR_CMD,
  ORC_CONTROL_GROUP_CMD,
  ORC_ATTITUDE_RATE_CMD,
  ORC_ATTITUDE_CMD,
  ORC_TILT_HDG_RATE_CMD,
  ORC_ACCELERATION_HDG_RATE_CMD,
  ORC_ACCELERATION_HDG_CMD,
  ORC_VELOCITY_HDG_RATE_CMD,
  ORC_VELOCITY_HDG_CMD,
  ORC_POSITION_CMD
};

/* feed-forward kinds, uav_system.hpp:112-115 */
enum { ORC_FF_VELOCITY_HDG_RATE = 0, ORC_FF_VELOCITY_HDG, ORC_FF_ACCELERATION_HDG_RATE, ORC_FF_ACCELERATION_HDG };

/* MultirotorModel::ModelParams, multirotor_model.hpp:24-88. Matrices row-major. */
typedef struct {
  int32_t n_motors;
  int32_t ground_enabled;
  int32_t takeoff_patch_enabled;
  int32_t _pad;
  double  g, mass, kf, km, prop_radius, arm_length, body_height, motor_time_constant;
  double  max_rpm, min_rpm, air_resistance_coeff, ground_z;
  double  J[9];
  double  allocation_matrix[4 * ORC_MAX_MOTORS]; /* row r, motor m at [r*ORC_MAX_MOTORS+m] */
} orc_model_params_t;

typedef struct { int32_t desaturation; int32_t _pad; } orc_mixer_params_t;               /* mixer.hpp:14-17 */
typedef struct { double kp, kd, ki; } orc_rate_params_t;                                /* rate_controller.hpp:14-19 */
typedef struct { double kp, kd, ki, max_rate_roll_pitch, max_rate_yaw; } orc_attitude_params_t; /* attitude_controller.hpp:14-21 */
typedef struct { double kp, kd, ki, max_acceleration; } orc_velocity_params_t;          /* velocity_controller.hpp:14-20 */
typedef struct { double kp, kd, ki, max_velocity; } orc_position_params_t;              /* position_controller.hpp:14-20 */

/* diagnostics: the three std::cout warnings of attitude_controller.hpp:196,236,245 + NaN rollbacks */
typedef struct {
  uint64_t hdg_rate_denom_small;   /* :195 */
  uint64_t projected_norm_small;   /* :235 */
  uint64_t yaw_rate_not_finite;    /* :244 */
  uint64_t nan_rollback;           /* multirotor_model.hpp:228-233 */
} orc_diag_t;

/* per-UAV publisher payloads, src/uav_system_ros.cpp:342-431 */
typedef struct {
  double position[3], orientation[4] /* x y z w */, velocity_body[3], angular_velocity[3], linear_acceleration[3], range;
} orc_uav_output_t;

typedef struct orc_swarm orc_swarm_t;

/* defaults of the ModelParams ctor (x500), multirotor_model.hpp:26-66; ground_z := 0 (uninitialised in the reference) */
void orc_model_params_default(orc_model_params_t* p);
/* UavSystemRos::calculateInertia, src/uav_system_ros.cpp:664-671 */
void orc_calculate_inertia(orc_model_params_t* p);
/* allocation row scaling, src/uav_system_ros.cpp:100-103 (in place, on the raw YAML matrix) */
void orc_scale_allocation(orc_model_params_t* p);

orc_swarm_t* orc_swarm_create(int32_t n_uavs);
void         orc_swarm_destroy(orc_swarm_t* s);
int32_t      orc_swarm_size(const orc_swarm_t* s);

/* UavSystem ctors, uav_system.hpp:127-153.  params==NULL -> UavSystem(void); pos==NULL -> 1-arg ctor
 * (no setStatePos, _initial_pos_ stays "uninitialised" := 0). pos is count x 3, heading is count. */
void orc_swarm_construct(orc_swarm_t* s, int32_t first, int32_t count, const orc_model_params_t* params,
                         const double* pos, const double* heading);

/* UavSystem::setParams, uav_system.hpp:404-409 (controllers re-created with default gains) */
void orc_swarm_set_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_model_params_t* params);
void orc_swarm_get_params(const orc_swarm_t* s, int32_t uav, orc_model_params_t* out);

/* controller param setters, uav_system.hpp:433-451 (each resets that controller's PIDs) */
void orc_swarm_set_mixer_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_mixer_params_t* p);
void orc_swarm_set_rate_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_rate_params_t* p);
void orc_swarm_set_attitude_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_attitude_params_t* p);
void orc_swarm_set_velocity_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_velocity_params_t* p);
void orc_swarm_set_position_params(orc_swarm_t* s, int32_t first, int32_t count, const orc_position_params_t* p);

/* UavSystem::setInput overloads, uav_system.hpp:175-248.  payload is count x stride doubles:
 *   ACTUATOR: motors[n_motors]           CONTROL_GROUP: roll,pitch,yaw,throttle
 *   ATTITUDE_RATE: rx,ry,rz,throttle     ATTITUDE: R[9] row-major, throttle
 *   TILT_HDG_RATE: tilt[3],heading_rate,throttle
 *   ACCELERATION_HDG(_RATE), VELOCITY_HDG(_RATE), POSITION: vec[3], heading(_rate)
 *   INPUT_UNKNOWN: payload ignored */
void orc_swarm_set_input(orc_swarm_t* s, int32_t first, int32_t count, int32_t mode, const double* payload, int32_t stride);
/* UavSystem::setFeedforward overloads, uav_system.hpp:254-272.  payload: vec[3], heading(_rate) */
void orc_swarm_set_feedforward(orc_swarm_t* s, int32_t first, int32_t count, int32_t kind, const double* payload, int32_t stride);

/* serial loop of UavSystem::makeStep(dt), src/multirotor_simulator.cpp:211-213 */
void orc_swarm_step(orc_swarm_t* s, double dt);
/* same, n_steps times, optionally split over n_threads pthreads (UAVs are independent) — the cpu_baseline leg */
void orc_swarm_step_n(orc_swarm_t* s, double dt, int32_t n_steps, int32_t n_threads);

/* MultirotorSimulator::handleCollisions, src/multirotor_simulator.cpp:295-359.
 * Neighbour search restated as an exhaustive scan accelerated by a uniform grid; pair predicate literal. */
void orc_swarm_handle_collisions(orc_swarm_t* s, int32_t enabled, int32_t crash, double rebounce);

void orc_swarm_apply_force(orc_swarm_t* s, int32_t first, int32_t count, const double* force);  /* uav_system.hpp:295 */
void orc_swarm_set_hold(orc_swarm_t* s, int32_t first, int32_t count, int32_t hold);              /* src/uav_system_ros.cpp:265 */
void orc_swarm_crash(orc_swarm_t* s, int32_t first, int32_t count);                              /* :278 */
void orc_swarm_has_crashed(const orc_swarm_t* s, int32_t first, int32_t count, int32_t* out);     /* :286 */

/* MultirotorModel::getState / setState (multirotor_model.hpp:416-433); any pointer may be NULL.
 * x,v,v_prev,omega: count x 3; R: count x 9 row-major; motor_rpm: count x ORC_MAX_MOTORS */
void orc_swarm_get_state(const orc_swarm_t* s, int32_t first, int32_t count, double* x, double* v, double* v_prev,
                         double* R, double* omega, double* motor_rpm);
void orc_swarm_set_state(orc_swarm_t* s, int32_t first, int32_t count, const double* x, const double* v,
                         const double* R, const double* omega, const double* motor_rpm);
void orc_swarm_get_imu(const orc_swarm_t* s, int32_t first, int32_t count, double* imu);         /* multirotor_model.hpp:484 */
void orc_swarm_get_external_force(const orc_swarm_t* s, int32_t first, int32_t count, double* f);
/* PID states for inspection: count x 24 = {pos,vel,att,rate} x {x,y,z} x {last_error, integral} */
void orc_swarm_get_pid(const orc_swarm_t* s, int32_t first, int32_t count, double* pid);
void orc_swarm_set_pid(orc_swarm_t* s, int32_t first, int32_t count, const double* pid);
/* Mixer::getAllocationMatrix, mixer.hpp:150: n_motors x 4 row-major */
void orc_swarm_get_mixer_allocation(const orc_swarm_t* s, int32_t uav, double* out);
void orc_swarm_get_diag(const orc_swarm_t* s, orc_diag_t* out);
/* publishOdometry / publishIMU / publishRangefinder payloads (src/uav_system_ros.cpp:342-431).  The orientation is
 * mrs_lib::AttitudeConverter(R) — mrs_lib (ctu-mrs/mrs_lib, version unpinned by package.xml:20) is NOT in the reference tree;
 * its published implementation forwards to Eigen::Quaterniond(R), whose algorithm (Eigen/src/Geometry/Quaternion.h,
 * quaternionbase_assign_impl<Other,3,3>) is restated here. */
/* UavSystemRos::timeoutInput, src/uav_system_ros.cpp:474-647 (mrs_lib/tf2 conversions restated from their published sources) */
void orc_swarm_timeout_input(orc_swarm_t* s, int32_t first, int32_t count);
/* UavSystemRos::callbackSetMass / callbackSetGroundZ, src/uav_system_ros.cpp:1028-1080 */
void orc_swarm_set_mass(orc_swarm_t* s, int32_t first, int32_t count, double mass);
void orc_swarm_set_ground_z(orc_swarm_t* s, int32_t first, int32_t count, double ground_z);
void orc_swarm_get_outputs(const orc_swarm_t* s, int32_t first, int32_t count, orc_uav_output_t* out);

/* building blocks exposed for known-answer tests */
/* one component of the path on the state of UAVs [first, first + count): ids and row layouts of mrs_swarm_debug_component */
void orc_swarm_debug_component(orc_swarm_t* s, int32_t component, int32_t first, int32_t count, const double* in, int32_t in_stride, double* out,
                               int32_t out_stride, double dt);
double orc_pid_update(double kp, double kd, double ki, double saturation, double antiwindup,
                      double* last_error, double* integral, double error, double dt);   /* pid.hpp:67-96 */
void   orc_llt_reorth(const double R[9], double out[9]);  /* R * inverse(matrixL(LLT(R^T R))), multirotor_model.hpp:249-253 */
void   orc_inverse3(const double m[9], double out[9]);    /* Eigen fixed 3x3 inverse (cofactors) */
int    orc_inverse_lu(const double* a, int n, double* out); /* Eigen dynamic inverse (partial-pivot LU) */

#ifdef __cplusplus
}
#endif
#endif
