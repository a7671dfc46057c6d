// uav_system.hpp — drop-in facade for the reference's header API
// (include/mrs_multirotor_simulator/uav_system/uav_system.hpp:16-118) on top of the C ABI of include/mrs_swarm.h.
//
//   mrs_multirotor_simulator::UavSystem   same public methods, same argument types, same semantics; one object is a
//                                          swarm of one UAV on the GPU (slow per call — use it for porting and tests)
//   mrs_multirotor_simulator::UavSwarm    the batch owner the simulator loop should hold instead of
//                                          std::vector<std::unique_ptr<UavSystemRos>> (src/multirotor_simulator.cpp:70):
//                                          swarm[i] is a non-owning view with the UavSystem API, swarm.makeStep(dt)
//                                          replaces the serial loop of src/multirotor_simulator.cpp:211-213 by one launch
//                                          and swarm.handleCollisions(...) replaces :295-359.
//
// All arithmetic happens in libmrs_swarm.so (HIP); this header only marshals values.  Errors of the library surface as
// std::runtime_error (the reference's API is void and cannot fail, the GPU can).
#ifndef UAV_SYSTEM_H
#define UAV_SYSTEM_H

#include <memory>
#include <stdexcept>
#include <utility>
#include <array>
#include <vector>

#include "multirotor_model.hpp"

#include "controllers/pid.hpp"
#include "controllers/mixer.hpp"
#include "controllers/rate_controller.hpp"
#include "controllers/attitude_controller.hpp"
#include "controllers/acceleration_controller.hpp"
#include "controllers/velocity_controller.hpp"
#include "controllers/position_controller.hpp"

namespace mrs_multirotor_simulator
{

// the UavSystem method set, bound to UAV `i_` of swarm `s_` (shared by UavSystem and UavSwarm::Ref)
class UavSystemApi {
public:
  enum INPUT_MODE  // uav_system.hpp:19-32
  {
    INPUT_UNKNOWN,
    ACTUATOR_CMD,
    CONTROL_GROUP_CMD,
    ATTITUDE_RATE_CMD,
    ATTITUDE_CMD,
    TILT_HDG_RATE_CMD,
    ACCELERATION_HDG_RATE_CMD,
    ACCELERATION_HDG_CMD,
    VELOCITY_HDG_RATE_CMD,
    VELOCITY_HDG_CMD,
    POSITION_CMD,
  };

  // uav_system.hpp:304.  On an owning UavSystem (a swarm of one) this is the reference call.  On a view of a larger swarm
  // (UavSwarm::Ref) it is refused: the reference's loop `for (i) uavs_[i]->makeStep(dt)` (src/multirotor_simulator.cpp:211-213)
  // would otherwise step the WHOLE swarm once per UAV — call UavSwarm::makeStep(dt) once instead.
  void makeStep(const double dt) {
    int32_t n = 0;
    mrs_throw_on_error(mrs_swarm_size(s_, &n));
    if (n != 1) throw std::logic_error("UavSwarm::Ref::makeStep steps every UAV of the swarm: call UavSwarm::makeStep(dt) once per tick instead");
    mrs_throw_on_error(mrs_swarm_step(s_, dt));
  }

  void crash(void) { mrs_throw_on_error(mrs_swarm_crash(s_, i_, 1)); }  // :278
  bool hasCrashed(void) {                                               // :286
    int32_t c = 0;
    mrs_throw_on_error(mrs_swarm_has_crashed(s_, i_, 1, &c));
    return c != 0;
  }

  void applyForce(const Eigen::Vector3d& force) {  // :295
    const double f[3] = {force(0), force(1), force(2)};
    mrs_throw_on_error(mrs_swarm_apply_force(s_, i_, 1, f));
  }

  // ---- setInput x11, uav_system.hpp:175-248 ----
  void setInput(const reference::Actuators& cmd) {
    double p[MRS_MAX_MOTORS] = {0};
    for (int m = 0; m < cmd.motors.size() && m < MRS_MAX_MOTORS; m++) p[m] = cmd.motors(m);
    input(MRS_ACTUATOR_CMD, p, MRS_MAX_MOTORS);
  }
  void setInput(const reference::ControlGroup& cmd) {
    const double p[4] = {cmd.roll, cmd.pitch, cmd.yaw, cmd.throttle};
    input(MRS_CONTROL_GROUP_CMD, p, 4);
  }
  void setInput(const reference::AttitudeRate& cmd) {
    const double p[4] = {cmd.rate_x, cmd.rate_y, cmd.rate_z, cmd.throttle};
    input(MRS_ATTITUDE_RATE_CMD, p, 4);
  }
  void setInput(const reference::Attitude& cmd) {
    double p[10];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) p[r * 3 + c] = cmd.orientation(r, c);
    p[9] = cmd.throttle;
    input(MRS_ATTITUDE_CMD, p, 10);
  }
  void setInput(const reference::TiltHdgRate& cmd) {
    const double p[5] = {cmd.tilt_vector(0), cmd.tilt_vector(1), cmd.tilt_vector(2), cmd.heading_rate, cmd.throttle};
    input(MRS_TILT_HDG_RATE_CMD, p, 5);
  }
  void setInput(const reference::AccelerationHdgRate& cmd) { input4(MRS_ACCELERATION_HDG_RATE_CMD, cmd.acceleration, cmd.heading_rate); }
  void setInput(const reference::AccelerationHdg& cmd) { input4(MRS_ACCELERATION_HDG_CMD, cmd.acceleration, cmd.heading); }
  void setInput(const reference::VelocityHdgRate& cmd) { input4(MRS_VELOCITY_HDG_RATE_CMD, cmd.velocity, cmd.heading_rate); }
  void setInput(const reference::VelocityHdg& cmd) { input4(MRS_VELOCITY_HDG_CMD, cmd.velocity, cmd.heading); }
  void setInput(const reference::Position& cmd) { input4(MRS_POSITION_CMD, cmd.position, cmd.heading); }
  void setInput(void) { input(MRS_INPUT_UNKNOWN, nullptr, 0); }

  // ---- setFeedforward x4, uav_system.hpp:254-272 ----
  void setFeedforward(const reference::AccelerationHdgRate& cmd) { ff(MRS_FF_ACCELERATION_HDG_RATE, cmd.acceleration, cmd.heading_rate); }
  void setFeedforward(const reference::AccelerationHdg& cmd) { ff(MRS_FF_ACCELERATION_HDG, cmd.acceleration, cmd.heading); }
  void setFeedforward(const reference::VelocityHdg& cmd) { ff(MRS_FF_VELOCITY_HDG, cmd.velocity, cmd.heading); }
  void setFeedforward(const reference::VelocityHdgRate& cmd) { ff(MRS_FF_VELOCITY_HDG_RATE, cmd.velocity, cmd.heading_rate); }

  MultirotorModel::State getState(void) {  // :386
    double x[3], v[3], vp[3], R[9], w[3], rpm[MRS_MAX_MOTORS];
    mrs_throw_on_error(mrs_swarm_get_state(s_, i_, 1, x, v, vp, R, w, rpm));
    mrs_model_params_t p;
    mrs_throw_on_error(mrs_swarm_get_params(s_, i_, &p));
    MultirotorModel::State st;
    st.x = Eigen::Vector3d(x[0], x[1], x[2]);
    st.v = Eigen::Vector3d(v[0], v[1], v[2]);
    st.v_prev = Eigen::Vector3d(vp[0], vp[1], vp[2]);
    st.omega  = Eigen::Vector3d(w[0], w[1], w[2]);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) st.R(r, c) = R[r * 3 + c];
    st.motor_rpm = Eigen::VectorXd::Zero(p.n_motors);
    for (int m = 0; m < p.n_motors; m++) st.motor_rpm(m) = rpm[m];
    return st;
  }

  // UavSystemRos::getPose (include/mrs_multirotor_simulator/uav_system_ros.h:48, src/uav_system_ros.cpp:289-292): the position
  Eigen::Vector3d getPose(void) {
    double x[3];
    mrs_throw_on_error(mrs_swarm_get_state(s_, i_, 1, x, nullptr, nullptr, nullptr, nullptr, nullptr));
    return Eigen::Vector3d(x[0], x[1], x[2]);
  }

  MultirotorModel::ModelParams getParams(void) {  // :395
    mrs_model_params_t c;
    mrs_throw_on_error(mrs_swarm_get_params(s_, i_, &c));
    MultirotorModel::ModelParams p;
    p.fromC(c);
    return p;
  }

  void setParams(const MultirotorModel::ModelParams& params) {  // :404 (controllers fall back to default gains)
    const mrs_model_params_t c = params.toC();
    mrs_throw_on_error(mrs_swarm_set_params(s_, i_, 1, &c));
  }

  Eigen::Vector3d getImuAcceleration(void) {  // :424
    double a[3];
    mrs_throw_on_error(mrs_swarm_get_imu(s_, i_, 1, a));
    return Eigen::Vector3d(a[0], a[1], a[2]);
  }

  void setMixerParams(const Mixer::Params& params) {  // :433-451
    const mrs_mixer_params_t c{params.desaturation ? 1 : 0, 0};
    mrs_throw_on_error(mrs_swarm_set_mixer_params(s_, i_, 1, &c));
  }
  void setRateControllerParams(const RateController::Params& params) {
    const mrs_rate_params_t c{params.kp, params.kd, params.ki};
    mrs_throw_on_error(mrs_swarm_set_rate_params(s_, i_, 1, &c));
  }
  void setAttitudeControllerParams(const AttitudeController::Params& params) {
    const mrs_attitude_params_t c{params.kp, params.kd, params.ki, params.max_rate_roll_pitch, params.max_rate_yaw};
    mrs_throw_on_error(mrs_swarm_set_attitude_params(s_, i_, 1, &c));
  }
  void setVelocityControllerParams(const VelocityController::Params& params) {
    const mrs_velocity_params_t c{params.kp, params.kd, params.ki, params.max_acceleration};
    mrs_throw_on_error(mrs_swarm_set_velocity_params(s_, i_, 1, &c));
  }
  void setPositionControllerParams(const PositionController::Params& params) {
    const mrs_position_params_t c{params.kp, params.kd, params.ki, params.max_velocity};
    mrs_throw_on_error(mrs_swarm_set_position_params(s_, i_, 1, &c));
  }

  Eigen::MatrixXd getMixerAllocation(void) {  // :415
    mrs_model_params_t p;
    mrs_throw_on_error(mrs_swarm_get_params(s_, i_, &p));
    double a[MRS_MAX_MOTORS * 4];
    mrs_throw_on_error(mrs_swarm_get_mixer_allocation(s_, i_, a));
    Eigen::MatrixXd m = Eigen::MatrixXd::Zero(p.n_motors, 4);
    for (int r = 0; r < p.n_motors; r++)
      for (int c = 0; c < 4; c++) m(r, c) = a[r * 4 + c];
    return m;
  }

protected:
  UavSystemApi(mrs_swarm_t* s, int i) : s_(s), i_(i) {}
  mrs_swarm_t* s_;
  int          i_;

private:
  void input(int mode, const double* p, int n) { mrs_throw_on_error(mrs_swarm_set_input(s_, i_, 1, mode, p, n)); }
  void input4(int mode, const Eigen::Vector3d& v, double h) {
    const double p[4] = {v(0), v(1), v(2), h};
    input(mode, p, 4);
  }
  void ff(int kind, const Eigen::Vector3d& v, double h) {
    const double p[4] = {v(0), v(1), v(2), h};
    mrs_throw_on_error(mrs_swarm_set_feedforward(s_, i_, 1, kind, p, 4));
  }
};

// ------------------------------------------------------------------------------------------------------------------
// the batch owner
// ------------------------------------------------------------------------------------------------------------------
class UavSwarm {
public:
  class Ref : public UavSystemApi {  // view of one UAV (non-owning); makeStep() is refused on it, see UavSystemApi::makeStep
  public:
    Ref(mrs_swarm_t* s, int i) : UavSystemApi(s, i) {}
  };

  explicit UavSwarm(int n_uavs, int device_id = -1, bool fast_arithmetic = true) : s_(nullptr) {
    mrs_throw_on_error(mrs_swarm_create(n_uavs, device_id, &s_));
    mrs_throw_on_error(mrs_swarm_set_arith(s_, fast_arithmetic ? MRS_ARITH_FAST : MRS_ARITH_LITERAL));
  }
  explicit UavSwarm(mrs_swarm_t* adopt) : s_(adopt) {}  // takes ownership of a library swarm (e.g. mrs_swarm_clone)
  ~UavSwarm() { mrs_swarm_destroy(s_); }
  UavSwarm(const UavSwarm&) = delete;
  UavSwarm& operator=(const UavSwarm&) = delete;

  int size() const {
    int32_t n = 0;
    mrs_swarm_size(s_, &n);
    return n;
  }
  mrs_swarm_t* handle() { return s_; }
  Ref          operator[](int i) { return Ref(s_, i); }
  Ref          at(int i) { return Ref(s_, i); }

  // UavSystem(model_params, spawn_pos, spawn_heading) for UAVs [first, first+count) — uav_system.hpp:144-153
  void construct(int first, int count, const MultirotorModel::ModelParams& params, const std::vector<Eigen::Vector3d>& spawn_pos,
                 const std::vector<double>& spawn_heading) {
    const mrs_model_params_t c = params.toC();
    std::vector<double>      p((size_t)count * 3);
    for (int k = 0; k < count; k++)
      for (int j = 0; j < 3; j++) p[(size_t)k * 3 + j] = spawn_pos[(size_t)k](j);
    mrs_throw_on_error(mrs_swarm_construct(s_, first, count, &c, p.data(), spawn_heading.data()));
  }

  void makeStep(double dt) { mrs_throw_on_error(mrs_swarm_step(s_, dt)); }  // src/multirotor_simulator.cpp:211-213
  void makeSteps(double dt, int n_steps, int substeps_per_launch = 1) { mrs_throw_on_error(mrs_swarm_step_n(s_, dt, n_steps, substeps_per_launch)); }
  void handleCollisions(bool enabled, bool crash, double rebounce) {  // src/multirotor_simulator.cpp:295-359
    mrs_throw_on_error(mrs_swarm_handle_collisions(s_, enabled, crash, rebounce));
  }
  void tick(double dt, int n_ticks, bool enabled, bool crash, double rebounce) {  // timerMain order, :211-217
    mrs_throw_on_error(mrs_swarm_tick_n(s_, dt, n_ticks, enabled, crash, rebounce));
  }
  // ---- one shard per GPU/process: this swarm holds the UAVs of `rank` out of n_total, collisions act across all shards ----
  // rank 0 creates the id and hands it to the others (any host channel); librccl_path = nullptr loads the system librccl.so
  static std::array<uint8_t, 128> commUniqueId(const char* librccl_path = nullptr) {
    std::array<uint8_t, 128> id{};
    mrs_throw_on_error(mrs_rccl_unique_id(librccl_path, id.data()));
    return id;
  }
  void commInit(int world, int rank, const std::array<uint8_t, 128>& id, int64_t n_total, const char* librccl_path = nullptr) {
    mrs_throw_on_error(mrs_swarm_comm_init(s_, librccl_path, world, rank, id.data(), n_total));
  }
  // timerMain x n_ticks on every rank: makeStep, all-gather of the 48-B records on the swarm's stream, handleCollisions over all UAVs
  void tickSharded(double dt, int n_ticks, bool enabled, bool crash, double rebounce) {
    mrs_throw_on_error(mrs_swarm_tick_sharded_n(s_, dt, n_ticks, enabled, crash, rebounce));
  }
  // the same over an in-process group (several UavSwarm objects of one process, one host thread each: multi-device hosts without
  // RCCL, virtual shards on one device) or over a caller-supplied all-gather (MPI, ...)
  void commInitLoopback(mrs_loopback_group_t* group, int rank, int64_t n_total) { mrs_throw_on_error(mrs_swarm_comm_init_loopback(s_, group, rank, n_total)); }
  void commInitCustom(int world, int rank, int64_t n_total, mrs_allgather_fn fn, void* user) {
    mrs_throw_on_error(mrs_swarm_comm_init_custom(s_, world, rank, n_total, fn, user));
  }
  void commDestroy() { mrs_throw_on_error(mrs_swarm_comm_destroy(s_)); }
  // MRS_EXCHANGE_EXPORT_SETS (default: boundary UAVs only between two neighbour searches) or MRS_EXCHANGE_FULL_GATHER
  void setExchange(int exchange) { mrs_throw_on_error(mrs_swarm_set_exchange(s_, exchange)); }
  mrs_comm_info_t commInfo() {
    mrs_comm_info_t ci;
    mrs_throw_on_error(mrs_swarm_comm_info(s_, &ci));
    return ci;
  }
  // spatially coherent shards for a swarm addressed by public index: order[k] = public index at position k of the x-sorted order;
  // rank r of `world` holds order[lo_r, hi_r) with equal-count ranges (the first n % world ranks one more)
  static std::vector<int64_t> slabPartition(const std::vector<Eigen::Vector3d>& pos, int world) {
    std::vector<double> p(pos.size() * 3);
    for (size_t k = 0; k < pos.size(); k++)
      for (int j = 0; j < 3; j++) p[k * 3 + (size_t)j] = pos[k](j);
    std::vector<int64_t> order(pos.size());
    mrs_throw_on_error(mrs_slab_partition(p.data(), (int64_t)pos.size(), world, order.data()));
    return order;
  }
  void synchronize() { mrs_throw_on_error(mrs_swarm_synchronize(s_)); }
  // collision ticks so far, and how many of them had to repeat the neighbour search
  std::pair<int64_t, int64_t> collisionStats() {
    int64_t t = 0, r = 0;
    mrs_throw_on_error(mrs_swarm_get_collision_stats(s_, &t, &r));
    return {t, r};
  }

  // ---- UavSystemRos semantics that live on the device ----
  // timeoutInput() for UAVs [first, first+count): safe command of the same mode (src/uav_system_ros.cpp:474-647)
  void timeoutInput(int first, int count) { mrs_throw_on_error(mrs_swarm_timeout_input(s_, first, count)); }
  // UavSystemRos::makeStep iterates the model only `if (_iterate_without_input_ || time_last_input_ > 0)` (:265): UAVs on hold
  // are skipped by makeStep / tick
  void setHold(int first, int count, bool hold) { mrs_throw_on_error(mrs_swarm_set_hold(s_, first, count, hold ? 1 : 0)); }
  // callbackSetMass / callbackSetGroundZ (src/uav_system_ros.cpp:1028-1080)
  void setMass(int first, int count, double mass) { mrs_throw_on_error(mrs_swarm_set_mass(s_, first, count, mass)); }
  void setGroundZ(int first, int count, double ground_z) { mrs_throw_on_error(mrs_swarm_set_ground_z(s_, first, count, ground_z)); }
  // what publishOdometry / publishIMU / publishRangefinder / publishPoses need, one packed download
  std::vector<mrs_uav_output_t> getOutputs(int first, int count) {
    std::vector<mrs_uav_output_t> out((size_t)count);
    mrs_throw_on_error(mrs_swarm_get_outputs(s_, first, count, out.data()));
    return out;
  }
  // the same without the host copy: the pointer stays valid until the next getOutputs* call
  const mrs_uav_output_t* getOutputsView(int first, int count) {
    const mrs_uav_output_t* v = nullptr;
    mrs_throw_on_error(mrs_swarm_get_outputs_view(s_, first, count, &v));
    return v;
  }
  // batched subscriber side: pinned rows to fill with setInput payloads (layout of mrs_swarm_set_input), then one commit
  double* inputStaging(int count, int stride) {
    double* rows = nullptr;
    mrs_throw_on_error(mrs_swarm_input_staging(s_, count, stride, &rows));
    return rows;
  }
  void commitInput(int first, int count, int mode, int stride) { mrs_throw_on_error(mrs_swarm_commit_input(s_, first, count, mode, stride)); }
  // the tail of the UavSystemRos constructor (:223-232) for the whole swarm: zero actuators, two makeStep(0.01)
  void warmUp() {
    std::vector<double> zeros((size_t)size() * MRS_MAX_MOTORS, 0.0);
    mrs_throw_on_error(mrs_swarm_set_input(s_, 0, size(), MRS_ACTUATOR_CMD, zeros.data(), MRS_MAX_MOTORS));
    mrs_throw_on_error(mrs_swarm_step_n(s_, 0.01, 2, 1));
  }

  // getPose() of every UAV (src/uav_system_ros.cpp:289), n x 3 row-major
  std::vector<double> getPoses() {
    std::vector<double> x((size_t)size() * 3);
    mrs_throw_on_error(mrs_swarm_get_state(s_, 0, size(), x.data(), nullptr, nullptr, nullptr, nullptr, nullptr));
    return x;
  }

private:
  mrs_swarm_t* s_;
};

// ------------------------------------------------------------------------------------------------------------------
// the reference's class: a swarm of one
// ------------------------------------------------------------------------------------------------------------------
class UavSystem : public UavSystemApi {
public:
  UavSystem(void) : UavSystemApi(nullptr, 0), own_(std::make_unique<UavSwarm>(1, -1, true)) { s_ = own_->handle(); }  // :127

  UavSystem(const MultirotorModel::ModelParams& model_params) : UavSystem() {  // :135
    const mrs_model_params_t c = model_params.toC();
    mrs_throw_on_error(mrs_swarm_construct(s_, 0, 1, &c, nullptr, nullptr));
  }

  UavSystem(const MultirotorModel::ModelParams& model_params, const Eigen::Vector3d spawn_pos, const double spawn_heading) : UavSystem() {  // :144
    const mrs_model_params_t c    = model_params.toC();
    const double             p[3] = {spawn_pos(0), spawn_pos(1), spawn_pos(2)};
    mrs_throw_on_error(mrs_swarm_construct(s_, 0, 1, &c, p, &spawn_heading));
  }

  // the reference object is a copy-assignable value (uav_system_ = UavSystem(...), src/uav_system_ros.cpp:105): a copy is an
  // independent UAV with the same state, command, feed-forwards, PIDs and parameters (mrs_swarm_clone); moves hand the device state over
  UavSystem(const UavSystem& o) : UavSystemApi(nullptr, 0) {
    mrs_swarm_t* c = nullptr;
    mrs_throw_on_error(mrs_swarm_clone(o.s_, &c));
    own_ = std::make_unique<UavSwarm>(c);
    s_   = c;
  }
  UavSystem& operator=(const UavSystem& o) {
    if (this != &o) {
      mrs_swarm_t* c = nullptr;
      mrs_throw_on_error(mrs_swarm_clone(o.s_, &c));
      own_ = std::make_unique<UavSwarm>(c);
      s_   = c;
    }
    return *this;
  }
  UavSystem(UavSystem&& o) noexcept : UavSystemApi(o.s_, 0), own_(std::move(o.own_)) { o.s_ = nullptr; }
  UavSystem& operator=(UavSystem&& o) noexcept {
    own_ = std::move(o.own_);
    s_   = o.s_;
    o.s_ = nullptr;
    return *this;
  }

private:
  std::unique_ptr<UavSwarm> own_;
};

}  // namespace mrs_multirotor_simulator

#endif  // UAV_SYSTEM_H
