// multirotor_model.hpp — the reference's MultirotorModel: value types (ModelParams, State) with the same field names
// (include/mrs_multirotor_simulator/uav_system/multirotor_model.hpp:24-98) and the stand-alone model object with its public
// methods (:100-131).  The dynamics run on the GPU behind include/mrs_swarm.h — a MultirotorModel object is a swarm of one UAV
// in ACTUATOR mode (slow per call: porting and tests; the hot loop belongs to UavSwarm) — this header only marshals values.
#ifndef MRS_MULTIROTOR_MODEL_HPP
#define MRS_MULTIROTOR_MODEL_HPP

#include <array>
#include <stdexcept>
#include <string>

#include "../../mrs_swarm.h"
#include "controllers/references.hpp"

#define N_INTERNAL_STATES 18

namespace mrs_multirotor_simulator
{

inline void mrs_throw_on_error(int rc) {
  if (rc != MRS_OK) throw std::runtime_error(std::string("libmrs_swarm: ") + mrs_last_error());
}

class MultirotorModel {
public:
  class ModelParams {
  public:
    ModelParams() {  // x500 defaults, multirotor_model.hpp:26-66 (computed by the library with the reference's arithmetic)
      mrs_model_params_t c;
      mrs_model_params_default(&c);
      fromC(c);
    }

    int    n_motors;
    double g, mass, kf, km, prop_radius, arm_length, body_height, motor_time_constant, max_rpm, min_rpm, air_resistance_coeff;

    Eigen::Matrix3d J;
    Eigen::MatrixXd allocation_matrix;  // 4 x n_motors

    bool   ground_enabled;
    double ground_z;  // uninitialised in the reference's ctor; 0 here
    bool   takeoff_patch_enabled;

    mrs_model_params_t toC() const {
      mrs_model_params_t c{};
      c.n_motors = n_motors;
      c.ground_enabled = ground_enabled ? 1 : 0;
      c.takeoff_patch_enabled = takeoff_patch_enabled ? 1 : 0;
      c.g = g; c.mass = mass; c.kf = kf; c.km = km; c.prop_radius = prop_radius; c.arm_length = arm_length;
      c.body_height = body_height; c.motor_time_constant = motor_time_constant; c.max_rpm = max_rpm; c.min_rpm = min_rpm;
      c.air_resistance_coeff = air_resistance_coeff; c.ground_z = ground_z;
      for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) c.J[r * 3 + q] = J(r, q);
      for (int r = 0; r < 4; r++)
        for (int m = 0; m < n_motors && m < MRS_MAX_MOTORS; m++) c.allocation_matrix[r * MRS_MAX_MOTORS + m] = allocation_matrix(r, m);
      return c;
    }
    void fromC(const mrs_model_params_t& c) {
      n_motors = c.n_motors;
      ground_enabled = c.ground_enabled != 0;
      takeoff_patch_enabled = c.takeoff_patch_enabled != 0;
      g = c.g; mass = c.mass; kf = c.kf; km = c.km; prop_radius = c.prop_radius; arm_length = c.arm_length;
      body_height = c.body_height; motor_time_constant = c.motor_time_constant; max_rpm = c.max_rpm; min_rpm = c.min_rpm;
      air_resistance_coeff = c.air_resistance_coeff; ground_z = c.ground_z;
      J = Eigen::Matrix3d::Zero();
      for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) J(r, q) = c.J[r * 3 + q];
      allocation_matrix = Eigen::MatrixXd::Zero(4, n_motors);
      for (int r = 0; r < 4; r++)
        for (int m = 0; m < n_motors; m++) allocation_matrix(r, m) = c.allocation_matrix[r * MRS_MAX_MOTORS + m];
    }
  };

  struct State  // multirotor_model.hpp:90-98
  {
    Eigen::Vector3d x;
    Eigen::Vector3d v;
    Eigen::Vector3d v_prev;
    Eigen::Matrix3d R;
    Eigen::Vector3d omega;
    Eigen::VectorXd motor_rpm;
  };

  // ---- the model object, multirotor_model.hpp:100-131 ----
  MultirotorModel() : s_(nullptr) { create(nullptr, nullptr, nullptr); }  // :158-163
  MultirotorModel(const ModelParams& params, const Eigen::Vector3d& spawn_pos, const double spawn_heading) : s_(nullptr) {  // :165-178
    const mrs_model_params_t c    = params.toC();
    const double             p[3] = {spawn_pos(0), spawn_pos(1), spawn_pos(2)};
    create(&c, p, &spawn_heading);
  }
  ~MultirotorModel() { mrs_swarm_destroy(s_); }
  MultirotorModel(const MultirotorModel& o) : s_(nullptr), external_moment_(o.external_moment_) { mrs_throw_on_error(mrs_swarm_clone(o.s_, &s_)); }
  MultirotorModel& operator=(const MultirotorModel& o) {
    if (this != &o) {
      mrs_swarm_t* c = nullptr;
      mrs_throw_on_error(mrs_swarm_clone(o.s_, &c));
      mrs_swarm_destroy(s_);
      s_               = c;
      external_moment_ = o.external_moment_;
    }
    return *this;
  }

  const MultirotorModel::State& getState(void) const {  // :414 (a reference into the object, refreshed by every call)
    double x[3], v[3], vp[3], R[9], w[3], rpm[MRS_MAX_MOTORS];
    mrs_throw_on_error(mrs_swarm_get_state(s_, 0, 1, x, v, vp, R, w, rpm));
    mrs_model_params_t p;
    mrs_throw_on_error(mrs_swarm_get_params(s_, 0, &p));
    state_.x      = Eigen::Vector3d(x[0], x[1], x[2]);
    state_.v      = Eigen::Vector3d(v[0], v[1], v[2]);
    state_.v_prev = Eigen::Vector3d(vp[0], vp[1], vp[2]);
    state_.omega  = Eigen::Vector3d(w[0], w[1], w[2]);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) state_.R(r, c) = R[r * 3 + c];
    state_.motor_rpm = Eigen::VectorXd::Zero(p.n_motors);
    for (int m = 0; m < p.n_motors; m++) state_.motor_rpm(m) = rpm[m];
    return state_;
  }

  void setState(const MultirotorModel::State& state) {  // :424-433 (v_prev stays)
    double x[3], v[3], R[9], w[3], rpm[MRS_MAX_MOTORS] = {0};
    for (int c = 0; c < 3; c++) {
      x[c] = state.x(c); v[c] = state.v(c); w[c] = state.omega(c);
      for (int q = 0; q < 3; q++) R[c * 3 + q] = state.R(c, q);
    }
    for (int m = 0; m < (int)state.motor_rpm.size() && m < MRS_MAX_MOTORS; m++) rpm[m] = state.motor_rpm(m);
    mrs_throw_on_error(mrs_swarm_set_state(s_, 0, 1, x, v, R, w, rpm));
  }

  void applyForce(const Eigen::Vector3d& force) { setExternalForce(force); }  // :292-295

  void setStatePos(const Eigen::Vector3d& pos, const double heading) {  // :439-446
    const double p[3] = {pos(0), pos(1), pos(2)};
    mrs_throw_on_error(mrs_swarm_set_state_pos(s_, 0, 1, p, &heading));
  }

  const Eigen::Vector3d& getExternalForce(void) const {  // :452
    double f[3];
    mrs_throw_on_error(mrs_swarm_get_external_force(s_, 0, 1, f));
    external_force_ = Eigen::Vector3d(f[0], f[1], f[2]);
    return external_force_;
  }
  void setExternalForce(const Eigen::Vector3d& force) {  // :460
    const double f[3] = {force(0), force(1), force(2)};
    mrs_throw_on_error(mrs_swarm_apply_force(s_, 0, 1, f));
  }
  // the external moment is never set through UavSystem in the reference (always zero on the path); the GPU model carries no term for it
  const Eigen::Vector3d& getExternalMoment(void) const { return external_moment_; }  // :468
  void                   setExternalMoment(const Eigen::Vector3d& moment) {          // :476
    if (moment(0) != 0.0 || moment(1) != 0.0 || moment(2) != 0.0)
      throw std::runtime_error("MultirotorModel::setExternalMoment: a non-zero external moment is not modelled by the GPU stepper");
    external_moment_ = moment;
  }

  void setInput(const reference::Actuators& input) {  // :392-410
    double p[MRS_MAX_MOTORS] = {0};
    for (int m = 0; m < (int)input.motors.size() && m < MRS_MAX_MOTORS; m++) p[m] = input.motors(m);
    mrs_throw_on_error(mrs_swarm_set_input(s_, 0, 1, MRS_ACTUATOR_CMD, p, MRS_MAX_MOTORS));
  }

  void step(const double& dt) { mrs_throw_on_error(mrs_swarm_step(s_, dt)); }  // :220-286

  // the ODE right-hand side in the reference's internal order [x, v, R col0, R col1, R col2, omega] (:204-214, :301-366);
  // boost::array in the reference, std::array here (same interface for indexing)
  typedef std::array<double, N_INTERNAL_STATES> InternalState;
  void operator()(const MultirotorModel::InternalState& x, MultirotorModel::InternalState& dxdt, const double /*t*/) {
    double a[18], o[18];
    for (int i = 0; i < 3; i++) {
      a[i] = x[i]; a[3 + i] = x[3 + i]; a[15 + i] = x[15 + i];
      a[6 + 3 * i + 0] = x[6 + i]; a[6 + 3 * i + 1] = x[9 + i]; a[6 + 3 * i + 2] = x[12 + i];
    }
    mrs_throw_on_error(mrs_swarm_debug_component(s_, MRS_COMP_MODEL_RHS, 0, 1, a, 18, o, 18, 0.001));
    for (int i = 0; i < 3; i++) {
      dxdt[i] = o[i]; dxdt[3 + i] = o[3 + i]; dxdt[15 + i] = o[15 + i];
      dxdt[6 + i] = o[6 + 3 * i + 0]; dxdt[9 + i] = o[6 + 3 * i + 1]; dxdt[12 + i] = o[6 + 3 * i + 2];
    }
  }

  Eigen::Vector3d getImuAcceleration() const {  // :484
    double a[3];
    mrs_throw_on_error(mrs_swarm_get_imu(s_, 0, 1, a));
    return Eigen::Vector3d(a[0], a[1], a[2]);
  }

  ModelParams getParams(void) {  // :384 (takeoff_patch_enabled as step() left it)
    mrs_model_params_t c;
    mrs_throw_on_error(mrs_swarm_get_params(s_, 0, &c));
    ModelParams p;
    p.fromC(c);
    return p;
  }
  void setParams(const ModelParams& params) {  // :374
    const mrs_model_params_t c = params.toC();
    mrs_throw_on_error(mrs_swarm_set_params(s_, 0, 1, &c));
  }

  void initializeState(void) {  // :183-198: everything zero, R = I (parameters stay)
    mrs_model_params_t c;
    mrs_throw_on_error(mrs_swarm_get_params(s_, 0, &c));
    mrs_throw_on_error(mrs_swarm_construct(s_, 0, 1, &c, nullptr, nullptr));
    const double zeros[MRS_MAX_MOTORS] = {0};
    mrs_throw_on_error(mrs_swarm_set_input(s_, 0, 1, MRS_ACTUATOR_CMD, zeros, MRS_MAX_MOTORS));
  }

  mrs_swarm_t* handle() { return s_; }

private:
  void create(const mrs_model_params_t* c, const double* pos, const double* heading) {
    mrs_throw_on_error(mrs_swarm_create(1, -1, &s_));
    try {
      if (c) mrs_throw_on_error(mrs_swarm_construct(s_, 0, 1, c, pos, heading));
      const double zeros[MRS_MAX_MOTORS] = {0};  // input_ = Zero(n_motors), :193
      mrs_throw_on_error(mrs_swarm_set_input(s_, 0, 1, MRS_ACTUATOR_CMD, zeros, MRS_MAX_MOTORS));
    }
    catch (...) {
      mrs_swarm_destroy(s_);
      s_ = nullptr;
      throw;
    }
  }
  mrs_swarm_t*            s_;
  mutable State           state_;
  mutable Eigen::Vector3d external_force_;
  Eigen::Vector3d         external_moment_;
};

}  // namespace mrs_multirotor_simulator
#endif
