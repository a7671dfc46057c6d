// VelocityController of the reference (controllers/velocity_controller.hpp:11-37): Params and both getControlSignal overloads
// (they share the three PIDs).
#ifndef MRS_VELOCITY_CONTROLLER_HPP
#define MRS_VELOCITY_CONTROLLER_HPP
#include "controller_probe.hpp"
namespace mrs_multirotor_simulator
{
class VelocityController {
public:
  class Params {
  public:
    double kp               = 2.0;
    double kd               = 0.05;
    double ki               = 0.01;
    double max_acceleration = 4.0;  // m/s^2
  };

  VelocityController() {}
  VelocityController(const MultirotorModel::ModelParams& model_params) : probe_(model_params) {}  // :46-51

  void setParams(const Params& params) {  // :57-62
    const mrs_velocity_params_t c{params.kp, params.kd, params.ki, params.max_acceleration};
    mrs_throw_on_error(mrs_swarm_set_velocity_params(probe_.handle(), 0, 1, &c));
  }

  reference::AccelerationHdgRate getControlSignal(const MultirotorModel::State& state, const reference::VelocityHdgRate& reference, const double& dt) {  // :68-83
    const Eigen::Vector3d a = run(state, reference.velocity, dt);
    return reference::AccelerationHdgRate(a, reference.heading_rate);
  }
  reference::AccelerationHdg getControlSignal(const MultirotorModel::State& state, const reference::VelocityHdg& reference, const double& dt) {  // :87-102
    const Eigen::Vector3d a = run(state, reference.velocity, dt);
    return reference::AccelerationHdg(a, reference.heading);
  }

private:
  Eigen::Vector3d run(const MultirotorModel::State& state, const Eigen::Vector3d& vref, double dt) {
    probe_.setState(state);
    const double in[3] = {vref(0), vref(1), vref(2)};
    double       out[3];
    probe_.run(MRS_COMP_VELOCITY, in, 3, out, 3, dt);
    return Eigen::Vector3d(out[0], out[1], out[2]);
  }
  detail::ControllerProbe probe_;
};
}  // namespace mrs_multirotor_simulator
#endif
