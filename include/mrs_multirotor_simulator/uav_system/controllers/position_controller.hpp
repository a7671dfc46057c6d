// PositionController of the reference (controllers/position_controller.hpp:11-36): Params and getControlSignal.
#ifndef MRS_POSITION_CONTROLLER_HPP
#define MRS_POSITION_CONTROLLER_HPP
#include "controller_probe.hpp"
namespace mrs_multirotor_simulator
{
class PositionController {
public:
  class Params {
  public:
    double kp           = 2.0;
    double kd           = 0.15;
    double ki           = 0.2;
    double max_velocity = 6.0;  // m/s
  };

  PositionController() {}
  PositionController(const MultirotorModel::ModelParams& model_params) : probe_(model_params) {}  // :51-56

  void setParams(const Params& params) {  // :62-67
    const mrs_position_params_t c{params.kp, params.kd, params.ki, params.max_velocity};
    mrs_throw_on_error(mrs_swarm_set_position_params(probe_.handle(), 0, 1, &c));
  }

  reference::VelocityHdg getControlSignal(const MultirotorModel::State& state, const reference::Position& reference, const double& dt) {  // :73-86
    probe_.setState(state);
    const double in[3] = {reference.position(0), reference.position(1), reference.position(2)};
    double       out[3];
    probe_.run(MRS_COMP_POSITION, in, 3, out, 3, dt);
    return reference::VelocityHdg(Eigen::Vector3d(out[0], out[1], out[2]), reference.heading);
  }

private:
  detail::ControllerProbe probe_;
};
}  // namespace mrs_multirotor_simulator
#endif
