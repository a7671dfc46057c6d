// RateController of the reference (controllers/rate_controller.hpp:11-35): Params and getControlSignal; the three PIDs live on the GPU.
#ifndef MRS_RATE_CONTROLLER_HPP
#define MRS_RATE_CONTROLLER_HPP
#include "controller_probe.hpp"
namespace mrs_multirotor_simulator
{
class RateController {
public:
  class Params {
  public:
    double kp = 4.0;
    double kd = 0.04;
    double ki = 0.0;
  };

  RateController() {}
  RateController(const MultirotorModel::ModelParams& model_params) : probe_(model_params) {}  // :43-47

  void setParams(const Params& params) {  // :49-54 (re-initialises the PIDs)
    const mrs_rate_params_t c{params.kp, params.kd, params.ki};
    mrs_throw_on_error(mrs_swarm_set_rate_params(probe_.handle(), 0, 1, &c));
  }

  reference::ControlGroup getControlSignal(const MultirotorModel::State& state, const reference::AttitudeRate& reference, const double& dt) {  // :67-81
    probe_.setState(state);
    const double in[4] = {reference.rate_x, reference.rate_y, reference.rate_z, reference.throttle};
    double       out[4];
    probe_.run(MRS_COMP_RATE, in, 4, out, 4, dt);
    reference::ControlGroup cg;
    cg.roll = out[0]; cg.pitch = out[1]; cg.yaw = out[2]; cg.throttle = out[3];
    return cg;
  }

private:
  detail::ControllerProbe probe_;
};
}  // namespace mrs_multirotor_simulator
#endif
