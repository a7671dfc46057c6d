// AccelerationController of the reference (controllers/acceleration_controller.hpp:10-24): no parameters, two getControlSignal overloads.
#ifndef MRS_ACCELERATION_CONTROLLER_HPP
#define MRS_ACCELERATION_CONTROLLER_HPP
#include "controller_probe.hpp"
namespace mrs_multirotor_simulator
{
class AccelerationController {
public:
  AccelerationController() {}
  AccelerationController(const MultirotorModel::ModelParams& model_params) : probe_(model_params) {}  // :33-36

  reference::TiltHdgRate getControlSignal(const MultirotorModel::State& state, const reference::AccelerationHdgRate& reference, const double dt) {  // :103-122
    probe_.setState(state);
    const double in[4] = {reference.acceleration(0), reference.acceleration(1), reference.acceleration(2), reference.heading_rate};
    double       out[5];
    probe_.run(MRS_COMP_ACCELERATION_HDG_RATE, in, 4, out, 5, dt > 0 ? dt : 0.001);
    reference::TiltHdgRate r;
    r.tilt_vector  = Eigen::Vector3d(out[0], out[1], out[2]);
    r.heading_rate = out[3];
    r.throttle     = out[4];
    return r;
  }

  reference::Attitude getControlSignal(const MultirotorModel::State& state, const reference::AccelerationHdg& reference, const double dt) {  // :44-97
    probe_.setState(state);
    const double in[4] = {reference.acceleration(0), reference.acceleration(1), reference.acceleration(2), reference.heading};
    double       out[10];
    probe_.run(MRS_COMP_ACCELERATION_HDG, in, 4, out, 10, dt > 0 ? dt : 0.001);
    reference::Attitude a;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) a.orientation(r, c) = out[r * 3 + c];
    a.throttle = out[9];
    return a;
  }

private:
  detail::ControllerProbe probe_;
};
}  // namespace mrs_multirotor_simulator
#endif
