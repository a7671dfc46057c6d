// AttitudeController of the reference (controllers/attitude_controller.hpp:11-46): Params and both getControlSignal overloads.
#ifndef MRS_ATTITUDE_CONTROLLER_HPP
#define MRS_ATTITUDE_CONTROLLER_HPP
#include "controller_probe.hpp"
namespace mrs_multirotor_simulator
{
class AttitudeController {
public:
  class Params {
  public:
    double kp                  = 6.0;
    double kd                  = 0.05;
    double ki                  = 0.01;
    double max_rate_roll_pitch = 10.0;  // [rad/s]
    double max_rate_yaw        = 1.0;   // [rad/s]
  };

  AttitudeController() {}
  AttitudeController(const MultirotorModel::ModelParams& model_params) : probe_(model_params) {}  // :54-58

  void setParams(const Params& params) {  // :64-69
    const mrs_attitude_params_t c{params.kp, params.kd, params.ki, params.max_rate_roll_pitch, params.max_rate_yaw};
    mrs_throw_on_error(mrs_swarm_set_attitude_params(probe_.handle(), 0, 1, &c));
  }

  reference::AttitudeRate getControlSignal(const MultirotorModel::State& state, const reference::Attitude& reference, const double& dt) {  // :79-100
    probe_.setState(state);
    double in[10], out[4];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) in[r * 3 + c] = reference.orientation(r, c);
    in[9] = reference.throttle;
    probe_.run(MRS_COMP_ATTITUDE, in, 10, out, 4, dt);
    return pack(out);
  }

  reference::AttitudeRate getControlSignal(const MultirotorModel::State& state, const reference::TiltHdgRate& reference, const double& dt) {  // :106-145
    probe_.setState(state);
    const double in[5] = {reference.tilt_vector(0), reference.tilt_vector(1), reference.tilt_vector(2), reference.heading_rate, reference.throttle};
    double       out[4];
    probe_.run(MRS_COMP_TILT_HDG_RATE, in, 5, out, 4, dt);
    return pack(out);
  }

  // the three std::cout warnings of :196,236,245 are counted instead of printed
  mrs_diag_t getDiagnostics() {
    mrs_diag_t d;
    mrs_throw_on_error(mrs_swarm_get_diag(probe_.handle(), &d));
    return d;
  }

private:
  static reference::AttitudeRate pack(const double out[4]) {
    reference::AttitudeRate r;
    r.rate_x = out[0]; r.rate_y = out[1]; r.rate_z = out[2]; r.throttle = out[3];
    return r;
  }
  detail::ControllerProbe probe_;
};
}  // namespace mrs_multirotor_simulator
#endif
