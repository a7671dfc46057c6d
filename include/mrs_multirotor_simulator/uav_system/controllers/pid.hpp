// PIDController of the reference (controllers/pid.hpp:9-36): same methods, same semantics; update() runs the PID device function of
// the cascade kernels on the GPU (mrs_debug_pid_update, reference operation order) with the state kept in this object.
#ifndef MRS_PID_HPP
#define MRS_PID_HPP
#include "../multirotor_model.hpp"
namespace mrs_multirotor_simulator
{
class PIDController {
private:
  double params_[5] = {0, 0, 0, -1, -1};  // kp, kd, ki, saturation, antiwindup (pid.hpp:15-19: only saturation and antiwindup have defaults)
  double state_[2]  = {0, 0};             // last_error, integral (:20-21)

public:
  PIDController() { reset(); }  // :42-45

  void setParams(const double& kp, const double& kd, const double& ki, const double& saturation, const double& antiwindup) {  // :47-54
    params_[0] = kp; params_[1] = kd; params_[2] = ki; params_[3] = saturation; params_[4] = antiwindup;
  }
  void setSaturation(const double saturation = -1) { params_[3] = saturation; }  // :56-59
  void reset(void) { state_[0] = state_[1] = 0.0; }                              // :61-65

  double update(const double& error, const double& dt) {  // :67-96
    double out = 0.0;
    mrs_throw_on_error(mrs_debug_pid_update(-1, MRS_ARITH_LITERAL, 1, params_, state_, &error, &dt, &out));
    return out;
  }
};
}  // namespace mrs_multirotor_simulator
#endif
