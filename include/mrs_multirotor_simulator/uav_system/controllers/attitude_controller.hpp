// AttitudeController::Params of the reference (controllers/attitude_controller.hpp:14-21).
#ifndef MRS_ATTITUDE_CONTROLLER_HPP
#define MRS_ATTITUDE_CONTROLLER_HPP
#include "../multirotor_model.hpp"
namespace mrs_multirotor_simulator
{
class AttitudeController {
public:
  class Params {
  public:
    double kp                  = 6.0;
    double kd                  = 0.05;
    double ki                  = 0.01;
    double max_rate_roll_pitch = 10.0;  // rad/s
    double max_rate_yaw        = 1.0;   // rad/s
  };
};
}  // namespace mrs_multirotor_simulator
#endif
