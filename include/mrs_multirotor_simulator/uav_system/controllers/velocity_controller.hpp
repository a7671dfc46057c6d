// VelocityController::Params of the reference (controllers/velocity_controller.hpp:14-20).
#ifndef MRS_VELOCITY_CONTROLLER_HPP
#define MRS_VELOCITY_CONTROLLER_HPP
#include "../multirotor_model.hpp"
namespace mrs_multirotor_simulator
{
class VelocityController {
public:
  struct Params
  {
    double kp               = 2.0;
    double kd               = 0.05;
    double ki               = 0.01;
    double max_acceleration = 4.0;  // m/s^2
  };
};
}  // namespace mrs_multirotor_simulator
#endif
