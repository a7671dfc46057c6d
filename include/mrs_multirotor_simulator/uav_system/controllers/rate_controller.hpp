// RateController::Params of the reference (controllers/rate_controller.hpp:14-19).
#ifndef MRS_RATE_CONTROLLER_HPP
#define MRS_RATE_CONTROLLER_HPP
#include "../multirotor_model.hpp"
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
};
}  // namespace mrs_multirotor_simulator
#endif
