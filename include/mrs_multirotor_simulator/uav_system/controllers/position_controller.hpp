// PositionController::Params of the reference (controllers/position_controller.hpp:14-20).
#ifndef MRS_POSITION_CONTROLLER_HPP
#define MRS_POSITION_CONTROLLER_HPP
#include "../multirotor_model.hpp"
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
};
}  // namespace mrs_multirotor_simulator
#endif
