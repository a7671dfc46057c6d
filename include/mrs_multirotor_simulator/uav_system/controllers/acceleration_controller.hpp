// AccelerationController has no parameters in the reference (controllers/acceleration_controller.hpp:11-22).
#ifndef MRS_ACCELERATION_CONTROLLER_HPP
#define MRS_ACCELERATION_CONTROLLER_HPP
#include "../multirotor_model.hpp"
namespace mrs_multirotor_simulator
{
class AccelerationController {};
}  // namespace mrs_multirotor_simulator
#endif
