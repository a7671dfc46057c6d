// references.hpp — the ten command value types of the reference (same names, same fields, same defaults) so that
// code written against include/mrs_multirotor_simulator/uav_system/controllers/references.hpp:15-271 compiles unchanged.
#ifndef MRS_REFERENCES_HPP
#define MRS_REFERENCES_HPP

#include "../eigen_compat.hpp"

namespace mrs_multirotor_simulator
{
namespace reference
{

class Actuators {  // references.hpp:15-30
public:
  Eigen::VectorXd motors;  // motor throttles scaled as [0, 1]
};

class ControlGroup {  // :34-62
public:
  double roll = 0, pitch = 0, yaw = 0, throttle = 0;
};

class AttitudeRate {  // :66-94
public:
  double rate_x = 0, rate_y = 0, rate_z = 0, throttle = 0;
};

class Attitude {  // :98-118
public:
  Attitude() { this->orientation = Eigen::Matrix3d::Identity(); }
  Eigen::Matrix3d orientation;
  double          throttle = 0;
};

class TiltHdgRate {  // :120-143 — the default tilt is Vector3d::Identity() == (1,0,0), kept
public:
  TiltHdgRate() { this->tilt_vector = Eigen::Vector3d::Identity(); }
  Eigen::Vector3d tilt_vector;
  double          heading_rate = 0;
  double          throttle     = 0;
};

class AccelerationHdgRate {  // :147-169
public:
  AccelerationHdgRate(const Eigen::Vector3d& acceleration_in, const double& heading_rate_in) : acceleration(acceleration_in), heading_rate(heading_rate_in) {}
  AccelerationHdgRate() { this->acceleration = Eigen::Vector3d::Zero(); }
  Eigen::Vector3d acceleration;
  double          heading_rate = 0;
};

class AccelerationHdg {  // :173-198
public:
  AccelerationHdg(const Eigen::Vector3d& acceleration_in, const double& heading_in) : acceleration(acceleration_in), heading(heading_in) {}
  AccelerationHdg() { this->acceleration = Eigen::Vector3d::Zero(); }
  Eigen::Vector3d acceleration;
  double          heading = 0;
};

class VelocityHdgRate {  // :202-227
public:
  VelocityHdgRate(const Eigen::Vector3d& velocity_in, const double& heading_rate_in) : velocity(velocity_in), heading_rate(heading_rate_in) {}
  VelocityHdgRate() { this->velocity = Eigen::Vector3d::Zero(); }
  Eigen::Vector3d velocity;
  double          heading_rate = 0;
};

class VelocityHdg {  // :231-256
public:
  VelocityHdg(const Eigen::Vector3d& velocity_in, const double& heading_in) : velocity(velocity_in), heading(heading_in) {}
  VelocityHdg() { this->velocity = Eigen::Vector3d::Zero(); }
  Eigen::Vector3d velocity;
  double          heading = 0;
};

class Position {  // :260-271
public:
  Position() { this->position = Eigen::Vector3d::Zero(); }
  Eigen::Vector3d position;
  double          heading = 0;
};

}  // namespace reference
}  // namespace mrs_multirotor_simulator
#endif
