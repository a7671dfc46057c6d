// multirotor_model.hpp — value types of the reference's MultirotorModel (ModelParams, State) with the same field names
// (include/mrs_multirotor_simulator/uav_system/multirotor_model.hpp:24-98).  The dynamics themselves run on the GPU
// behind include/mrs_swarm.h; this header only converts to and from the C ABI's plain structs.
#ifndef MRS_MULTIROTOR_MODEL_HPP
#define MRS_MULTIROTOR_MODEL_HPP

#include <stdexcept>
#include <string>

#include "../../mrs_swarm.h"
#include "controllers/references.hpp"

#define N_INTERNAL_STATES 18

namespace mrs_multirotor_simulator
{

inline void mrs_throw_on_error(int rc) {
  if (rc != MRS_OK) throw std::runtime_error(std::string("libmrs_swarm: ") + mrs_last_error());
}

class MultirotorModel {
public:
  class ModelParams {
  public:
    ModelParams() {  // x500 defaults, multirotor_model.hpp:26-66 (computed by the library with the reference's arithmetic)
      mrs_model_params_t c;
      mrs_model_params_default(&c);
      fromC(c);
    }

    int    n_motors;
    double g, mass, kf, km, prop_radius, arm_length, body_height, motor_time_constant, max_rpm, min_rpm, air_resistance_coeff;

    Eigen::Matrix3d J;
    Eigen::MatrixXd allocation_matrix;  // 4 x n_motors

    bool   ground_enabled;
    double ground_z;  // uninitialised in the reference's ctor; 0 here
    bool   takeoff_patch_enabled;

    mrs_model_params_t toC() const {
      mrs_model_params_t c{};
      c.n_motors = n_motors;
      c.ground_enabled = ground_enabled ? 1 : 0;
      c.takeoff_patch_enabled = takeoff_patch_enabled ? 1 : 0;
      c.g = g; c.mass = mass; c.kf = kf; c.km = km; c.prop_radius = prop_radius; c.arm_length = arm_length;
      c.body_height = body_height; c.motor_time_constant = motor_time_constant; c.max_rpm = max_rpm; c.min_rpm = min_rpm;
      c.air_resistance_coeff = air_resistance_coeff; c.ground_z = ground_z;
      for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) c.J[r * 3 + q] = J(r, q);
      for (int r = 0; r < 4; r++)
        for (int m = 0; m < n_motors && m < MRS_MAX_MOTORS; m++) c.allocation_matrix[r * MRS_MAX_MOTORS + m] = allocation_matrix(r, m);
      return c;
    }
    void fromC(const mrs_model_params_t& c) {
      n_motors = c.n_motors;
      ground_enabled = c.ground_enabled != 0;
      takeoff_patch_enabled = c.takeoff_patch_enabled != 0;
      g = c.g; mass = c.mass; kf = c.kf; km = c.km; prop_radius = c.prop_radius; arm_length = c.arm_length;
      body_height = c.body_height; motor_time_constant = c.motor_time_constant; max_rpm = c.max_rpm; min_rpm = c.min_rpm;
      air_resistance_coeff = c.air_resistance_coeff; ground_z = c.ground_z;
      J = Eigen::Matrix3d::Zero();
      for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) J(r, q) = c.J[r * 3 + q];
      allocation_matrix = Eigen::MatrixXd::Zero(4, n_motors);
      for (int r = 0; r < 4; r++)
        for (int m = 0; m < n_motors; m++) allocation_matrix(r, m) = c.allocation_matrix[r * MRS_MAX_MOTORS + m];
    }
  };

  struct State  // multirotor_model.hpp:90-98
  {
    Eigen::Vector3d x;
    Eigen::Vector3d v;
    Eigen::Vector3d v_prev;
    Eigen::Matrix3d R;
    Eigen::Vector3d omega;
    Eigen::VectorXd motor_rpm;
  };
};

}  // namespace mrs_multirotor_simulator
#endif
