// facade_test.cpp — uses include/mrs_multirotor_simulator/uav_system/uav_system.hpp exactly the way the reference's
// UavSystemRos constructor and timer loop use the original header (src/uav_system_ros.cpp:96-157,223-232;
// src/multirotor_simulator.cpp:211-217).  Prints states that tests/test_facade_cpp.py compares with the oracle.
#include <cstdio>
#include <mrs_multirotor_simulator/uav_system/uav_system.hpp>

using namespace mrs_multirotor_simulator;

static void print_state(const char* tag, MultirotorModel::State st, Eigen::Vector3d imu) {
  std::printf("%s", tag);
  for (int i = 0; i < 3; i++) std::printf(" %.17g", st.x(i));
  for (int i = 0; i < 3; i++) std::printf(" %.17g", st.v(i));
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) std::printf(" %.17g", st.R(r, c));
  for (int i = 0; i < 3; i++) std::printf(" %.17g", st.omega(i));
  for (int i = 0; i < st.motor_rpm.size(); i++) std::printf(" %.17g", st.motor_rpm(i));
  for (int i = 0; i < 3; i++) std::printf(" %.17g", imu(i));
  std::printf("\n");
}

int main() {
  // --- one UavSystem, BASELINE config 1 ---
  MultirotorModel::ModelParams model_params;  // x500 defaults
  model_params.ground_enabled        = true;
  model_params.ground_z              = 0.0;
  model_params.takeoff_patch_enabled = false;
  UavSystem uav_system;
  uav_system = UavSystem(model_params, Eigen::Vector3d(10, 15, 0), 3.14);
  uav_system.setMixerParams(Mixer::Params());
  uav_system.setRateControllerParams(RateController::Params());
  uav_system.setAttitudeControllerParams(AttitudeController::Params());
  uav_system.setVelocityControllerParams(VelocityController::Params());
  uav_system.setPositionControllerParams(PositionController::Params());
  reference::Actuators actuators_cmd;
  actuators_cmd.motors = Eigen::VectorXd::Zero(model_params.n_motors);
  uav_system.setInput(actuators_cmd);
  uav_system.makeStep(0.01);
  uav_system.makeStep(0.01);
  print_state("WARMUP", uav_system.getState(), uav_system.getImuAcceleration());
  reference::Position cmd;
  cmd.position = Eigen::Vector3d(12, 13, 5);
  cmd.heading  = 1.0;
  uav_system.setInput(cmd);
  for (int k = 0; k < 1000; k++) uav_system.makeStep(0.001);
  print_state("STEP1000", uav_system.getState(), uav_system.getImuAcceleration());
  uav_system.setFeedforward(reference::VelocityHdg(Eigen::Vector3d(0.5, -0.25, 0.1), 0));
  uav_system.applyForce(Eigen::Vector3d(1.0, 2.0, -0.5));
  for (int k = 0; k < 100; k++) uav_system.makeStep(0.001);
  print_state("STEP1100", uav_system.getState(), uav_system.getImuAcceleration());
  Eigen::MatrixXd alloc = uav_system.getMixerAllocation();
  std::printf("ALLOC %d %d %.17g %.17g\n", (int)alloc.rows(), (int)alloc.cols(), alloc(0, 0), alloc(3, 2));
  uav_system.crash();
  std::printf("CRASHED %d mass %.17g\n", (int)uav_system.hasCrashed(), uav_system.getParams().mass);
  {  // UavSystemRos::getPose: the position of getState()
    const Eigen::Vector3d pose = uav_system.getPose();
    const auto            st   = uav_system.getState();
    std::printf("POSE %d\n", (int)(pose(0) == st.x(0) && pose(1) == st.x(1) && pose(2) == st.x(2)));
  }

  // --- a swarm: the simulator loop with one launch per tick ---
  const int n = 400;
  UavSwarm  swarm(n, -1, /*fast_arithmetic=*/false);
  std::vector<Eigen::Vector3d> pos;
  std::vector<double>          hdg;
  for (int i = 0; i < n; i++) {
    pos.push_back(Eigen::Vector3d(4.0 * (i / 20), 4.0 * (i % 20), 0.0));
    hdg.push_back(0.0);
  }
  swarm.construct(0, n, model_params, pos, hdg);
  for (int i = 0; i < n; i++) {
    reference::Position c;
    c.position = Eigen::Vector3d(pos[i](0) + 1.0, pos[i](1) - 2.0, 3.0 + 0.01 * i);
    c.heading  = 0.001 * i;
    swarm[i].setInput(c);
  }
  for (int k = 0; k < 200; k++) {
    swarm.makeStep(0.001);
    swarm.handleCollisions(true, false, 100.0);
  }
  swarm.timeoutInput(0, 200);
  swarm.setMass(100, 50, 2.4);
  for (int k = 0; k < 50; k++) swarm.makeStep(0.001);
  std::vector<mrs_uav_output_t> outs = swarm.getOutputs(0, n);
  std::printf("OUT7 %.17g %.17g %.17g %.17g %.17g\n", outs[7].orientation[3], outs[7].velocity_body[0], outs[7].range, outs[120].position[2],
              outs[120].linear_acceleration[2]);
  print_state("SWARM7", swarm[7].getState(), swarm[7].getImuAcceleration());
  print_state("SWARM399", swarm[399].getState(), swarm[399].getImuAcceleration());
  return 0;
}
