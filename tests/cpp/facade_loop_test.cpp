// facade_loop_test.cpp — the reference's simulator loop, UNCHANGED, over 400 stand-alone UavSystem objects:
//     for (size_t i = 0; i < uavs_.size(); i++) uavs_[i]->makeStep(simulation_time_step);      src/multirotor_simulator.cpp:211-213
// with what UavSystemRos::makeStep does around the call — getState() right after it (src/uav_system_ros.cpp:270-282) — and the
// UavSystemRos constructor's sequence per object (controller parameters, zero actuators, two warm-up steps: :109-157, :223-232).
// The objects live in the process-wide UavPool (uav_system.hpp): the first makeStep of a round steps every slot with ONE launch,
// the others consume their result, getState() is served from ONE download.  Disturbances the guess must survive: a setInput and an
// applyForce that arrive between the round's launch and the object's own makeStep, a makeStep with another dt, copies of objects,
// an object that is destroyed and one that is created in the middle of the run.
// Prints the final states (tests/test_facade_cpp.py replays the scenario on the CPU oracle), the pool's counters and the measured
// time per makeStep + getState call; `facade_loop_test single` runs the same with MRS_FACADE_SPECULATE=0 semantics (every object
// stepped on its own: one launch and one synchronisation per call — what the facade cost before the pool).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include <mrs_multirotor_simulator/uav_system/uav_system.hpp>

using namespace mrs_multirotor_simulator;

static void print_state(const char* tag, int i, const MultirotorModel::State& st) {
  std::printf("%s %d", tag, i);
  for (int k = 0; k < 3; k++) std::printf(" %.17g", st.x(k));
  for (int k = 0; k < 3; k++) std::printf(" %.17g", st.v(k));
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) std::printf(" %.17g", st.R(r, c));
  for (int k = 0; k < 3; k++) std::printf(" %.17g", st.omega(k));
  for (int k = 0; k < (int)st.motor_rpm.size(); k++) std::printf(" %.17g", st.motor_rpm(k));
  std::printf("\n");
}

int main(int argc, char** argv) {
  const bool   single = argc > 1 && std::strcmp(argv[1], "single") == 0;
  const int    n      = 400;
  const double dt     = 0.001;
  if (single) setenv("MRS_FACADE_SPECULATE", "0", 1);

  MultirotorModel::ModelParams model_params;  // x500 defaults
  model_params.ground_enabled        = true;
  model_params.ground_z              = 0.0;
  model_params.takeoff_patch_enabled = false;

  std::vector<std::unique_ptr<UavSystem>> uavs_;
  for (int i = 0; i < n; i++) {  // UavSystemRos::UavSystemRos, one object after the other (src/multirotor_simulator.cpp:150-157)
    uavs_.push_back(std::make_unique<UavSystem>(model_params, Eigen::Vector3d(4.0 * (i / 20), 4.0 * (i % 20), 0.0), 0.01 * i));
    UavSystem& u = *uavs_.back();
    u.setMixerParams(Mixer::Params());
    u.setRateControllerParams(RateController::Params());
    u.setAttitudeControllerParams(AttitudeController::Params());
    u.setVelocityControllerParams(VelocityController::Params());
    u.setPositionControllerParams(PositionController::Params());
    reference::Actuators a;
    a.motors = Eigen::VectorXd::Zero(model_params.n_motors);
    u.setInput(a);
    u.makeStep(0.01);
    u.makeStep(0.01);
  }
  for (int i = 0; i < n; i++) {
    reference::Position cmd;
    cmd.position = Eigen::Vector3d(4.0 * (i / 20) + 1.0, 4.0 * (i % 20) - 2.0, 3.0 + 0.01 * i);
    cmd.heading  = 0.001 * i;
    uavs_[(size_t)i]->setInput(cmd);
  }
  const UavPool::Stats s0 = UavPool::instance().stats();

  double    checksum = 0.0;
  const int ticks = single ? 40 : 300, warm = single ? 5 : 20;
  auto      t0 = std::chrono::steady_clock::now();
  UavPool::Stats s_warm = s0;
  for (int tick = 0; tick < ticks; tick++) {
    if (tick == warm) {
      t0     = std::chrono::steady_clock::now();
      s_warm = UavPool::instance().stats();
    }
    for (size_t i = 0; i < uavs_.size(); i++) {
      if (!single && tick == 120 && i == 11) {  // a subscriber callback between two makeStep calls of the loop: UAV 17 has NOT stepped yet
        reference::Position cmd;
        cmd.position = Eigen::Vector3d(0.0, 0.0, 9.0);
        cmd.heading  = 1.0;
        uavs_[17]->setInput(cmd);
        uavs_[300]->applyForce(Eigen::Vector3d(1.0, -2.0, 0.5));
      }
      if (!single && tick == 150 && i == 40) {
        uavs_[40]->makeStep(2.0 * dt);  // one object with another step
      } else {
        uavs_[i]->makeStep(dt);
      }
      const MultirotorModel::State st = uavs_[i]->getState();  // src/uav_system_ros.cpp:270-282
      checksum += st.x(2);
    }
    if (!single && tick == 200) {
      *uavs_[5] = *uavs_[6];                                  // copy assignment (src/uav_system_ros.cpp:105)
      uavs_[7]  = std::make_unique<UavSystem>(*uavs_[8]);     // an object destroyed, a copy-constructed one in its place
    }
  }
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const UavPool::Stats s1 = UavPool::instance().stats();
  const int    timed = ticks - warm;
  std::printf("MODE %s\n", single ? "single" : "pooled");
  std::printf("LATENCY_US_PER_CALL %.3f\n", el / ((double)timed * n) * 1e6);
  std::printf("TICK_US %.1f\n", el / timed * 1e6);
  std::printf("STATS rounds %lld consumed %lld single_steps %lld rollbacks %lld state_hits %lld state_misses %lld grows %lld\n", s1.rounds - s0.rounds,
              s1.consumed - s0.consumed, s1.single_steps - s0.single_steps, s1.rollbacks - s0.rollbacks, s1.state_hits - s0.state_hits,
              s1.state_misses - s0.state_misses, s1.grows);
  std::printf("TIMED rounds %lld single_steps %lld ticks %d\n", s1.rounds - s_warm.rounds, s1.single_steps - s_warm.single_steps, timed);
  std::printf("CHECKSUM %.17g\n", checksum);
  for (int i : {0, 5, 6, 7, 8, 17, 40, 300, 399}) print_state("STATE", i, uavs_[(size_t)i]->getState());
  if (!single) {
    // UavSystem() created AFTER rounds have run: the rounds' launches step every slot of the pool, the free ones too —
    // a late default-constructed object must still start from the reference's zero state (uav_system.hpp:127-133), whether its slot
    // was used before (the first: slot of the object destroyed at tick 200) or never (the second)
    UavSystem late_a, late_b;
    std::printf("LATESLOTS %d %d\n", late_a.poolSlot(), late_b.poolSlot());
    print_state("LATE0", 0, late_a.getState());
    print_state("LATE0", 1, late_b.getState());
    reference::Actuators a;
    a.motors = Eigen::VectorXd::Zero(4);
    for (int m = 0; m < 4; m++) a.motors(m) = 0.55 + 0.01 * m;
    late_a.setInput(a);
    late_b.setInput(a);
    for (int k = 0; k < 3; k++) {
      late_a.makeStep(dt);
      late_b.makeStep(dt);
    }
    print_state("LATE3", 0, late_a.getState());
    print_state("LATE3", 1, late_b.getState());
  }
  return 0;
}
