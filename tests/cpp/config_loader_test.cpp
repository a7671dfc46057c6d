// config_loader_test.cpp — include/mrs_multirotor_simulator/config_loader.hpp: parses the files given on the command line (later
// ones override earlier ones) and prints what the loader derives; with --run it also builds the swarm on the GPU and steps it.
#include <cstdio>
#include <cstring>
#include <mrs_multirotor_simulator/config_loader.hpp>

using namespace mrs_multirotor_simulator;

int main(int argc, char** argv) {
  ParamTree cfg;
  bool      run = false;
  for (int i = 1; i < argc; i++) {
    if (!std::strcmp(argv[i], "--run"))
      run = true;
    else
      cfg.loadFile(argv[i]);
  }
  const SimulatorConfig sc = simulatorConfigFromTree(cfg);
  std::printf("SIM %.17g %.17g %.17g %d %d %.17g %d %.17g\n", sc.simulation_rate, sc.clock_rate, sc.realtime_factor, (int)sc.collisions_enabled,
              (int)sc.collisions_crash, sc.collisions_rebounce, (int)sc.iterate_without_input, sc.input_timeout);
  const std::vector<UavSpawn> uavs = uavSpawnsFromConfig(cfg);
  for (auto& u : uavs) {
    std::printf("UAV %s %s %.17g %.17g %.17g %.17g\n", u.name.c_str(), u.type.c_str(), u.x, u.y, u.z, u.heading);
    const MultirotorModel::ModelParams p = modelParamsFromConfig(cfg, u.type);
    std::printf("PARAMS %s %d %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %d %.17g %d", u.type.c_str(), p.n_motors, p.g, p.mass,
                p.kf, p.km, p.prop_radius, p.arm_length, p.body_height, p.motor_time_constant, p.max_rpm, p.min_rpm, p.air_resistance_coeff,
                (int)p.ground_enabled, p.ground_z, (int)p.takeoff_patch_enabled);
    for (int r = 0; r < 3; r++) std::printf(" %.17g", p.J(r, r));
    for (int r = 0; r < 4; r++)
      for (int m = 0; m < p.n_motors; m++) std::printf(" %.17g", p.allocation_matrix(r, m));
    std::printf("\n");
  }
  if (run) {
    UavSwarm swarm((int)uavs.size(), -1, /*fast_arithmetic=*/false);
    constructSwarmFromConfig(swarm, cfg, uavs);
    for (int i = 0; i < (int)uavs.size(); i++) {
      reference::Position c;
      c.position = Eigen::Vector3d(uavs[(size_t)i].x + 1.0, uavs[(size_t)i].y - 1.0, uavs[(size_t)i].z + 2.0);
      c.heading  = 0.3;
      swarm[i].setInput(c);
    }
    swarm.makeSteps(0.001, 300);
    const std::vector<double> x = swarm.getPoses();
    for (size_t i = 0; i < uavs.size(); i++) std::printf("POSE %zu %.17g %.17g %.17g\n", i, x[i * 3], x[i * 3 + 1], x[i * 3 + 2]);
  }
  return 0;
}
