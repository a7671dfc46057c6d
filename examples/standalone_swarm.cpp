// standalone_swarm.cpp — the reference's standalone simulator (tmux/standalone_*: one nodelet, N UAVs, position goals) without ROS:
// parameter files in the layout of the reference's config/ directory -> one GPU swarm -> the MultirotorSimulator loop.
//
//   g++ -std=c++17 -O2 -I include examples/standalone_swarm.cpp -o standalone_swarm -L mrs_multirotor_simulator_amd -lmrs_swarm -lpthread
//       (add -DMRS_NO_EIGEN where Eigen3 is not installed, and an rpath or LD_LIBRARY_PATH to the directory of libmrs_swarm.so)
//   ./standalone_swarm <seconds of wall time> config/multirotor_simulator.yaml config/uavs.yaml config/uavs/x500.yaml config/controllers/*.yaml [custom.yaml ...]
//
// Every UAV is sent to a goal 5 m above and beside its spawn point (what tmux/standalone_400_uavs' goto.py does with random goals),
// the loop runs paced by simulation_rate x realtime_factor, and once per wall second a status line is printed.
#include <cstdio>
#include <cstdlib>
#include <mrs_multirotor_simulator/config_loader.hpp>

using namespace mrs_multirotor_simulator;

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s <wall seconds> <yaml> [<yaml> ...]\n", argv[0]);
    return 2;
  }
  try {
    const double wall_seconds = std::atof(argv[1]);
    ParamTree    cfg;
    for (int i = 2; i < argc; i++) cfg.loadFile(argv[i]);
    const std::vector<UavSpawn> uavs = uavSpawnsFromConfig(cfg);
    UavSwarm                    swarm((int)uavs.size());
    constructSwarmFromConfig(swarm, cfg, uavs);
    MultirotorSimulator sim(swarm, (int)uavs.size(), simulatorConfigFromTree(cfg));
    for (int i = 0; i < (int)uavs.size(); i++) {
      reference::Position goal;
      goal.position = Eigen::Vector3d(uavs[(size_t)i].x + 3.0, uavs[(size_t)i].y - 2.0, uavs[(size_t)i].z + 5.0);
      goal.heading  = 1.0;
      swarm[i].setInput(goal);
      sim.inputReceived(i);
    }
    std::printf("%d UAVs, simulation_rate %.0f Hz, realtime_factor %.2f\n", (int)uavs.size(), sim.config().simulation_rate, sim.config().realtime_factor);
    // the publishers of the reference (odometry / IMU / range of every UAV after every step, src/uav_system_ros.cpp:278-282; the pose
    // array every tick, src/multirotor_simulator.cpp:215): one packed payload per tick, downloaded while the next tick runs
    long long published = 0;
    double    uav0[3]   = {0, 0, 0}, stamp = 0.0;
    sim.setPublisher([&](double t, const void* payload, int count) {
      const mrs_uav_output_t* out = static_cast<const mrs_uav_output_t*>(payload);
      if (count > 0)
        for (int j = 0; j < 3; j++) uav0[j] = out[0].position[j];
      stamp = t;
      published++;
    });
    double elapsed = 0.0;
    while (elapsed < wall_seconds) {
      const double chunk = std::min(1.0, wall_seconds - elapsed);
      sim.spinFor(chunk);
      elapsed += chunk;
      sim.flushPublisher();  // (the payload of the last tick: nothing stays in flight while this thread prints)
      const auto cs = swarm.collisionStats();
      std::printf("t_sim %8.3f s  rtf %.2f  uav0 at (%.2f, %.2f, %.2f)  ticks %lld, collision ticks %lld (searches %lld)  published %lld payloads, last stamped %.3f s\n",
                  sim.simTime(), sim.actualRtf(), uav0[0], uav0[1], uav0[2], (long long)sim.ticks(), (long long)cs.first, (long long)cs.second, published, stamp);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
