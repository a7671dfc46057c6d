// simulator_gpu_test.cpp — MultirotorSimulator (include/mrs_multirotor_simulator/multirotor_simulator.hpp) over the real UavSwarm:
// the watchdog / hold logic of UavSystemRos::makeStep (src/uav_system_ros.cpp:243-271) end to end on the GPU.
// Exit code 0 and "ok ..." lines on success.
#include <cmath>
#include <cstdio>
#include <vector>
#include <mrs_multirotor_simulator/multirotor_simulator.hpp>

using namespace mrs_multirotor_simulator;

#define CHECK(c)                                                 \
  do {                                                           \
    if (!(c)) {                                                  \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                  \
    }                                                            \
  } while (0)

static bool same(const std::vector<double>& a, const std::vector<double>& b, int uav) {
  for (int j = 0; j < 3; j++)
    if (a[(size_t)uav * 3 + j] != b[(size_t)uav * 3 + j]) return false;
  return true;
}

int main() {
  const int                    n = 130;  // two full 64-UAV blocks and a tail
  MultirotorModel::ModelParams mp;
  mp.ground_enabled = true;
  mp.ground_z       = 0.0;
  UavSwarm                     swarm(n, -1, /*fast_arithmetic=*/true);
  std::vector<Eigen::Vector3d> pos;
  std::vector<double>          hdg;
  for (int i = 0; i < n; i++) {
    pos.push_back(Eigen::Vector3d(5.0 * (i / 12), 5.0 * (i % 12), 10.0));
    hdg.push_back(0.1 * i);
  }
  swarm.construct(0, n, mp, pos, hdg);
  swarm.warmUp();

  SimulatorConfig cfg;
  cfg.simulation_rate       = 1000.0;
  cfg.clock_rate            = 250.0;
  cfg.iterate_without_input = false;
  cfg.input_timeout         = 0.02;
  cfg.collisions_enabled    = true;
  cfg.collisions_crash      = false;
  MultirotorSimulator sim(swarm, n, cfg);

  // nobody has an input: the swarm is frozen although the loop ticks (and although the UAVs hang in mid-air)
  const std::vector<double> x0 = swarm.getPoses();
  int                       clocks = 0;
  for (int k = 0; k < 40; k++) clocks += sim.timerMain();
  CHECK(clocks == 10);
  const std::vector<double> x1 = swarm.getPoses();
  for (int i = 0; i < n; i++) CHECK(same(x0, x1, i));
  std::printf("ok frozen_without_input\n");

  // two UAVs receive position commands and fly; the rest stays put
  const int fly[2] = {3, 70};
  for (int u : fly) {
    reference::Position c;
    c.position = Eigen::Vector3d(pos[(size_t)u](0) + 2.0, pos[(size_t)u](1), 12.0);
    c.heading  = 0.0;
    swarm[u].setInput(c);
    sim.inputReceived(u);
  }
  for (int k = 0; k < 15; k++) sim.timerMain();
  const std::vector<double> x2 = swarm.getPoses();
  for (int i = 0; i < n; i++) CHECK(same(x1, x2, i) == (i != 3 && i != 70));
  std::printf("ok only_commanded_uavs_move\n");

  // UAV 70 keeps being commanded, UAV 3 falls silent: after input_timeout it gets the hover command of its mode and goes on hold
  for (int k = 0; k < 30; k++) {
    if (k % 5 == 0) {
      reference::Position c;
      c.position = Eigen::Vector3d(pos[70](0) + 2.0, pos[70](1), 12.0);
      swarm[70].setInput(c);
      sim.inputReceived(70);
    }
    sim.timerMain();
  }
  CHECK(!sim.hasInput(3) && sim.hasInput(70));
  const std::vector<double> x3 = swarm.getPoses();
  for (int k = 0; k < 10; k++) {
    if (k % 5 == 0) sim.inputReceived(70);
    sim.timerMain();
  }
  const std::vector<double> x4 = swarm.getPoses();
  CHECK(same(x3, x4, 3) && !same(x3, x4, 70));
  std::printf("ok timeout_puts_on_hold\n");

  // paced run: 1000 Hz at RTF 0.5 for 0.2 wall seconds ~ 100 ticks, RTF telemetry moves towards 0.5
  sim.reconfigure(0.5, false, true, false, 100.0);
  const int64_t t0 = sim.ticks();
  sim.spinFor(1.25);
  const int64_t dticks = sim.ticks() - t0;
  CHECK(dticks > 100 && dticks <= 626);  // upper bound = rate x RTF x time; a loaded host may lose ticks
  CHECK(sim.actualRtf() < 1.0 && sim.actualRtf() >= 0.9);  // one status update: 0.9 * 1.0 + 0.1 * ~0.5
  const auto cs = swarm.collisionStats();
  CHECK(cs.first == sim.ticks());
  std::printf("ok paced %lld ticks, rtf %.3f, %lld neighbour searches\n", (long long)dticks, sim.actualRtf(), (long long)cs.second);

  // publishers (src/uav_system_ros.cpp:278-282, src/multirotor_simulator.cpp:215): every tick's payload arrives exactly once, one tick
  // late, stamped with its own sim time, and equals what a synchronous getOutputs() sees right after that tick — checked on a twin
  // swarm stepped in lock step (collisions on: the ticks stay lazily evaluated behind the pipelined download)
  {
    UavSwarm twin(n, -1, true);
    twin.construct(0, n, mp, pos, hdg);
    twin.warmUp();
    UavSwarm pub(n, -1, true);
    pub.construct(0, n, mp, pos, hdg);
    pub.warmUp();
    for (int u = 0; u < n; u++) {
      reference::Position c;
      c.position = Eigen::Vector3d(pos[(size_t)u](0) + 1.0, pos[(size_t)u](1) - 1.0, 11.0);
      c.heading  = 0.3;
      twin[u].setInput(c);
      pub[u].setInput(c);
    }
    SimulatorConfig pc = cfg;
    pc.iterate_without_input = true;
    MultirotorSimulator psim(pub, n, pc);
    std::vector<std::vector<mrs_uav_output_t>> want;
    std::vector<double>                        stamps;
    int                                        seen = 0, bad = 0;
    psim.setPublisher([&](double t, const void* payload, int count) {
      const mrs_uav_output_t* o = static_cast<const mrs_uav_output_t*>(payload);
      if (count != n || seen >= (int)want.size() || std::fabs(t - stamps[(size_t)seen]) > 1e-12) bad++;
      else
        for (int u = 0; u < n; u++)
          for (int j = 0; j < 3; j++)
            if (o[u].position[j] != want[(size_t)seen][(size_t)u].position[j] || o[u].linear_acceleration[j] != want[(size_t)seen][(size_t)u].linear_acceleration[j]) bad++;
      seen++;
    });
    const int pticks = 25;
    for (int k = 0; k < pticks; k++) {
      twin.makeStep(1.0 / pc.simulation_rate);
      want.push_back(twin.getOutputs(0, n));
      twin.handleCollisions(pc.collisions_enabled, pc.collisions_crash, pc.collisions_rebounce);
      stamps.push_back(psim.simTime() + 1.0 / pc.simulation_rate);
      psim.timerMain();
      CHECK(seen == k);  // one tick late
    }
    psim.flushPublisher();
    CHECK(seen == pticks && bad == 0);
    std::printf("ok pipelined_publisher %d ticks\n", seen);
  }
  return 0;
}
