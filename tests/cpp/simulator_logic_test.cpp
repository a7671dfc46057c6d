// Unit test of include/mrs_multirotor_simulator/multirotor_simulator.hpp with a recording stand-in for the swarm (no GPU):
// clock, tick order, input watchdog, hold mask, RTF filter, pause/pacing, randd.  Prints "ok" lines; any failure aborts.
#include <cassert>
#include <cstdio>
#include <string>
#include <vector>

#include <mrs_multirotor_simulator/multirotor_simulator.hpp>

using namespace mrs_multirotor_simulator;

struct FakeSwarm {
  std::vector<std::string> log;
  std::vector<int>         hold;
  explicit FakeSwarm(int n) : hold((size_t)n, 0) {}
  void makeStep(double dt) { log.push_back("step " + std::to_string(dt)); }
  void handleCollisions(bool e, bool c, double r) { log.push_back("coll " + std::to_string(e) + std::to_string(c) + " " + std::to_string(r)); }
  void timeoutInput(int first, int count) { log.push_back("timeout " + std::to_string(first) + "+" + std::to_string(count)); }
  void setHold(int first, int count, bool h) {
    for (int k = 0; k < count; k++) hold[(size_t)first + k] = h;
    log.push_back(std::string(h ? "hold " : "release ") + std::to_string(first) + "+" + std::to_string(count));
  }
};
using Sim = BasicMultirotorSimulator<FakeSwarm>;

#define CHECK(c)                                                    \
  do {                                                              \
    if (!(c)) {                                                     \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c);    \
      return 1;                                                     \
    }                                                               \
  } while (0)

int main() {
  {  // tick order and clock decimation (src/multirotor_simulator.cpp:205-229): 250 Hz simulation, 100 Hz clock
    FakeSwarm       sw(3);
    SimulatorConfig cfg;
    cfg.simulation_rate = 250.0;
    cfg.clock_rate      = 100.0;
    cfg.collisions_crash = false;
    Sim sim(sw, 3, cfg);
    int clocks = 0;
    for (int k = 0; k < 250; k++) clocks += sim.timerMain();
    CHECK(sw.log.size() == 500 && sw.log[0].rfind("step 0.004", 0) == 0 && sw.log[1].rfind("coll 10 100", 0) == 0);
    CHECK(std::fabs(sim.simTime() - 1.0) < 1e-12);
    // 4-ms ticks against a 10-ms minimum: a message every 3rd tick (12 ms >= 10 ms(1 - 1e-6))
    CHECK(clocks == 83);
    CHECK(sim.ticks() == 250);
    std::printf("ok clock\n");
  }
  {  // clock at the simulation rate: every tick publishes (the (1 - 1e-6) slack of :221)
    FakeSwarm       sw(1);
    SimulatorConfig cfg;
    Sim             sim(sw, 1, cfg);
    int             clocks = 0;
    for (int k = 0; k < 100; k++) clocks += sim.timerMain();
    CHECK(clocks == 100);
    std::printf("ok clock_same_rate\n");
  }
  {  // input watchdog with iterate_without_input = false (src/uav_system_ros.cpp:243-271)
    FakeSwarm       sw(4);
    SimulatorConfig cfg;
    cfg.iterate_without_input = false;
    cfg.input_timeout         = 0.05;
    Sim sim(sw, 4, cfg);
    CHECK(sw.log.size() == 1 && sw.log[0] == "hold 0+4");  // nobody has an input yet
    for (int k = 0; k < 10; k++) sim.timerMain();
    sim.inputReceived(1);
    sim.inputReceived(2);
    CHECK(sw.hold[0] == 1 && sw.hold[1] == 0 && sw.hold[2] == 0 && sw.hold[3] == 1);
    sw.log.clear();
    for (int k = 0; k < 3; k++) sim.timerMain();  // 30 ms later: nothing yet
    for (auto& l : sw.log) CHECK(l.rfind("timeout", 0) != 0);
    sim.inputReceived(2);  // UAV 2 keeps receiving commands
    sw.log.clear();
    for (int k = 0; k < 4; k++) sim.timerMain();  // UAV 1 is now > 50 ms old
    int n_to = 0;
    for (auto& l : sw.log) n_to += l == "timeout 1+1";
    CHECK(n_to == 1);
    CHECK(sw.hold[1] == 1 && sw.hold[2] == 0 && !sim.hasInput(1) && sim.hasInput(2));
    // the timeout command is set BEFORE the step of the same tick (the UAV is then on hold, so that step skips it)
    size_t i_to = 0, i_hold = 0;
    for (size_t i = 0; i < sw.log.size(); i++) {
      if (sw.log[i] == "timeout 1+1") i_to = i;
      if (sw.log[i] == "hold 1+1") i_hold = i;
    }
    CHECK(i_hold == i_to + 1 && sw.log[i_hold + 1].rfind("step", 0) == 0);
    for (int k = 0; k < 10; k++) sim.timerMain();  // UAV 2 times out too; contiguous runs are batched
    CHECK(!sim.hasInput(2) && sw.hold[2] == 1);
    sw.log.clear();
    for (int k = 0; k < 20; k++) sim.timerMain();  // no stamps left: no scans, no calls
    for (auto& l : sw.log) CHECK(l.rfind("timeout", 0) != 0 && l.rfind("hold", 0) != 0);
    std::printf("ok watchdog\n");
  }
  {  // with iterate_without_input the watchdog still replaces the command, but nothing is put on hold
    FakeSwarm       sw(2);
    SimulatorConfig cfg;
    cfg.input_timeout = 0.02;
    Sim sim(sw, 2, cfg);
    sim.timerMain();
    sim.inputReceived(0);
    sim.inputReceived(1);
    for (int k = 0; k < 5; k++) sim.timerMain();
    int n_to = 0, n_hold = 0;
    for (auto& l : sw.log) n_to += l == "timeout 0+2", n_hold += l.rfind("hold", 0) == 0;
    CHECK(n_to == 1 && n_hold == 0);
    std::printf("ok watchdog_iterate\n");
  }
  {  // RTF telemetry (:238-258) and reconfigure (:264-289)
    FakeSwarm       sw(1);
    SimulatorConfig cfg;
    Sim             sim(sw, 1, cfg);
    for (int k = 0; k < 50; k++) sim.timerMain();  // half a sim second in "one wall second"
    CHECK(std::fabs(sim.timerStatus() - (0.9 * 1.0 + 0.1 * 0.5)) < 1e-12);
    CHECK(std::fabs(sim.timerStatus() - (0.9 * 0.95 + 0.1 * 0.0)) < 1e-12);
    CHECK(std::fabs(sim.wallPeriod() - 0.01) < 1e-15);
    sim.reconfigure(4.0, false, false, true, 50.0);
    CHECK(std::fabs(sim.wallPeriod() - 0.0025) < 1e-15);
    sw.log.clear();
    sim.timerMain();
    CHECK(sw.log[1].rfind("coll 01 50", 0) == 0);
    std::printf("ok rtf\n");
  }
  {  // pacing and pause: 200 Hz x RTF 2 for 0.25 wall seconds ~ 100 ticks; paused: none
    FakeSwarm       sw(1);
    SimulatorConfig cfg;
    cfg.simulation_rate = 200.0;
    cfg.realtime_factor = 2.0;
    Sim sim(sw, 1, cfg);
    sim.spinFor(0.25);
    CHECK(sim.ticks() >= 10 && sim.ticks() <= 101);  // never more than rate x RTF x time; a loaded host may lose ticks (no catch-up bursts)
    const int64_t before = sim.ticks();
    sim.reconfigure(2.0, true, true, true, 100.0);
    sim.spinFor(0.05);
    CHECK(sim.ticks() == before);
    std::printf("ok pacing %lld\n", (long long)before);
  }
  {  // randd (src/uav_system_ros.cpp:653-658): span floor(to - from), sample through float, libc generator
    std::srand(1);
    const double a = randd(-15.0, 15.0);
    std::srand(1);
    const double expect = std::floor(30.0) * (double((float)std::rand()) / double(RAND_MAX)) + -15.0;
    CHECK(a == expect);
    std::srand(1);
    const double b = randd(-0.4, 0.4);  // floor(0.8) = 0: no randomisation below one metre — the quirk
    CHECK(b == -0.4);
    std::srand(7);
    double x = 1, y = 2, z = 3, h = 0.5;
    randomizeSpawn(15, 15, 15, x, y, z, h);
    std::srand(7);
    double ex = 1 + randd(-15, 15), ey = 2 + randd(-15, 15), ez = 3 + randd(-15, 15), eh = 0.5 + randd(-3.14, 3.14);
    CHECK(x == ex && y == ey && z == ez && h == eh);
    std::printf("ok randd\n");
  }
  return 0;
}
