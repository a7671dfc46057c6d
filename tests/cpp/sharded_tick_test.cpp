// sharded_tick_test.cpp — the library-driven multi-GPU tick (mrs_swarm_comm_init / mrs_swarm_tick_sharded_n) from a plain C++ host:
// no PyTorch in the process, RCCL comes from the system loader path.  A one-rank communicator on the one GPU of the test box; the
// sharded ticks must give what the local ticks give on a twin swarm (timerMain order, src/multirotor_simulator.cpp:211-217).
// Exit code 0 and "ok ..." lines on success.
#include <cmath>
#include <cstdio>
#include <mrs_multirotor_simulator/multirotor_simulator.hpp>

using namespace mrs_multirotor_simulator;

#define CHECK(c)                                                 \
  do {                                                           \
    if (!(c)) {                                                  \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
      return 1;                                                  \
    }                                                            \
  } while (0)

int main() {
  const int                    n = 700;
  MultirotorModel::ModelParams mp;
  mp.ground_enabled = true;
  mp.ground_z       = 0.0;
  std::vector<Eigen::Vector3d> pos;
  std::vector<double>          hdg;
  for (int i = 0; i < n; i++) {  // pairs 0.6 m apart (inside the collision distance of two x500), pairs 3 m from each other
    const int pair = i / 2;
    pos.push_back(Eigen::Vector3d(3.0 * (pair % 20) + 0.6 * (i % 2), 3.0 * (pair / 20), 8.0 + 0.01 * (i % 7)));
    hdg.push_back(0.05 * i);
  }
  UavSwarm a(n, -1, /*fast_arithmetic=*/false), b(n, -1, false);
  for (UavSwarm* s : {&a, &b}) {
    s->construct(0, n, mp, pos, hdg);
    s->warmUp();
    for (int i = 0; i < n; i++) {
      reference::Position c;
      c.position = Eigen::Vector3d(pos[(size_t)i](0), pos[(size_t)i](1), pos[(size_t)i](2) + 1.0);
      c.heading  = 0.0;
      (*s)[i].setInput(c);
    }
  }
  bool refused = false;
  try {
    a.tickSharded(0.001, 1, true, false, 100.0);
  } catch (const std::exception&) {
    refused = true;
  }
  CHECK(refused);  // no communicator yet
  const auto id = UavSwarm::commUniqueId();
  a.commInit(1, 0, id, n);
  std::printf("ok communicator\n");

  a.tickSharded(0.001, 20, true, false, 100.0);
  b.tick(0.001, 20, true, false, 100.0);
  const std::vector<double> xa = a.getPoses(), xb = b.getPoses();
  double                    worst = 0, moved = 0;
  for (size_t k = 0; k < xa.size(); k++) {
    worst = std::fmax(worst, std::fabs(xa[k] - xb[k]) / (1.0 + std::fabs(xb[k])));
  }
  for (int i = 0; i < n; i++) moved = std::fmax(moved, std::fabs(xa[(size_t)i * 3] - pos[(size_t)i](0)));
  std::printf("worst relative difference to the local ticks %.3e, largest push along x %.3f m\n", worst, moved);
  CHECK(worst < 1e-12);
  CHECK(moved > 0.001);  // the rebounce forces did act
  CHECK(a.collisionStats().first == 20);
  std::printf("ok sharded_ticks_equal_local_ticks\n");

  a.tickSharded(0.001, 1, true, true, 100.0);  // crash mode: the pairs are still within reach, every UAV goes down
  b.tick(0.001, 1, true, true, 100.0);
  int crashed = 0;
  for (int i = 0; i < n; i++) crashed += a[i].hasCrashed();
  CHECK(crashed == n);
  a.tickSharded(0.001, 200, true, false, 100.0);  // and falls; the ticks keep agreeing with the local ones
  b.tick(0.001, 200, true, false, 100.0);
  const std::vector<double> ya = a.getPoses(), yb = b.getPoses();
  for (size_t k = 0; k < ya.size(); k++) CHECK(std::fabs(ya[k] - yb[k]) <= 1e-12 * (1.0 + std::fabs(yb[k])));
  CHECK(ya[2] < pos[0](2) - 0.1);
  a.commDestroy();
  std::printf("ok crash_and_destroy\n");
  return 0;
}
