// sharded_tick_test.cpp — the library-driven multi-GPU tick (mrs_swarm_comm_init / mrs_swarm_tick_sharded_n) from a plain C++ host:
// no PyTorch in the process, RCCL comes from the system loader path.  A one-rank communicator on the one GPU of the test box; the
// sharded ticks must give what the local ticks give on a twin swarm (timerMain order, src/multirotor_simulator.cpp:211-217).
// Exit code 0 and "ok ..." lines on success.
#include <cmath>
#include <cstdio>
#include <memory>
#include <thread>
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

  // ---- three shards of one swarm in this process (x-sorted slabs, public index kept through the permutation), one host thread each,
  //      exchanging through an in-process loopback group: the export-set exchange from a plain C++ host ----
  {
    const int world = 3, m = 900;
    std::vector<Eigen::Vector3d> p3;
    for (int i = 0; i < m; i++) {  // a sheet of UAVs 1.9 m apart in x (collisions only inside the pairs below), slab faces cut through it
      const int pair = i / 2;
      p3.push_back(Eigen::Vector3d(1.9 * (pair % 30) + 0.55 * (i % 2), 2.5 * (pair / 30), 9.0 + 0.02 * (i % 5)));
    }
    const std::vector<int64_t> order = UavSwarm::slabPartition(p3, world);
    UavSwarm whole(m, -1, false);
    whole.construct(0, m, mp, p3, std::vector<double>((size_t)m, 0.0));
    whole.warmUp();
    mrs_loopback_group_t* group = nullptr;
    CHECK(mrs_loopback_group_create(world, &group) == MRS_OK);
    std::vector<std::unique_ptr<UavSwarm>> shard;
    std::vector<std::vector<int64_t>>      own((size_t)world);
    for (int r = 0; r < world; r++) {
      const int64_t base = m / world, rem = m % world, lo = r * base + (r < rem ? r : rem), hi = lo + base + (r < rem ? 1 : 0);
      std::vector<Eigen::Vector3d> pr;
      for (int64_t k = lo; k < hi; k++) {
        own[(size_t)r].push_back(order[(size_t)k]);
        pr.push_back(p3[(size_t)order[(size_t)k]]);
      }
      shard.push_back(std::make_unique<UavSwarm>((int)(hi - lo), -1, false));
      shard.back()->construct(0, (int)(hi - lo), mp, pr, std::vector<double>(pr.size(), 0.0));
      shard.back()->warmUp();
      shard.back()->commInitLoopback(group, r, m);
    }
    for (int i = 0; i < m; i++) {
      reference::Position c;
      c.position = Eigen::Vector3d(p3[(size_t)i](0) + 0.5, p3[(size_t)i](1), p3[(size_t)i](2) + 1.0);
      whole[i].setInput(c);
    }
    for (int r = 0; r < world; r++)
      for (size_t k = 0; k < own[(size_t)r].size(); k++) {
        reference::Position c;
        const auto&         q = p3[(size_t)own[(size_t)r][k]];
        c.position = Eigen::Vector3d(q(0) + 0.5, q(1), q(2) + 1.0);
        (*shard[(size_t)r])[(int)k].setInput(c);
      }
    whole.tick(0.001, 150, true, false, 100.0);
    std::vector<std::thread> th;
    std::vector<int>         failed((size_t)world, 0);
    for (int r = 0; r < world; r++)
      th.emplace_back([&, r] {
        try {
          shard[(size_t)r]->tickSharded(0.001, 150, true, false, 100.0);  // collective: every rank from its own thread
        } catch (const std::exception& e) {
          std::printf("rank %d: %s\n", r, e.what());
          failed[(size_t)r] = 1;
        }
      });
    for (auto& t : th) t.join();
    for (int r = 0; r < world; r++) CHECK(!failed[(size_t)r]);
    const std::vector<double> xw = whole.getPoses();
    double                    worst3 = 0;
    for (int r = 0; r < world; r++) {
      const std::vector<double> xs = shard[(size_t)r]->getPoses();
      for (size_t k = 0; k < own[(size_t)r].size(); k++)
        for (int j = 0; j < 3; j++) {
          const double ref = xw[(size_t)own[(size_t)r][k] * 3 + (size_t)j];
          worst3           = std::fmax(worst3, std::fabs(xs[k * 3 + (size_t)j] - ref) / (1.0 + std::fabs(ref)));
        }
    }
    const mrs_comm_info_t ci = shard[1]->commInfo();
    std::printf("3 loopback shards vs one swarm: worst relative difference %.3e; rank 1 sends %lld B per tick (%lld on a search tick), %lld searches in %lld ticks\n",
                worst3, (long long)ci.bytes_per_tick, (long long)ci.bytes_per_rebuild, (long long)ci.searches, (long long)ci.ticks);
    CHECK(worst3 < 1e-12);
    CHECK(ci.exchange == MRS_EXCHANGE_EXPORT_SETS && ci.ticks == 150 && ci.searches >= 1 && ci.searches < 60);
    CHECK(ci.bytes_per_tick < ci.bytes_per_rebuild);
    const UavSwarm::SearchStats ss = shard[1]->searchStats();  // (the same counts on every rank: the decisions follow from collective data)
    std::printf("searches %lld, of them on a halo exchange %lld, repeated on all records %lld\n", (long long)ss.searches, (long long)ss.halo_searches, (long long)ss.halo_repeats);
    CHECK(ss.searches == ci.searches && ss.halo_searches + ss.halo_repeats <= ss.searches && ss.halo_repeats <= ss.halo_searches);
    for (int r = 0; r < world; r++) CHECK(shard[(size_t)r]->searchStats().halo_searches == ss.halo_searches);
    th.clear();
    for (int r = 0; r < world; r++) th.emplace_back([&, r] { shard[(size_t)r]->commDestroy(); });
    for (auto& t : th) t.join();
    shard.clear();
    CHECK(mrs_loopback_group_destroy(group) == MRS_OK);
    std::printf("ok loopback_export_sets\n");
  }
  return 0;
}
