// multirotor_simulator.hpp — the reference's simulator loop without ROS, on top of the batch engine.
//
// What the reference's nodelet does around the hot path (src/multirotor_simulator.cpp:198-289) and what every
// UavSystemRos does around its UavSystem (src/uav_system_ros.cpp:243-283, 653-658, 679-1022), restated as plain C++ so
// that a port keeps the behaviour when the per-UAV objects become one swarm:
//   * sim clock: sim_time += 1/simulation_rate per tick; a clock message is due when sim_time - last_published >=
//     (1/clock_rate)(1 - 1e-6)                                                     (timerMain, :205-229)
//   * tick order: makeStep for every UAV -> publishPoses -> handleCollisions       (:211-217)
//   * pacing: wall period 1 / (simulation_rate * realtime_factor); pause stops the timer; the dynamic-reconfigure
//     parameters (realtime_factor, paused, collisions enabled / crash / rebounce) may change between ticks (:264-289)
//   * telemetry: once per wall second, actual_rtf = 0.9 actual_rtf + 0.1 (sim seconds advanced in that second) (:238-258)
//   * input watchdog per UAV: a command stamps time_last_input; when now - time_last_input > input_timeout the UAV gets
//     the safe command of its mode (timeoutInput) and the stamp is cleared; the model is iterated only
//     if (iterate_without_input || time_last_input > 0)                            (src/uav_system_ros.cpp:243-271)
//   * spawn randomisation `randd`                                                  (src/uav_system_ros.cpp:89-94, 653-658)
//   * publishers: every UAV's odometry / IMU / range after its step (src/uav_system_ros.cpp:278-282) and the pose array every
//     tick (src/multirotor_simulator.cpp:215, 365-389) — here ONE packed download per tick, started behind the tick's launch and
//     handed to the publisher callback while the NEXT tick runs (mrs_swarm_get_outputs_async): messages leave one tick late in
//     wall time, stamped with the sim time of the tick they describe
// Time is kept in integer nanoseconds like ros::Time.  "now" for the watchdog is the last PUBLISHED clock value: the
// reference's callbacks read ros::Time::now(), which under use_sim_time is the /clock message this very node sent last.
//
// The class is a template over the swarm type so that the logic can be unit-tested without a GPU; `MultirotorSimulator`
// below is the instantiation over UavSwarm.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

namespace mrs_multirotor_simulator {

// config/multirotor_simulator.yaml (defaults of the shipped file)
struct SimulatorConfig {
  double simulation_rate       = 100.0;  // Hz
  double clock_rate            = 100.0;  // Hz
  double realtime_factor       = 1.0;
  bool   paused                = false;
  bool   collisions_enabled    = true;
  bool   collisions_crash      = true;
  double collisions_rebounce   = 100.0;
  bool   iterate_without_input = true;
  double input_timeout         = 1.0;  // s
};

// UavSystemRos::randd (src/uav_system_ros.cpp:653-658), quirks included: the span is floor(to - from) and the unit sample
// goes through float.  Uses the C library generator like the reference (never seeded there: srand default).
inline double randd(double from, double to) {
  const double zero_to_one = double((float)std::rand()) / double(RAND_MAX);
  return std::floor(to - from) * zero_to_one + from;
}

// spawn randomisation of one UAV (src/uav_system_ros.cpp:89-94): four draws, in this order
inline void randomizeSpawn(double bounds_x, double bounds_y, double bounds_z, double& x, double& y, double& z, double& heading) {
  x += randd(-bounds_x, bounds_x);
  y += randd(-bounds_y, bounds_y);
  z += randd(-bounds_z, bounds_z);
  heading += randd(-3.14, 3.14);
}

namespace detail {
// does the swarm type offer the pipelined publisher download (UavSwarm does; the GPU-less test doubles need not)
template <class S, class = void>
struct has_async_outputs : std::false_type {};
template <class S>
struct has_async_outputs<S, std::void_t<decltype(std::declval<S&>().outputsWait(std::declval<S&>().getOutputsAsync(0, 0)))>> : std::true_type {};
}  // namespace detail

template <class SwarmT>
class BasicMultirotorSimulator {
public:
  using ns_t = int64_t;
  // publisher callback: (sim time of the tick the payload describes, packed payloads of all UAVs — mrs_uav_output_t[count] —, count)
  using PublishFn = std::function<void(double, const void*, int)>;

  BasicMultirotorSimulator(SwarmT& swarm, int n_uavs, const SimulatorConfig& cfg, double sim_time_start = 0.0)
      : swarm_(swarm), cfg_(cfg), n_(n_uavs), time_last_input_((size_t)n_uavs, 0) {
    sim_time_            = toNs(sim_time_start);
    last_published_time_ = sim_time_;
    last_sim_time_status_ = sim_time_;
    // UavSystemRos constructor: time_last_input_ = 0 (src/uav_system_ros.cpp:10) — nothing is iterated before the first
    // command unless iterate_without_input
    if (!cfg_.iterate_without_input && n_ > 0) swarm_.setHold(0, n_, true);
  }

  // ---- subscriber callbacks (src/uav_system_ros.cpp:679-1022): after swarm[i].setInput(cmd) ----
  void inputReceived(int uav) {
    ns_t& t = time_last_input_[(size_t)uav];
    if (t == 0 && !cfg_.iterate_without_input) swarm_.setHold(uav, 1, false);
    t = std::max<ns_t>(now(), 1);  // "> ros::Time(0)" means "has an input"
    if (oldest_input_ == 0 || t < oldest_input_) oldest_input_ = t;
  }

  // ---- publishers (src/uav_system_ros.cpp:278-282, src/multirotor_simulator.cpp:215): fn gets every tick's payload, one tick late ----
  void setPublisher(PublishFn fn) { publish_ = std::move(fn); }
  // hand out the payload of the last tick (end of a run, before a pause: nothing stays in flight)
  void flushPublisher() {
    if constexpr (detail::has_async_outputs<SwarmT>::value) {
      if (pending_ticket_ >= 0 && publish_) {
        int count = 0;
        const void* v = swarm_.outputsWait(pending_ticket_, &count);
        publish_(toSec(pending_time_), v, count);
      }
      pending_ticket_ = -1;
    }
  }

  // ---- timerMain (src/multirotor_simulator.cpp:198-230): one tick; true when a clock message is due ----
  bool timerMain() {
    const double step = 1.0 / cfg_.simulation_rate;
    sim_time_ += toNs(step);
    checkInputTimeouts();                                                                  // UavSystemRos::makeStep, first half
    swarm_.makeStep(step);                                                                 // :211-213
    if constexpr (detail::has_async_outputs<SwarmT>::value) {                              // :215 publishPoses (+ uav_system_ros.cpp:278-282)
      if (publish_ && n_ > 0) {
        const int ticket = swarm_.getOutputsAsync(0, n_);  // behind this tick's launch; the copy runs beside the next tick
        const ns_t stamp = sim_time_;
        flushPublisher();                                  // the PREVIOUS tick's payload: landed while this tick was being queued
        pending_ticket_ = ticket;
        pending_time_   = stamp;
      }
    }
    swarm_.handleCollisions(cfg_.collisions_enabled, cfg_.collisions_crash, cfg_.collisions_rebounce);  // :217
    ticks_++;
    if (toSec(sim_time_ - last_published_time_) >= (1.0 / cfg_.clock_rate) * (1.0 - 1e-6)) {  // :221
      last_published_time_ = sim_time_;
      return true;
    }
    return false;
  }

  // ---- timerStatus (:238-258): call once per wall second ----
  double timerStatus() {
    const double last_sec_rtf = toSec(sim_time_ - last_sim_time_status_) / 1.0;
    last_sim_time_status_     = sim_time_;
    actual_rtf_               = 0.9 * actual_rtf_ + 0.1 * last_sec_rtf;
    return actual_rtf_;
  }

  // ---- callbackDrs (:264-289) ----
  void reconfigure(double realtime_factor, bool paused, bool collisions_enabled, bool collisions_crash, double collisions_rebounce) {
    cfg_.realtime_factor     = realtime_factor;
    cfg_.paused              = paused;
    cfg_.collisions_enabled  = collisions_enabled;
    cfg_.collisions_crash    = collisions_crash;
    cfg_.collisions_rebounce = collisions_rebounce;
  }

  double wallPeriod() const { return 1.0 / (cfg_.simulation_rate * cfg_.realtime_factor); }  // :181, :285

  // Wall-clock paced loop: the WallTimer of the reference.  Runs for `wall_seconds`, ticking every wallPeriod() unless
  // paused, calling timerStatus() once per wall second.  on_clock(sim_seconds) is invoked for every due clock message.
  template <class ClockFn>
  void spinFor(double wall_seconds, ClockFn on_clock) {
    using clk            = std::chrono::steady_clock;
    const auto t_begin   = clk::now();
    auto       next_tick = t_begin, next_status = t_begin + std::chrono::seconds(1);
    while (std::chrono::duration<double>(clk::now() - t_begin).count() < wall_seconds) {
      const auto t = clk::now();
      if (t >= next_status) {
        timerStatus();
        next_status += std::chrono::seconds(1);
      }
      if (cfg_.paused) {  // timer_main_.stop()
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        next_tick = clk::now();
        continue;
      }
      if (t < next_tick) {
        std::this_thread::sleep_until(std::min(next_tick, next_status));
        continue;
      }
      if (timerMain()) on_clock(toSec(sim_time_));
      next_tick += std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(wallPeriod()));
      if (next_tick < clk::now()) next_tick = clk::now();  // a late timer does not fire a burst to catch up
    }
  }
  void spinFor(double wall_seconds) {
    spinFor(wall_seconds, [](double) {});
  }

  double  simTime() const { return toSec(sim_time_); }
  double  actualRtf() const { return actual_rtf_; }
  int64_t ticks() const { return ticks_; }
  bool    hasInput(int uav) const { return time_last_input_[(size_t)uav] > 0; }
  const SimulatorConfig& config() const { return cfg_; }

private:
  static ns_t   toNs(double s) { return (ns_t)std::llround(s * 1e9); }
  static double toSec(ns_t t) { return (double)t * 1e-9; }
  ns_t          now() const { return last_published_time_; }

  // UavSystemRos::makeStep, :247-261, for all UAVs.  `oldest_input_` is a lower bound of the live stamps, so the scan only
  // runs when somebody can actually have timed out.
  void checkInputTimeouts() {
    if (oldest_input_ == 0) return;
    const ns_t t = now();
    if (toSec(t - oldest_input_) <= cfg_.input_timeout) return;
    ns_t oldest = 0;
    int  run_begin = -1;
    for (int i = 0; i <= n_; i++) {
      bool expired = false;
      if (i < n_) {
        ns_t& tli = time_last_input_[(size_t)i];
        if (tli > 0) {
          if (toSec(t - tli) > cfg_.input_timeout) {
            expired = true;
            tli     = 0;
          } else if (oldest == 0 || tli < oldest) {
            oldest = tli;
          }
        }
      }
      if (expired && run_begin < 0) run_begin = i;
      if (!expired && run_begin >= 0) {  // contiguous runs go down as one call
        swarm_.timeoutInput(run_begin, i - run_begin);
        if (!cfg_.iterate_without_input) swarm_.setHold(run_begin, i - run_begin, true);
        run_begin = -1;
      }
    }
    oldest_input_ = oldest;
  }

  SwarmT&           swarm_;
  SimulatorConfig   cfg_;
  int               n_;
  std::vector<ns_t> time_last_input_;
  ns_t              oldest_input_ = 0;
  ns_t              sim_time_ = 0, last_published_time_ = 0, last_sim_time_status_ = 0;
  double            actual_rtf_ = 1.0;  // multirotor_simulator.cpp:62
  int64_t           ticks_      = 0;
  PublishFn         publish_;
  int               pending_ticket_ = -1;
  ns_t              pending_time_   = 0;
};

}  // namespace mrs_multirotor_simulator

#include "uav_system/uav_system.hpp"

namespace mrs_multirotor_simulator {
// the simulator loop over the GPU swarm
using MultirotorSimulator = BasicMultirotorSimulator<UavSwarm>;
}  // namespace mrs_multirotor_simulator
