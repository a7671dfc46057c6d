// uav_system.hpp — drop-in facade for the reference's header API
// (include/mrs_multirotor_simulator/uav_system/uav_system.hpp:16-118) on top of the C ABI of include/mrs_swarm.h.
//
//   mrs_multirotor_simulator::UavSystem   same public methods, same argument types, same semantics.  All UavSystem objects of a
//                                          process live in ONE pooled swarm on the GPU (UavPool below): the reference's unchanged loop
//                                          `for (i) uavs_[i]->makeStep(dt)` (src/multirotor_simulator.cpp:211-213) costs one launch per
//                                          tick — the first makeStep of a round steps the whole pool, the others consume their result —
//                                          and getState() after each call (src/uav_system_ros.cpp:270-282) is served from one download
//   mrs_multirotor_simulator::UavSwarm    the batch owner the simulator loop should hold instead of
//                                          std::vector<std::unique_ptr<UavSystemRos>> (src/multirotor_simulator.cpp:70):
//                                          swarm[i] is a non-owning view with the UavSystem API, swarm.makeStep(dt)
//                                          replaces the serial loop of src/multirotor_simulator.cpp:211-213 by one launch
//                                          and swarm.handleCollisions(...) replaces :295-359.
//
// All arithmetic happens in libmrs_swarm.so (HIP); this header only marshals values.  Errors of the library surface as
// std::runtime_error (the reference's API is void and cannot fail, the GPU can).
#ifndef UAV_SYSTEM_H
#define UAV_SYSTEM_H

#include <cstdlib>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <utility>
#include <array>
#include <vector>

#include "multirotor_model.hpp"

#include "controllers/pid.hpp"
#include "controllers/mixer.hpp"
#include "controllers/rate_controller.hpp"
#include "controllers/attitude_controller.hpp"
#include "controllers/acceleration_controller.hpp"
#include "controllers/velocity_controller.hpp"
#include "controllers/position_controller.hpp"

namespace mrs_multirotor_simulator
{

// ------------------------------------------------------------------------------------------------------------------
// The pool behind the stand-alone UavSystem objects: one swarm on the GPU, one slot per object.
//
// The reference steps its UAVs one by one (src/multirotor_simulator.cpp:211-213) and looks at each state right after
// (src/uav_system_ros.cpp:270-282).  Done literally on a GPU that is a launch and a synchronisation per UAV and tick.  The pool
// SPECULATES instead: the first makeStep(dt) of a round — recognised by an object stepping for the second time — saves a copy of
// the pool's state, steps EVERY slot with one launch and downloads all states with one copy; the makeStep(dt) calls of the other
// objects then find their slot "ahead" and consume the result, and their getState() is served from the download.  Whatever does not
// fit the guess is put right, slot by slot: an object that is read or written before its makeStep of the round (setInput from a
// subscriber callback, applyForce from handleCollisions, ...) is rolled back to its pre-step copy first, so the call sees and changes
// the state the reference would, and its later makeStep steps that slot alone (mrs_swarm_step_range); so does a makeStep with
// another dt.  Results never depend on the guess — only the number of launches does.  A caller that steps few of its objects per
// round would pay for the undone work: the pool notices (more than half of a round unclaimed) and steps slot by slot until it has
// seen a full round again.
// ------------------------------------------------------------------------------------------------------------------
class UavPool {
public:
  struct Stats { long long rounds = 0, consumed = 0, single_steps = 0, rollbacks = 0, state_hits = 0, state_misses = 0, grows = 0; };

  static UavPool& instance() {
    // never destroyed: UavSystem objects of static storage may outlive any function-local static constructed after them, and their
    // destructors call release(); the swarms go with the process (no HIP call during runtime teardown)
    static UavPool* const pool = new UavPool;
    return *pool;
  }
  // before the first UavSystem is created: device and arithmetic flavour of the pool (default: current device, FAST; the environment
  // variable MRS_FACADE_ARITH=literal selects the reference's operation order, MRS_FACADE_SPECULATE=0 steps every object on its own)
  static void configure(int device_id, bool fast_arithmetic) {
    UavPool& p = instance();
    std::lock_guard<std::recursive_mutex> lk(p.mtx_);
    if (p.live_) throw std::logic_error("UavPool::configure: the pool is already in use");
    p.device_ = device_id;
    p.fast_   = fast_arithmetic;
  }

  std::recursive_mutex& mutex() { return mtx_; }
  mrs_swarm_t*          live() { return live_; }
  Stats                 stats() {
    std::lock_guard<std::recursive_mutex> lk(mtx_);
    return stats_;
  }
  int size() {
    std::lock_guard<std::recursive_mutex> lk(mtx_);
    return n_live_;
  }

  int acquire() {  // a slot holding UavSystem()
    std::lock_guard<std::recursive_mutex> lk(mtx_);
    if (free_.empty()) grow();
    const int slot = free_.back();
    free_.pop_back();
    if (dirty_[(size_t)slot]) mrs_throw_on_error(mrs_swarm_construct(live_, slot, 1, nullptr, nullptr, nullptr));  // (a reused slot)
    used_[(size_t)slot] = 1;
    dirty_[(size_t)slot] = 1;
    ahead_[(size_t)slot] = 0;
    round_of_[(size_t)slot] = 0;
    cached_[(size_t)slot] = 0;
    mode_[(size_t)slot] = MRS_INPUT_UNKNOWN;
    for (int k = 0; k < kCmdWidth; k++) cmd_[(size_t)slot * kCmdWidth + k] = 0.0;
    for (int k = 0; k < 3; k++) force_[(size_t)slot * 3 + k] = 0.0;
    n_live_++;
    return slot;
  }
  void release(int slot) noexcept {
    std::lock_guard<std::recursive_mutex> lk(mtx_);
    if (slot < 0 || slot >= cap_ || !used_[(size_t)slot]) return;
    if (ahead_[(size_t)slot]) n_ahead_--;
    if (round_of_[(size_t)slot] == round_ && !speculate_) seen_in_round_--;
    used_[(size_t)slot] = ahead_[(size_t)slot] = cached_[(size_t)slot] = 0;
    if (cmd_dirty_[(size_t)slot]) {
      cmd_dirty_[(size_t)slot] = 0;
      n_cmd_dirty_--;
    }
    free_.push_back(slot);
    n_live_--;
  }

  // before any access to the slot other than makeStep: the caller must see (and change) the state the reference object would have
  void touch(int slot, bool writes) {
    if (ahead_[(size_t)slot]) early_access(slot);
    flush_pending();
    if (writes) cached_[(size_t)slot] = 0;
  }
  // UavSystem::applyForce / setInput of a pooled object: kept in host tables and sent in ONE upload before the next launch (the
  // reference's handleCollisions calls applyForce for EVERY UAV on every tick, its subscriber callbacks setInput at command rate:
  // a device write per call would cost more than the step).  A command equal to the one in force changes nothing — not even the guess.
  void applyForce(int slot, const double f[3]) {
    std::lock_guard<std::recursive_mutex> lk(mtx_);
    double* cur = &force_[(size_t)slot * 3];
    if (cur[0] == f[0] && cur[1] == f[1] && cur[2] == f[2]) return;
    if (ahead_[(size_t)slot]) early_access(slot);
    cur[0] = f[0]; cur[1] = f[1]; cur[2] = f[2];
    force_dirty_ = true;
  }
  void setInput(int slot, int mode, const double* payload, int n) {
    std::lock_guard<std::recursive_mutex> lk(mtx_);
    double*    cur  = &cmd_[(size_t)slot * kCmdWidth];
    bool       same = mode_[(size_t)slot] == mode;
    for (int k = 0; k < kCmdWidth && same; k++) same = cur[k] == (k < n ? payload[k] : 0.0);
    if (same && mode != MRS_INPUT_UNKNOWN) return;
    if (ahead_[(size_t)slot]) early_access(slot);
    for (int k = 0; k < kCmdWidth; k++) cur[k] = k < n ? payload[k] : 0.0;
    mode_[(size_t)slot] = mode;
    if (!cmd_dirty_[(size_t)slot]) {
      cmd_dirty_[(size_t)slot] = 1;
      n_cmd_dirty_++;
    }
  }
  // the downloaded state of a slot, if it still is the slot's state (nullptr: ask the device)
  const mrs_uav_state_t* cached(int slot) {
    if (ahead_[(size_t)slot]) early_access(slot);
    if (cached_[(size_t)slot]) {
      stats_.state_hits++;
      return &cache_[(size_t)slot];
    }
    stats_.state_misses++;
    return nullptr;
  }

  // dst = src (UavSystem copy construction / assignment): device state and the host tables
  void copy(int dst, int src) {
    std::lock_guard<std::recursive_mutex> lk(mtx_);
    touch(src, false);
    touch(dst, true);
    mrs_throw_on_error(mrs_swarm_copy_uavs(live_, dst, live_, src, 1));
    mode_[(size_t)dst] = mode_[(size_t)src];
    for (int k = 0; k < kCmdWidth; k++) cmd_[(size_t)dst * kCmdWidth + k] = cmd_[(size_t)src * kCmdWidth + k];
    for (int k = 0; k < 3; k++) force_[(size_t)dst * 3 + k] = force_[(size_t)src * 3 + k];
    dirty_[(size_t)dst] = 1;
  }

  void makeStep(int slot, double dt) {
    std::lock_guard<std::recursive_mutex> lk(mtx_);
    if (ahead_[(size_t)slot]) {
      if (dt == round_dt_) {  // the round's launch has stepped this slot already
        ahead_[(size_t)slot] = 0;
        n_ahead_--;
        round_of_[(size_t)slot] = round_;
        stats_.consumed++;
        return;
      }
      rollback(slot);  // (stepped with another dt than this object asks for)
    }
    const bool again = round_of_[(size_t)slot] == round_;  // this object has made its step of the running round: the next round begins
    if (again && !speculate_) {  // observing: has every object stepped since this one stepped last?
      if (seen_in_round_ == n_live_ && allow_speculation_) speculate_ = true;
      round_++;
      seen_in_round_ = 0;
    }
    if (again && speculate_ && n_live_ > 1) {
      if (n_ahead_ > 0) {  // results of the old round nobody asked for: undo them
        const bool mostly_unclaimed = 2 * n_ahead_ > n_live_;
        for (int k = 0; k < cap_ && n_ahead_ > 0; k++)
          if (ahead_[(size_t)k]) rollback(k);
        if (mostly_unclaimed) {  // this caller does not step its objects round by round
          speculate_ = false;
          round_++;
          seen_in_round_ = 0;
          step_alone(slot, dt);
          return;
        }
      }
      // ---- a round: pending commands and forces, copy, ONE launch over every slot, ONE download ----
      flush_pending();
      round_rollbacks_ = 0;
      mrs_throw_on_error(mrs_swarm_copy_uavs(backup_, 0, live_, 0, cap_));
      mrs_throw_on_error(mrs_swarm_step(live_, dt));
      mrs_throw_on_error(mrs_swarm_get_states(live_, 0, cap_, cache_.data()));
      round_++;
      round_dt_ = dt;
      n_ahead_  = 0;
      for (int k = 0; k < cap_; k++) {
        ahead_[(size_t)k]  = used_[(size_t)k] && k != slot;
        cached_[(size_t)k] = used_[(size_t)k];
        n_ahead_ += ahead_[(size_t)k];
        if (!used_[(size_t)k]) dirty_[(size_t)k] = 1;  // the launch has stepped the free slots too: no longer UavSystem()'s zero state
      }
      round_of_[(size_t)slot] = round_;
      stats_.rounds++;
      return;
    }
    step_alone(slot, dt);
  }

  ~UavPool() {
    if (live_) mrs_swarm_destroy(live_);
    if (backup_) mrs_swarm_destroy(backup_);
  }

private:
  UavPool() {
    if (const char* e = std::getenv("MRS_FACADE_ARITH")) fast_ = !(e[0] == 'l' || e[0] == 'L' || e[0] == '0');
    if (const char* e = std::getenv("MRS_FACADE_SPECULATE")) allow_speculation_ = std::atoi(e) != 0;
  }
  UavPool(const UavPool&) = delete;
  UavPool& operator=(const UavPool&) = delete;

  void step_alone(int slot, double dt) {
    flush_pending();
    mrs_throw_on_error(mrs_swarm_step_range(live_, slot, 1, dt));
    cached_[(size_t)slot] = 0;
    if (round_of_[(size_t)slot] != round_) {
      round_of_[(size_t)slot] = round_;
      if (!speculate_) seen_in_round_++;
    }
    stats_.single_steps++;
    if (!allow_speculation_) speculate_ = false;
  }
  // a slot is read or written before its makeStep of the round: back to its pre-step copy.  When that happens to most of a round
  // the round starts at the wrong place of the caller's loop (its first object should trigger it): drop the guess, watch a full
  // round of single steps, start again from the first object that repeats.
  void early_access(int slot) {
    rollback(slot);
    if (2 * ++round_rollbacks_ > n_live_ && speculate_) {
      for (int k = 0; k < cap_ && n_ahead_ > 0; k++)
        if (ahead_[(size_t)k]) rollback(k);
      speculate_ = false;
      round_++;
      seen_in_round_ = 0;
    }
  }
  // host tables -> device: all forces in one call; commands one by one while few, else run by run of equal input mode
  void flush_pending() {
    if (force_dirty_) {
      force_dirty_ = false;
      mrs_throw_on_error(mrs_swarm_apply_force(live_, 0, cap_, force_.data()));
    }
    if (n_cmd_dirty_ == 0) return;
    if (n_cmd_dirty_ <= 4) {
      for (int k = 0; k < cap_ && n_cmd_dirty_ > 0; k++)
        if (cmd_dirty_[(size_t)k]) send_commands(k, 1);
      return;
    }
    for (int k = 0; k < cap_;) {
      int e = k + 1;
      while (e < cap_ && mode_[(size_t)e] == mode_[(size_t)k] && used_[(size_t)e] == used_[(size_t)k]) e++;
      bool any = false;
      for (int j = k; j < e; j++) any = any || cmd_dirty_[(size_t)j];
      if (any && used_[(size_t)k]) send_commands(k, e - k);
      k = e;
    }
  }
  void send_commands(int first, int count) {
    mrs_throw_on_error(mrs_swarm_set_input(live_, first, count, mode_[(size_t)first], &cmd_[(size_t)first * kCmdWidth], kCmdWidth));
    for (int j = first; j < first + count; j++)
      if (cmd_dirty_[(size_t)j]) {
        cmd_dirty_[(size_t)j] = 0;
        n_cmd_dirty_--;
      }
  }
  void rollback(int slot) {  // the slot as it was before the round's launch
    mrs_throw_on_error(mrs_swarm_copy_uavs(live_, slot, backup_, slot, 1));
    ahead_[(size_t)slot]  = 0;
    cached_[(size_t)slot] = 0;
    n_ahead_--;
    stats_.rollbacks++;
  }
  void grow() {
    const int ncap = cap_ ? 2 * cap_ : 64;  // (multiples of 64: whole blocks of the step kernels)
    if (!live_) {
      mrs_throw_on_error(mrs_swarm_create(ncap, device_, &live_));
      mrs_throw_on_error(mrs_swarm_set_arith(live_, fast_ ? MRS_ARITH_FAST : MRS_ARITH_LITERAL));
      if (!allow_speculation_) speculate_ = false;
    } else {
      for (int k = 0; k < cap_ && n_ahead_ > 0; k++)
        if (ahead_[(size_t)k]) rollback(k);
      flush_pending();
      mrs_swarm_t* bigger = nullptr;
      mrs_throw_on_error(mrs_swarm_clone_resized(live_, ncap, &bigger));
      mrs_swarm_destroy(live_);
      mrs_swarm_destroy(backup_);
      live_   = bigger;
      backup_ = nullptr;
      stats_.grows++;
    }
    mrs_throw_on_error(mrs_swarm_clone(live_, &backup_));  // raw storage for the pre-step copies (never stepped itself)
    for (int k = ncap - 1; k >= cap_; k--) free_.push_back(k);
    used_.resize((size_t)ncap, 0);
    dirty_.resize((size_t)ncap, 0);
    ahead_.resize((size_t)ncap, 0);
    cached_.resize((size_t)ncap, 0);
    round_of_.resize((size_t)ncap, 0);
    cache_.resize((size_t)ncap);
    mode_.resize((size_t)ncap, MRS_INPUT_UNKNOWN);
    cmd_dirty_.resize((size_t)ncap, 0);
    cmd_.resize((size_t)ncap * kCmdWidth, 0.0);
    force_.resize((size_t)ncap * 3, 0.0);
    cap_ = ncap;
  }

  std::recursive_mutex mtx_;
  mrs_swarm_t *        live_ = nullptr, *backup_ = nullptr;
  int                  device_ = -1, cap_ = 0, n_live_ = 0, n_ahead_ = 0, seen_in_round_ = 0;
  bool                 fast_ = true, speculate_ = true, allow_speculation_ = true;
  unsigned long long   round_ = 1;
  double               round_dt_ = 0.0;
  std::vector<int>     free_;
  std::vector<char>    used_, dirty_, ahead_, cached_;
  std::vector<unsigned long long> round_of_;
  std::vector<mrs_uav_state_t>    cache_;
  static constexpr int kCmdWidth = 10;  // widest setInput payload (Attitude: R[9] + throttle), mrs_swarm.h
  std::vector<int>     mode_;
  std::vector<char>    cmd_dirty_;
  std::vector<double>  cmd_, force_;
  int                  n_cmd_dirty_ = 0, round_rollbacks_ = 0;
  bool                 force_dirty_ = false;
  Stats                stats_;
};

// the UavSystem method set, bound to UAV `i_` of swarm `s_` (shared by UavSystem and UavSwarm::Ref); pool_ != nullptr: a pooled
// UavSystem — slot i_ of the pool's swarm, every call under the pool's lock, state getters served from the round's download
class UavSystemApi {
  struct Access {  // the swarm to call, held exclusively while the caller works on it
    std::unique_lock<std::recursive_mutex> lock;
    mrs_swarm_t*                           s;
  };
  Access access(bool writes) {
    if (!pool_) return Access{std::unique_lock<std::recursive_mutex>(), s_};
    Access a{std::unique_lock<std::recursive_mutex>(pool_->mutex()), nullptr};
    pool_->touch(i_, writes);
    a.s = pool_->live();
    return a;
  }
  // the round's downloaded state of a pooled object (nullptr: not pooled, or not current)
  const mrs_uav_state_t* cached_state(std::unique_lock<std::recursive_mutex>& lk) {
    if (!pool_) return nullptr;
    lk = std::unique_lock<std::recursive_mutex>(pool_->mutex());
    return pool_->cached(i_);
  }

public:
  enum INPUT_MODE  // uav_system.hpp:19-32
  {
    INPUT_UNKNOWN,
    ACTUATOR_CMD,
    CONTROL_GROUP_CMD,
    ATTITUDE_RATE_CMD,
    ATTITUDE_CMD,
    TILT_HDG_RATE_CMD,
    ACCELERATION_HDG_RATE_CMD,
    ACCELERATION_HDG_CMD,
    VELOCITY_HDG_RATE_CMD,
    VELOCITY_HDG_CMD,
    POSITION_CMD,
  };

  // uav_system.hpp:304.  On an owning UavSystem (a swarm of one) this is the reference call.  On a view of a larger swarm
  // (UavSwarm::Ref) it is refused: the reference's loop `for (i) uavs_[i]->makeStep(dt)` (src/multirotor_simulator.cpp:211-213)
  // would otherwise step the WHOLE swarm once per UAV — call UavSwarm::makeStep(dt) once instead.
  void makeStep(const double dt) {
    if (pool_) {  // a pooled UavSystem: the round's launch may have stepped this slot already (UavPool::makeStep)
      pool_->makeStep(i_, dt);
      return;
    }
    int32_t n = 0;
    mrs_throw_on_error(mrs_swarm_size(s_, &n));
    if (n != 1) throw std::logic_error("UavSwarm::Ref::makeStep steps every UAV of the swarm: call UavSwarm::makeStep(dt) once per tick instead");
    mrs_throw_on_error(mrs_swarm_step(s_, dt));
  }

  void crash(void) {  // :278
    Access a = access(true);
    mrs_throw_on_error(mrs_swarm_crash(a.s, i_, 1));
  }
  bool hasCrashed(void) {  // :286
    {
      std::unique_lock<std::recursive_mutex> lk;
      if (const mrs_uav_state_t* c = cached_state(lk)) return c->crashed != 0;
    }
    Access  a = access(false);
    int32_t c = 0;
    mrs_throw_on_error(mrs_swarm_has_crashed(a.s, i_, 1, &c));
    return c != 0;
  }

  void applyForce(const Eigen::Vector3d& force) {  // :295
    const double f[3] = {force(0), force(1), force(2)};
    if (pool_) {
      pool_->applyForce(i_, f);
      return;
    }
    mrs_throw_on_error(mrs_swarm_apply_force(s_, i_, 1, f));
  }

  // ---- setInput x11, uav_system.hpp:175-248 ----
  void setInput(const reference::Actuators& cmd) {
    double p[MRS_MAX_MOTORS] = {0};
    for (int m = 0; m < cmd.motors.size() && m < MRS_MAX_MOTORS; m++) p[m] = cmd.motors(m);
    input(MRS_ACTUATOR_CMD, p, MRS_MAX_MOTORS);
  }
  void setInput(const reference::ControlGroup& cmd) {
    const double p[4] = {cmd.roll, cmd.pitch, cmd.yaw, cmd.throttle};
    input(MRS_CONTROL_GROUP_CMD, p, 4);
  }
  void setInput(const reference::AttitudeRate& cmd) {
    const double p[4] = {cmd.rate_x, cmd.rate_y, cmd.rate_z, cmd.throttle};
    input(MRS_ATTITUDE_RATE_CMD, p, 4);
  }
  void setInput(const reference::Attitude& cmd) {
    double p[10];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) p[r * 3 + c] = cmd.orientation(r, c);
    p[9] = cmd.throttle;
    input(MRS_ATTITUDE_CMD, p, 10);
  }
  void setInput(const reference::TiltHdgRate& cmd) {
    const double p[5] = {cmd.tilt_vector(0), cmd.tilt_vector(1), cmd.tilt_vector(2), cmd.heading_rate, cmd.throttle};
    input(MRS_TILT_HDG_RATE_CMD, p, 5);
  }
  void setInput(const reference::AccelerationHdgRate& cmd) { input4(MRS_ACCELERATION_HDG_RATE_CMD, cmd.acceleration, cmd.heading_rate); }
  void setInput(const reference::AccelerationHdg& cmd) { input4(MRS_ACCELERATION_HDG_CMD, cmd.acceleration, cmd.heading); }
  void setInput(const reference::VelocityHdgRate& cmd) { input4(MRS_VELOCITY_HDG_RATE_CMD, cmd.velocity, cmd.heading_rate); }
  void setInput(const reference::VelocityHdg& cmd) { input4(MRS_VELOCITY_HDG_CMD, cmd.velocity, cmd.heading); }
  void setInput(const reference::Position& cmd) { input4(MRS_POSITION_CMD, cmd.position, cmd.heading); }
  void setInput(void) { input(MRS_INPUT_UNKNOWN, nullptr, 0); }

  // ---- setFeedforward x4, uav_system.hpp:254-272 ----
  void setFeedforward(const reference::AccelerationHdgRate& cmd) { ff(MRS_FF_ACCELERATION_HDG_RATE, cmd.acceleration, cmd.heading_rate); }
  void setFeedforward(const reference::AccelerationHdg& cmd) { ff(MRS_FF_ACCELERATION_HDG, cmd.acceleration, cmd.heading); }
  void setFeedforward(const reference::VelocityHdg& cmd) { ff(MRS_FF_VELOCITY_HDG, cmd.velocity, cmd.heading); }
  void setFeedforward(const reference::VelocityHdgRate& cmd) { ff(MRS_FF_VELOCITY_HDG_RATE, cmd.velocity, cmd.heading_rate); }

  MultirotorModel::State getState(void) {  // :386
    mrs_uav_state_t rec;
    {
      std::unique_lock<std::recursive_mutex> lk;
      if (const mrs_uav_state_t* c = cached_state(lk)) {
        rec = *c;  // (a pooled object inside a round: the round's ONE download)
      } else {
        lk = std::unique_lock<std::recursive_mutex>();
        Access a = access(false);
        mrs_throw_on_error(mrs_swarm_get_states(a.s, i_, 1, &rec));
      }
    }
    MultirotorModel::State st;
    st.x = Eigen::Vector3d(rec.x[0], rec.x[1], rec.x[2]);
    st.v = Eigen::Vector3d(rec.v[0], rec.v[1], rec.v[2]);
    st.v_prev = Eigen::Vector3d(rec.v_prev[0], rec.v_prev[1], rec.v_prev[2]);
    st.omega  = Eigen::Vector3d(rec.omega[0], rec.omega[1], rec.omega[2]);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) st.R(r, c) = rec.R[r * 3 + c];
    st.motor_rpm = Eigen::VectorXd::Zero(rec.n_motors);
    for (int m = 0; m < rec.n_motors; m++) st.motor_rpm(m) = rec.motor_rpm[m];
    return st;
  }

  // UavSystemRos::getPose (include/mrs_multirotor_simulator/uav_system_ros.h:48, src/uav_system_ros.cpp:289-292): the position
  Eigen::Vector3d getPose(void) {
    {
      std::unique_lock<std::recursive_mutex> lk;
      if (const mrs_uav_state_t* c = cached_state(lk)) return Eigen::Vector3d(c->x[0], c->x[1], c->x[2]);
    }
    Access a = access(false);
    double x[3];
    mrs_throw_on_error(mrs_swarm_get_state(a.s, i_, 1, x, nullptr, nullptr, nullptr, nullptr, nullptr));
    return Eigen::Vector3d(x[0], x[1], x[2]);
  }

  MultirotorModel::ModelParams getParams(void) {  // :395
    Access             a = access(false);
    mrs_model_params_t c;
    mrs_throw_on_error(mrs_swarm_get_params(a.s, i_, &c));
    MultirotorModel::ModelParams p;
    p.fromC(c);
    return p;
  }

  void setParams(const MultirotorModel::ModelParams& params) {  // :404 (controllers fall back to default gains)
    Access                   a = access(true);
    const mrs_model_params_t c = params.toC();
    mrs_throw_on_error(mrs_swarm_set_params(a.s, i_, 1, &c));
  }

  Eigen::Vector3d getImuAcceleration(void) {  // :424
    {
      std::unique_lock<std::recursive_mutex> lk;
      if (const mrs_uav_state_t* c = cached_state(lk)) return Eigen::Vector3d(c->imu_acceleration[0], c->imu_acceleration[1], c->imu_acceleration[2]);
    }
    Access a = access(false);
    double v[3];
    mrs_throw_on_error(mrs_swarm_get_imu(a.s, i_, 1, v));
    return Eigen::Vector3d(v[0], v[1], v[2]);
  }

  void setMixerParams(const Mixer::Params& params) {  // :433-451
    Access                   a = access(true);
    const mrs_mixer_params_t c{params.desaturation ? 1 : 0, 0};
    mrs_throw_on_error(mrs_swarm_set_mixer_params(a.s, i_, 1, &c));
  }
  void setRateControllerParams(const RateController::Params& params) {
    Access                  a = access(true);
    const mrs_rate_params_t c{params.kp, params.kd, params.ki};
    mrs_throw_on_error(mrs_swarm_set_rate_params(a.s, i_, 1, &c));
  }
  void setAttitudeControllerParams(const AttitudeController::Params& params) {
    Access                      a = access(true);
    const mrs_attitude_params_t c{params.kp, params.kd, params.ki, params.max_rate_roll_pitch, params.max_rate_yaw};
    mrs_throw_on_error(mrs_swarm_set_attitude_params(a.s, i_, 1, &c));
  }
  void setVelocityControllerParams(const VelocityController::Params& params) {
    Access                      a = access(true);
    const mrs_velocity_params_t c{params.kp, params.kd, params.ki, params.max_acceleration};
    mrs_throw_on_error(mrs_swarm_set_velocity_params(a.s, i_, 1, &c));
  }
  void setPositionControllerParams(const PositionController::Params& params) {
    Access                      a = access(true);
    const mrs_position_params_t c{params.kp, params.kd, params.ki, params.max_velocity};
    mrs_throw_on_error(mrs_swarm_set_position_params(a.s, i_, 1, &c));
  }

  Eigen::MatrixXd getMixerAllocation(void) {  // :415
    Access             acc = access(false);
    mrs_swarm_t* const s_  = acc.s;
    mrs_model_params_t p;
    mrs_throw_on_error(mrs_swarm_get_params(s_, i_, &p));
    double a[MRS_MAX_MOTORS * 4];
    mrs_throw_on_error(mrs_swarm_get_mixer_allocation(s_, i_, a));
    Eigen::MatrixXd m = Eigen::MatrixXd::Zero(p.n_motors, 4);
    for (int r = 0; r < p.n_motors; r++)
      for (int c = 0; c < 4; c++) m(r, c) = a[r * 4 + c];
    return m;
  }

protected:
  UavSystemApi(mrs_swarm_t* s, int i) : s_(s), i_(i) {}
  mrs_swarm_t* s_;
  int          i_;
  UavPool*     pool_ = nullptr;

private:
  void input(int mode, const double* p, int n) {
    if (pool_) {
      pool_->setInput(i_, mode, p, n);
      return;
    }
    mrs_throw_on_error(mrs_swarm_set_input(s_, i_, 1, mode, p, n));
  }
  void input4(int mode, const Eigen::Vector3d& v, double h) {
    const double p[4] = {v(0), v(1), v(2), h};
    input(mode, p, 4);
  }
  void ff(int kind, const Eigen::Vector3d& v, double h) {
    Access       a    = access(true);
    const double p[4] = {v(0), v(1), v(2), h};
    mrs_throw_on_error(mrs_swarm_set_feedforward(a.s, i_, 1, kind, p, 4));
  }
};

// ------------------------------------------------------------------------------------------------------------------
// the batch owner
// ------------------------------------------------------------------------------------------------------------------
class UavSwarm {
public:
  class Ref : public UavSystemApi {  // view of one UAV (non-owning); makeStep() is refused on it, see UavSystemApi::makeStep
  public:
    Ref(mrs_swarm_t* s, int i) : UavSystemApi(s, i) {}
  };

  explicit UavSwarm(int n_uavs, int device_id = -1, bool fast_arithmetic = true) : s_(nullptr) {
    mrs_throw_on_error(mrs_swarm_create(n_uavs, device_id, &s_));
    mrs_throw_on_error(mrs_swarm_set_arith(s_, fast_arithmetic ? MRS_ARITH_FAST : MRS_ARITH_LITERAL));
  }
  explicit UavSwarm(mrs_swarm_t* adopt) : s_(adopt) {}  // takes ownership of a library swarm (e.g. mrs_swarm_clone)
  ~UavSwarm() { mrs_swarm_destroy(s_); }
  UavSwarm(const UavSwarm&) = delete;
  UavSwarm& operator=(const UavSwarm&) = delete;

  int size() const {
    int32_t n = 0;
    mrs_swarm_size(s_, &n);
    return n;
  }
  mrs_swarm_t* handle() { return s_; }
  Ref          operator[](int i) { return Ref(s_, i); }
  Ref          at(int i) { return Ref(s_, i); }

  // UavSystem(model_params, spawn_pos, spawn_heading) for UAVs [first, first+count) — uav_system.hpp:144-153
  void construct(int first, int count, const MultirotorModel::ModelParams& params, const std::vector<Eigen::Vector3d>& spawn_pos,
                 const std::vector<double>& spawn_heading) {
    const mrs_model_params_t c = params.toC();
    std::vector<double>      p((size_t)count * 3);
    for (int k = 0; k < count; k++)
      for (int j = 0; j < 3; j++) p[(size_t)k * 3 + j] = spawn_pos[(size_t)k](j);
    mrs_throw_on_error(mrs_swarm_construct(s_, first, count, &c, p.data(), spawn_heading.data()));
  }

  void makeStep(double dt) { mrs_throw_on_error(mrs_swarm_step(s_, dt)); }  // src/multirotor_simulator.cpp:211-213
  void makeSteps(double dt, int n_steps, int substeps_per_launch = 1) { mrs_throw_on_error(mrs_swarm_step_n(s_, dt, n_steps, substeps_per_launch)); }
  void handleCollisions(bool enabled, bool crash, double rebounce) {  // src/multirotor_simulator.cpp:295-359
    mrs_throw_on_error(mrs_swarm_handle_collisions(s_, enabled, crash, rebounce));
  }
  void tick(double dt, int n_ticks, bool enabled, bool crash, double rebounce) {  // timerMain order, :211-217
    mrs_throw_on_error(mrs_swarm_tick_n(s_, dt, n_ticks, enabled, crash, rebounce));
  }
  // ---- one shard per GPU/process: this swarm holds the UAVs of `rank` out of n_total, collisions act across all shards ----
  // rank 0 creates the id and hands it to the others (any host channel); librccl_path = nullptr loads the system librccl.so
  static std::array<uint8_t, 128> commUniqueId(const char* librccl_path = nullptr) {
    std::array<uint8_t, 128> id{};
    mrs_throw_on_error(mrs_rccl_unique_id(librccl_path, id.data()));
    return id;
  }
  void commInit(int world, int rank, const std::array<uint8_t, 128>& id, int64_t n_total, const char* librccl_path = nullptr) {
    mrs_throw_on_error(mrs_swarm_comm_init(s_, librccl_path, world, rank, id.data(), n_total));
  }
  // timerMain x n_ticks on every rank: makeStep, all-gather of the 48-B records on the swarm's stream, handleCollisions over all UAVs
  void tickSharded(double dt, int n_ticks, bool enabled, bool crash, double rebounce) {
    mrs_throw_on_error(mrs_swarm_tick_sharded_n(s_, dt, n_ticks, enabled, crash, rebounce));
  }
  // the same over an in-process group (several UavSwarm objects of one process, one host thread each: multi-device hosts without
  // RCCL, virtual shards on one device) or over a caller-supplied all-gather (MPI, ...)
  void commInitLoopback(mrs_loopback_group_t* group, int rank, int64_t n_total) { mrs_throw_on_error(mrs_swarm_comm_init_loopback(s_, group, rank, n_total)); }
  void commInitCustom(int world, int rank, int64_t n_total, mrs_allgather_fn fn, void* user) {
    mrs_throw_on_error(mrs_swarm_comm_init_custom(s_, world, rank, n_total, fn, user));
  }
  void commDestroy() { mrs_throw_on_error(mrs_swarm_comm_destroy(s_)); }
  // MRS_EXCHANGE_EXPORT_SETS (default: boundary UAVs only between two neighbour searches) or MRS_EXCHANGE_FULL_GATHER
  void setExchange(int exchange) { mrs_throw_on_error(mrs_swarm_set_exchange(s_, exchange)); }
  mrs_comm_info_t commInfo() {
    mrs_comm_info_t ci;
    mrs_throw_on_error(mrs_swarm_comm_info(s_, &ci));
    return ci;
  }
  // neighbour searches of the export-set exchange on this rank: all, the ones that exchanged halos instead of all records, the halo
  // searches repeated on all records, entries per rank of the next halo block (mrs_swarm_get_search_stats)
  struct SearchStats {
    int64_t searches, halo_searches, halo_repeats, halo_capacity;
  };
  SearchStats searchStats() {
    SearchStats st{};
    mrs_throw_on_error(mrs_swarm_get_search_stats(s_, &st.searches, &st.halo_searches, &st.halo_repeats, &st.halo_capacity));
    return st;
  }
  // spatially coherent shards for a swarm addressed by public index: order[k] = public index at position k of the x-sorted order;
  // rank r of `world` holds order[lo_r, hi_r) with equal-count ranges (the first n % world ranks one more)
  static std::vector<int64_t> slabPartition(const std::vector<Eigen::Vector3d>& pos, int world) {
    std::vector<double> p(pos.size() * 3);
    for (size_t k = 0; k < pos.size(); k++)
      for (int j = 0; j < 3; j++) p[k * 3 + (size_t)j] = pos[k](j);
    std::vector<int64_t> order(pos.size());
    mrs_throw_on_error(mrs_slab_partition(p.data(), (int64_t)pos.size(), world, order.data()));
    return order;
  }
  // a spawn order that follows space, for callers free to choose which UAV gets which index (mrs_cell_order): order[k] = the caller's
  // index of the UAV to spawn k-th
  static std::vector<int64_t> cellOrder(const std::vector<Eigen::Vector3d>& pos, double cell = 0.0) {
    std::vector<double> p(pos.size() * 3);
    for (size_t k = 0; k < pos.size(); k++)
      for (int j = 0; j < 3; j++) p[k * 3 + (size_t)j] = pos[k](j);
    std::vector<int64_t> order(pos.size());
    mrs_throw_on_error(mrs_cell_order(p.data(), (int64_t)pos.size(), cell, order.data()));
    return order;
  }
  void synchronize() { mrs_throw_on_error(mrs_swarm_synchronize(s_)); }
  // collision ticks so far, and how many of them had to repeat the neighbour search
  std::pair<int64_t, int64_t> collisionStats() {
    int64_t t = 0, r = 0;
    mrs_throw_on_error(mrs_swarm_get_collision_stats(s_, &t, &r));
    return {t, r};
  }

  // ---- UavSystemRos semantics that live on the device ----
  // timeoutInput() for UAVs [first, first+count): safe command of the same mode (src/uav_system_ros.cpp:474-647)
  void timeoutInput(int first, int count) { mrs_throw_on_error(mrs_swarm_timeout_input(s_, first, count)); }
  // UavSystemRos::makeStep iterates the model only `if (_iterate_without_input_ || time_last_input_ > 0)` (:265): UAVs on hold
  // are skipped by makeStep / tick
  void setHold(int first, int count, bool hold) { mrs_throw_on_error(mrs_swarm_set_hold(s_, first, count, hold ? 1 : 0)); }
  // callbackSetMass / callbackSetGroundZ (src/uav_system_ros.cpp:1028-1080)
  void setMass(int first, int count, double mass) { mrs_throw_on_error(mrs_swarm_set_mass(s_, first, count, mass)); }
  void setGroundZ(int first, int count, double ground_z) { mrs_throw_on_error(mrs_swarm_set_ground_z(s_, first, count, ground_z)); }
  // what publishOdometry / publishIMU / publishRangefinder / publishPoses need, one packed download
  std::vector<mrs_uav_output_t> getOutputs(int first, int count) {
    std::vector<mrs_uav_output_t> out((size_t)count);
    mrs_throw_on_error(mrs_swarm_get_outputs(s_, first, count, out.data()));
    return out;
  }
  // the same without the host copy: the pointer stays valid until the next getOutputs* call
  const mrs_uav_output_t* getOutputsView(int first, int count) {
    const mrs_uav_output_t* v = nullptr;
    mrs_throw_on_error(mrs_swarm_get_outputs_view(s_, first, count, &v));
    return v;
  }
  // the same PIPELINED with the steps (mrs_swarm_get_outputs_async / mrs_swarm_outputs_wait): start the download behind the steps
  // queued so far, wait for it later — after the next tick has been queued — and publish while that tick runs
  int getOutputsAsync(int first, int count) {
    int32_t ticket = -1;
    mrs_throw_on_error(mrs_swarm_get_outputs_async(s_, first, count, &ticket));
    return ticket;
  }
  const mrs_uav_output_t* outputsWait(int ticket, int* count = nullptr) {
    const mrs_uav_output_t* v = nullptr;
    int32_t                 c = 0;
    mrs_throw_on_error(mrs_swarm_outputs_wait(s_, ticket, &v, &c));
    if (count) *count = c;
    return v;
  }
  // batched subscriber side: pinned rows to fill with setInput payloads (layout of mrs_swarm_set_input), then one commit
  double* inputStaging(int count, int stride) {
    double* rows = nullptr;
    mrs_throw_on_error(mrs_swarm_input_staging(s_, count, stride, &rows));
    return rows;
  }
  void commitInput(int first, int count, int mode, int stride) { mrs_throw_on_error(mrs_swarm_commit_input(s_, first, count, mode, stride)); }
  // the tail of the UavSystemRos constructor (:223-232) for the whole swarm: zero actuators, two makeStep(0.01)
  void warmUp() {
    std::vector<double> zeros((size_t)size() * MRS_MAX_MOTORS, 0.0);
    mrs_throw_on_error(mrs_swarm_set_input(s_, 0, size(), MRS_ACTUATOR_CMD, zeros.data(), MRS_MAX_MOTORS));
    mrs_throw_on_error(mrs_swarm_step_n(s_, 0.01, 2, 1));
  }

  // getPose() of every UAV (src/uav_system_ros.cpp:289), n x 3 row-major
  std::vector<double> getPoses() {
    std::vector<double> x((size_t)size() * 3);
    mrs_throw_on_error(mrs_swarm_get_state(s_, 0, size(), x.data(), nullptr, nullptr, nullptr, nullptr, nullptr));
    return x;
  }

private:
  mrs_swarm_t* s_;
};

// ------------------------------------------------------------------------------------------------------------------
// the reference's class: one slot of the process-wide pool
// ------------------------------------------------------------------------------------------------------------------
class UavSystem : public UavSystemApi {
public:
  UavSystem(void) : UavSystemApi(nullptr, -1) {  // :127
    pool_ = &UavPool::instance();
    i_    = pool_->acquire();
  }

  UavSystem(const MultirotorModel::ModelParams& model_params) : UavSystem() {  // :135
    const mrs_model_params_t c = model_params.toC();
    std::lock_guard<std::recursive_mutex> lk(pool_->mutex());
    pool_->touch(i_, true);
    mrs_throw_on_error(mrs_swarm_construct(pool_->live(), i_, 1, &c, nullptr, nullptr));
  }

  UavSystem(const MultirotorModel::ModelParams& model_params, const Eigen::Vector3d spawn_pos, const double spawn_heading) : UavSystem() {  // :144
    const mrs_model_params_t c    = model_params.toC();
    const double             p[3] = {spawn_pos(0), spawn_pos(1), spawn_pos(2)};
    std::lock_guard<std::recursive_mutex> lk(pool_->mutex());
    pool_->touch(i_, true);
    mrs_throw_on_error(mrs_swarm_construct(pool_->live(), i_, 1, &c, p, &spawn_heading));
  }

  // the reference object is a copy-assignable value (uav_system_ = UavSystem(...), src/uav_system_ros.cpp:105): a copy is an
  // independent UAV with the same state, command, feed-forwards, PIDs and parameters (mrs_swarm_copy_uavs inside the pool); moves
  // hand the slot over
  UavSystem(const UavSystem& o) : UavSystem() { pool_->copy(i_, o.i_); }
  UavSystem& operator=(const UavSystem& o) {
    if (this != &o) pool_->copy(i_, o.i_);
    return *this;
  }
  UavSystem(UavSystem&& o) noexcept : UavSystemApi(nullptr, o.i_) {
    pool_ = o.pool_;
    o.i_  = -1;
  }
  UavSystem& operator=(UavSystem&& o) noexcept {
    if (this != &o) {
      if (i_ >= 0) pool_->release(i_);
      i_   = o.i_;
      o.i_ = -1;
    }
    return *this;
  }
  ~UavSystem() {
    if (i_ >= 0 && pool_) pool_->release(i_);
  }

  int poolSlot() const { return i_; }
};

}  // namespace mrs_multirotor_simulator

#endif  // UAV_SYSTEM_H
