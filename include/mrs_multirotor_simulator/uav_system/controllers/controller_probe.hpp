// controller_probe.hpp — what the stand-alone controller classes of the facade share: a one-UAV swarm that holds the airframe
// constants, the controller gains and the PID state on the GPU, and runs ONE component of the cascade per call
// (mrs_swarm_debug_component: the very device functions the step kernels are made of).  Copying a controller copies its PID state.
#ifndef MRS_CONTROLLER_PROBE_HPP
#define MRS_CONTROLLER_PROBE_HPP
#include "../multirotor_model.hpp"
namespace mrs_multirotor_simulator
{
namespace detail
{
class ControllerProbe {
public:
  ControllerProbe() : s_(nullptr) { mrs_throw_on_error(mrs_swarm_create(1, -1, &s_)); }
  explicit ControllerProbe(const MultirotorModel::ModelParams& model_params) : ControllerProbe() {
    const mrs_model_params_t c = model_params.toC();
    if (mrs_swarm_construct(s_, 0, 1, &c, nullptr, nullptr) != MRS_OK) {
      mrs_swarm_destroy(s_);
      s_ = nullptr;
      mrs_throw_on_error(MRS_ERR_ARG);
    }
  }
  ~ControllerProbe() { mrs_swarm_destroy(s_); }
  ControllerProbe(const ControllerProbe& o) : s_(nullptr) { mrs_throw_on_error(mrs_swarm_clone(o.s_, &s_)); }
  ControllerProbe& operator=(const ControllerProbe& o) {
    if (this != &o) {
      mrs_swarm_t* c = nullptr;
      mrs_throw_on_error(mrs_swarm_clone(o.s_, &c));
      mrs_swarm_destroy(s_);
      s_ = c;
    }
    return *this;
  }
  mrs_swarm_t* handle() { return s_; }

  // the controllers read the state they are handed (getControlSignal(state, reference, dt)), never a state of their own
  void setState(const MultirotorModel::State& st) {
    double x[3], v[3], R[9], w[3], rpm[MRS_MAX_MOTORS] = {0};
    for (int c = 0; c < 3; c++) {
      x[c] = st.x(c); v[c] = st.v(c); w[c] = st.omega(c);
      for (int q = 0; q < 3; q++) R[c * 3 + q] = st.R(c, q);
    }
    for (int m = 0; m < (int)st.motor_rpm.size() && m < MRS_MAX_MOTORS; m++) rpm[m] = st.motor_rpm(m);
    mrs_throw_on_error(mrs_swarm_set_state(s_, 0, 1, x, v, R, w, rpm));
  }
  void run(int component, const double* in, int n_in, double* out, int n_out, double dt) {
    mrs_throw_on_error(mrs_swarm_debug_component(s_, component, 0, 1, in, n_in, out, n_out, dt));
  }

private:
  mrs_swarm_t* s_;
};
}  // namespace detail
}  // namespace mrs_multirotor_simulator
#endif
