// Mixer of the reference (controllers/mixer.hpp:10-37): Params, allocation matrix and getControlSignal — evaluated on the GPU by the
// mixer device function of the step kernels (mrs_swarm_debug_component, MRS_COMP_MIXER).
#ifndef MRS_MIXER_HPP
#define MRS_MIXER_HPP
#include "controller_probe.hpp"
namespace mrs_multirotor_simulator
{
class Mixer {
public:
  class Params {
  public:
    bool desaturation = true;
  };

  Mixer() {}                                                                            // mixer.hpp:47-49
  Mixer(const MultirotorModel::ModelParams& model_params) : probe_(model_params) {}     // :51-55 (calculateAllocation on the host side of the library)

  void setParams(const Params& params) {  // :61-66
    const mrs_mixer_params_t c{params.desaturation ? 1 : 0, 0};
    mrs_throw_on_error(mrs_swarm_set_mixer_params(probe_.handle(), 0, 1, &c));
  }

  reference::Actuators getControlSignal(const reference::ControlGroup& reference) {  // :107-144
    const double in[4] = {reference.roll, reference.pitch, reference.yaw, reference.throttle};
    double       out[MRS_MAX_MOTORS];
    probe_.run(MRS_COMP_MIXER, in, 4, out, MRS_MAX_MOTORS, 0.001);
    mrs_model_params_t p;
    mrs_throw_on_error(mrs_swarm_get_params(probe_.handle(), 0, &p));
    reference::Actuators a;
    a.motors = Eigen::VectorXd::Zero(p.n_motors);
    for (int m = 0; m < p.n_motors; m++) a.motors(m) = out[m];
    return a;
  }

  Eigen::MatrixXd getAllocationMatrix(void) {  // :150-152 (n_motors x 4)
    mrs_model_params_t p;
    mrs_throw_on_error(mrs_swarm_get_params(probe_.handle(), 0, &p));
    double a[MRS_MAX_MOTORS * 4];
    mrs_throw_on_error(mrs_swarm_get_mixer_allocation(probe_.handle(), 0, a));
    Eigen::MatrixXd m = Eigen::MatrixXd::Zero(p.n_motors, 4);
    for (int r = 0; r < p.n_motors; r++)
      for (int c = 0; c < 4; c++) m(r, c) = a[r * 4 + c];
    return m;
  }

private:
  detail::ControllerProbe probe_;
};
}  // namespace mrs_multirotor_simulator
#endif
