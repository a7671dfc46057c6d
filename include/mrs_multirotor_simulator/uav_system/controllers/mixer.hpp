// Mixer::Params of the reference (controllers/mixer.hpp:14-17); the mixing itself runs in the GPU step kernel.
#ifndef MRS_MIXER_HPP
#define MRS_MIXER_HPP
#include "../multirotor_model.hpp"
namespace mrs_multirotor_simulator
{
class Mixer {
public:
  class Params {
  public:
    bool desaturation = true;
  };
};
}  // namespace mrs_multirotor_simulator
#endif
