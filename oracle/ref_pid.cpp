// ref_pid.cpp — C wrapper around the REFERENCE's own PIDController, compiled from the reference header where it lies
// (include/mrs_multirotor_simulator/uav_system/controllers/pid.hpp — it needs nothing but <math.h>) into oracle/_ref/libref_pid.so.
// Test infrastructure: pins row C1 of SURVEY §8(a) — oracle/uav_oracle.c:pid_update and, through it, the GPU kernels — against the
// reference itself.  Nothing of the reference is copied into this repository; without /root/reference the prebuilt library is used.
#include <mrs_multirotor_simulator/uav_system/controllers/pid.hpp>

using mrs_multirotor_simulator::PIDController;

extern "C" {
void* ref_pid_create(void) { return new PIDController(); }
void  ref_pid_destroy(void* p) { delete static_cast<PIDController*>(p); }
void  ref_pid_reset(void* p) { static_cast<PIDController*>(p)->reset(); }
void  ref_pid_set_params(void* p, double kp, double kd, double ki, double saturation, double antiwindup) {
  static_cast<PIDController*>(p)->setParams(kp, kd, ki, saturation, antiwindup);
}
void   ref_pid_set_saturation(void* p, double saturation) { static_cast<PIDController*>(p)->setSaturation(saturation); }
double ref_pid_update(void* p, double error, double dt) { return static_cast<PIDController*>(p)->update(error, dt); }
}
