// l0_classes_test.cpp — the stand-alone L0 classes of the header API (MultirotorModel, the controllers, PIDController) and the value
// semantics of UavSystem, used the way code written against the reference's headers uses them
// (include/mrs_multirotor_simulator/uav_system/multirotor_model.hpp:100-131, controllers/*.hpp, src/uav_system_ros.cpp:105).
// Prints rows that tests/test_l0_classes.py compares with the oracle.
#include <cstdio>
#include <mrs_multirotor_simulator/uav_system/uav_system.hpp>

using namespace mrs_multirotor_simulator;

static void row(const char* tag, const double* v, int n) {
  std::printf("%s", tag);
  for (int i = 0; i < n; i++) std::printf(" %.17g", v[i]);
  std::printf("\n");
}
static void print_state(const char* tag, const MultirotorModel::State& st, const Eigen::Vector3d& imu) {
  double v[32];
  int    k = 0;
  for (int i = 0; i < 3; i++) v[k++] = st.x(i);
  for (int i = 0; i < 3; i++) v[k++] = st.v(i);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) v[k++] = st.R(r, c);
  for (int i = 0; i < 3; i++) v[k++] = st.omega(i);
  for (int i = 0; i < (int)st.motor_rpm.size(); i++) v[k++] = st.motor_rpm(i);
  for (int i = 0; i < 3; i++) v[k++] = imu(i);
  row(tag, v, k);
}

int main() {
  MultirotorModel::ModelParams params;  // x500
  params.ground_enabled        = true;
  params.ground_z              = 0.0;
  params.takeoff_patch_enabled = false;

  // ---- MultirotorModel on its own ----
  MultirotorModel model(params, Eigen::Vector3d(1.0, -2.0, 3.0), 0.7);
  reference::Actuators act;
  act.motors = Eigen::VectorXd::Zero(params.n_motors);
  for (int m = 0; m < params.n_motors; m++) act.motors(m) = 0.45 + 0.02 * m;
  model.setInput(act);
  for (int k = 0; k < 200; k++) model.step(0.001);
  print_state("MODEL200", model.getState(), model.getImuAcceleration());
  model.applyForce(Eigen::Vector3d(0.5, -1.0, 2.0));
  MultirotorModel copy = model;  // value semantics: an independent model from here on
  for (int k = 0; k < 50; k++) model.step(0.001);
  print_state("MODEL250", model.getState(), model.getImuAcceleration());
  for (int k = 0; k < 50; k++) copy.step(0.001);
  print_state("COPY250", copy.getState(), copy.getImuAcceleration());
  model.setStatePos(Eigen::Vector3d(5.0, 6.0, 7.0), -1.1);  // x, R and the spawn height only
  print_state("SETPOS", model.getState(), model.getImuAcceleration());
  {
    MultirotorModel::State st = copy.getState();
    st.v     = Eigen::Vector3d(1.0, 2.0, -0.5);
    st.omega = Eigen::Vector3d(0.1, -0.2, 0.3);
    copy.setState(st);
    copy.step(0.001);
    print_state("SETSTATE", copy.getState(), copy.getImuAcceleration());  // IMU shows that v_prev was left alone by setState
    const Eigen::Vector3d& f = copy.getExternalForce();
    const double fv[3] = {f(0), f(1), f(2)};
    row("FEXT", fv, 3);
    MultirotorModel::InternalState y, dy;
    const MultirotorModel::State&  s2 = copy.getState();
    for (int i = 0; i < 3; i++) {
      y[i] = s2.x(i); y[3 + i] = s2.v(i); y[15 + i] = s2.omega(i);
      y[6 + i] = s2.R(i, 0); y[9 + i] = s2.R(i, 1); y[12 + i] = s2.R(i, 2);
    }
    copy(y, dy, 0.0);
    row("RHS", dy.data(), 18);
    bool refused = false;
    try {
      copy.setExternalMoment(Eigen::Vector3d(0.0, 0.1, 0.0));
    }
    catch (const std::runtime_error&) {
      refused = true;
    }
    std::printf("MOMENT_REFUSED %d\n", (int)refused);
  }

  // ---- the controllers, one at a time, on a state handed in ----
  MultirotorModel::State st;
  st.x = Eigen::Vector3d(0.3, -0.2, 4.0);
  st.v = Eigen::Vector3d(0.5, 0.1, -0.3);
  st.v_prev = st.v;
  st.omega  = Eigen::Vector3d(0.05, -0.1, 0.2);
  {  // a tilted attitude
    const double c = 0.9553364891256060, s = 0.2955202066613396;  // cos / sin 0.3
    st.R = Eigen::Matrix3d::Identity();
    st.R(1, 1) = c; st.R(1, 2) = -s; st.R(2, 1) = s; st.R(2, 2) = c;
  }
  st.motor_rpm = Eigen::VectorXd::Zero(params.n_motors);
  PositionController     pos_c(params);
  VelocityController     vel_c(params);
  AccelerationController acc_c(params);
  AttitudeController     att_c(params);
  RateController         rate_c(params);
  Mixer                  mixer(params);
  pos_c.setParams(PositionController::Params());
  vel_c.setParams(VelocityController::Params());
  att_c.setParams(AttitudeController::Params());
  rate_c.setParams(RateController::Params());
  mixer.setParams(Mixer::Params());
  reference::Position pref;
  pref.position = Eigen::Vector3d(2.0, 1.0, 6.0);
  pref.heading  = 0.4;
  for (int call = 0; call < 2; call++) {  // twice: the PIDs remember
    const reference::VelocityHdg     v  = pos_c.getControlSignal(st, pref, 0.01);
    const reference::AccelerationHdg a  = vel_c.getControlSignal(st, v, 0.01);
    const reference::Attitude        at = acc_c.getControlSignal(st, a, 0.01);
    const reference::AttitudeRate    ar = att_c.getControlSignal(st, at, 0.01);
    const reference::ControlGroup    cg = rate_c.getControlSignal(st, ar, 0.01);
    const reference::Actuators       m  = mixer.getControlSignal(cg);
    double out[64];
    int    k = 0;
    for (int i = 0; i < 3; i++) out[k++] = v.velocity(i);
    out[k++] = v.heading;
    for (int i = 0; i < 3; i++) out[k++] = a.acceleration(i);
    out[k++] = a.heading;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) out[k++] = at.orientation(r, c);
    out[k++] = at.throttle;
    out[k++] = ar.rate_x; out[k++] = ar.rate_y; out[k++] = ar.rate_z; out[k++] = ar.throttle;
    out[k++] = cg.roll; out[k++] = cg.pitch; out[k++] = cg.yaw; out[k++] = cg.throttle;
    for (int i = 0; i < (int)m.motors.size(); i++) out[k++] = m.motors(i);
    row(call == 0 ? "CASCADE0" : "CASCADE1", out, k);
  }
  {  // the heading-rate branch of the cascade
    const reference::AccelerationHdgRate a(Eigen::Vector3d(0.5, -0.4, 1.0), 0.3);
    const reference::TiltHdgRate         t  = acc_c.getControlSignal(st, a, 0.01);
    const reference::AttitudeRate        ar = att_c.getControlSignal(st, t, 0.01);
    const double out[9] = {t.tilt_vector(0), t.tilt_vector(1), t.tilt_vector(2), t.heading_rate, t.throttle, ar.rate_x, ar.rate_y, ar.rate_z, ar.throttle};
    row("TILT", out, 9);
    const Eigen::MatrixXd A = mixer.getAllocationMatrix();
    std::printf("MIXALLOC %d %d %.17g\n", (int)A.rows(), (int)A.cols(), A(0, 0));
  }

  // ---- PIDController ----
  {
    PIDController pid;
    pid.setParams(2.0, 0.15, 0.2, 6.0, 1.0);
    const double errs[6] = {0.5, 0.4, 5.0, -7.0, 0.1, 0.05};
    double       out[6];
    for (int k = 0; k < 6; k++) {
      if (k == 4) pid.setSaturation(0.15);
      out[k] = pid.update(errs[k], 0.01);
    }
    row("PID", out, 6);
    pid.reset();
    const double o2 = pid.update(0.25, 0.01);
    row("PIDRESET", &o2, 1);
  }

  // ---- UavSystem is a copy-assignable value (src/uav_system_ros.cpp:105) ----
  {
    UavSystem a(params, Eigen::Vector3d(0, 0, 2), 0.0);
    reference::Position cmd;
    cmd.position = Eigen::Vector3d(1, 1, 3);
    a.setInput(cmd);
    for (int k = 0; k < 100; k++) a.makeStep(0.001);
    UavSystem b;
    b = a;            // copy assignment: PIDs, command, state travel
    UavSystem c(a);   // copy construction
    for (int k = 0; k < 100; k++) {
      a.makeStep(0.001);
      b.makeStep(0.001);
    }
    print_state("UAV_A", a.getState(), a.getImuAcceleration());
    print_state("UAV_B", b.getState(), b.getImuAcceleration());
    print_state("UAV_C", c.getState(), c.getImuAcceleration());  // still at step 100
    UavSwarm swarm(3, -1, false);
    bool refused = false;
    try {
      swarm[1].makeStep(0.001);  // the reference's per-UAV loop on views of a batch: refused, not N x stepping
    }
    catch (const std::logic_error&) {
      refused = true;
    }
    std::printf("REF_MAKESTEP_REFUSED %d\n", (int)refused);
  }
  return 0;
}
