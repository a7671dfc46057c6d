"""Known-answer tests pinning the CPU oracle (oracle/uav_oracle.c) to values derivable from the reference
source alone.  The reference ships no tests or vectors for this path, so these analytic
cases — not reference outputs — are what anchors the dynamics oracle ("parity unpinned")."""
import math

import numpy as np
import pytest

from helpers import oracle_params

DT = 0.001


def x500(O, **kw):
    kw.setdefault("ground_enabled", False)
    return oracle_params("x500", **kw)


def test_default_params_are_x500(oracle):
    p = oracle.default_params()
    assert (p.n_motors, p.mass, p.kf, p.km) == (4, 2.0, 0.00000027087, 0.07)
    J = np.array(p.J).reshape(3, 3)
    # multirotor_model.hpp:44-47
    assert J[0, 0] == 2.0 * (3.0 * 0.25 * 0.25 + 0.1 * 0.1) / 12.0 == J[1, 1]
    assert J[2, 2] == (2.0 * 0.25 * 0.25) / 2.0
    assert abs(J[0, 0] - 0.0329166666666) < 1e-12 and J[2, 2] == 0.0625
    A = np.array(p.allocation_matrix).reshape(4, 8)[:, :4]
    assert np.allclose(A[3], 0.00000027087) and np.allclose(A[0], np.array([-.707, .707, .707, -.707]) * 0.25 * 0.00000027087, rtol=1e-15)
    assert np.allclose(A[2], np.array([-1, -1, 1, 1]) * 0.07 * (3 * 0.15) * 0.00000027087, rtol=1e-15)
    assert p.takeoff_patch_enabled == 1 and p.ground_enabled == 0


def test_airframe_params_match_default_ctor(oracle):
    """config/uavs/x500.yaml through the UavSystemRos init == the header's default ModelParams."""
    p, d = x500(oracle, takeoff_patch_enabled=True), oracle.default_params()
    for k in ("n_motors", "mass", "kf", "km", "prop_radius", "arm_length", "body_height", "motor_time_constant",
              "max_rpm", "min_rpm", "air_resistance_coeff", "g"):
        assert getattr(p, k) == getattr(d, k), k
    assert list(p.J) == list(d.J) and list(p.allocation_matrix) == list(d.allocation_matrix)


def test_hover_is_a_fixed_point(oracle):
    """R=I, v=omega=0, rpm=hover, throttle=hover: one step leaves x,v,omega unchanged; imu=(0,0,g)."""
    O = oracle
    p = x500(O)
    hover_rpm = math.sqrt(2.0 * 9.81 / (4 * 2.7087e-7))
    assert abs(hover_rpm - 4255.386896998892) < 1e-9
    thr = (hover_rpm - 1170) / (7800 - 1170)
    assert abs(thr - 0.4653675561084301) < 1e-14
    s = O.OracleSwarm(1)
    s.construct(0, 1, p, [[1.0, 2.0, 3.0]], [0.0])
    s.set_state(0, 1, x=[[1, 2, 3]], v=np.zeros((1, 3)), R=np.eye(3)[None], omega=np.zeros((1, 3)),
                motor_rpm=[[hover_rpm] * 4 + [0] * 4])
    s.set_input(0, 1, O.ACTUATOR_CMD, [[thr] * 4])
    for _ in range(10):
        s.step(DT)
    st = s.get_state()
    assert np.allclose(st["x"], [[1, 2, 3]], atol=1e-13) and np.allclose(st["v"], 0, atol=1e-12)
    assert np.allclose(st["omega"], 0, atol=1e-13) and np.allclose(st["R"][0], np.eye(3), atol=1e-15)
    assert np.allclose(s.get_imu(), [[0, 0, 9.81]], atol=1e-9)
    assert np.allclose(st["motor_rpm"][0, :4], hover_rpm, rtol=1e-14)


def test_free_fall_is_exact_for_rk4(oracle):
    """rpm=0, no drag: v_z=-g dt, z=z0-g dt^2/2 (RK4 integrates quadratics exactly); rpm after the filter = (1-c)*min_rpm."""
    O = oracle
    p = x500(O)
    p.air_resistance_coeff = 0.0
    s = O.OracleSwarm(1)
    s.construct(0, 1, p, [[0, 0, 10.0]], [0.0])
    s.set_input(0, 1, O.ACTUATOR_CMD, [[0.0] * 4])
    s.step(DT)
    st = s.get_state()
    assert abs(st["v"][0, 2] - (-9.81 * DT)) < 1e-17 and abs(st["x"][0, 2] - (10 - 0.5 * 9.81 * DT * DT)) < 2e-15
    assert abs((st["x"][0, 2] - 10) - (-4.905e-06)) < 1e-15
    c = math.exp(-DT / 0.03)
    assert abs(c - 0.9672161004820059) < 1e-16 and abs(math.exp(-0.01 / 0.03) - 0.7165313105737893) < 1e-16
    assert np.allclose(st["motor_rpm"][0, :4], (1 - c) * 1170, rtol=1e-15) and abs(st["motor_rpm"][0, 0] - 38.357162436) < 1e-8
    # imu of a free-falling body: R^T((v-v_prev)/dt + g e3) = 0
    assert np.allclose(s.get_imu(), 0, atol=1e-12)
    assert np.array_equal(st["v_prev"], st["v"])


def test_torque_free_spin_about_body_z(oracle):
    """omega=(0,0,w), balanced rotors: R stays a rotation about z by w*dt (RK4 truncation (w dt)^5/120), omega constant."""
    O = oracle
    p = x500(O)
    w = 7.0
    hover_rpm = math.sqrt(2.0 * 9.81 / (4 * 2.7087e-7))
    s = O.OracleSwarm(1)
    s.construct(0, 1, p, [[0, 0, 10.0]], [0.0])
    s.set_state(0, 1, x=[[0, 0, 10]], v=np.zeros((1, 3)), R=np.eye(3)[None], omega=[[0, 0, w]], motor_rpm=[[hover_rpm] * 4 + [0] * 4])
    s.set_input(0, 1, O.ACTUATOR_CMD, [[(hover_rpm - 1170) / 6630] * 4])
    n = 100
    for _ in range(n):
        s.step(DT)
    st = s.get_state()
    a = w * DT * n
    Rz = np.array([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]])
    assert np.allclose(st["R"][0], Rz, atol=1e-11)
    assert np.allclose(st["omega"][0], [0, 0, w], atol=1e-12)
    assert np.allclose(st["R"][0].T @ st["R"][0], np.eye(3), atol=1e-15)


def test_mixer_allocation_matrices(oracle):
    """Mixer::calculateAllocation (mixer.hpp:72-101): x500 and f550 normalised pseudo-inverses (SURVEY §8c vi)."""
    O = oracle
    s = O.OracleSwarm(2)
    s.construct(0, 1, x500(O), [[0, 0, 0]], [0.0])
    s.construct(1, 1, oracle_params("f550"), [[0, 0, 0]], [0.0])
    r = math.sqrt(0.5)
    assert np.allclose(s.get_mixer_allocation(0), [[-r, -r, -1, 1], [r, r, -1, 1], [r, -r, 1, 1], [-r, r, 1, 1]], atol=1e-12)
    f = s.get_mixer_allocation(1)
    exp = [[1, 0, 1, 1], [-1, 0, -1, 1], [-.501718, -.865031, 1, 1], [.501718, .865031, -1, 1], [.501718, -.865031, -1, 1],
           [-.501718, .865031, 1, 1]]
    assert np.allclose(f, exp, atol=1e-6)
    assert np.allclose(np.hypot(f[:, 0], f[:, 1]), 1, atol=1e-15)


def test_warmup_steps_of_uav_system_ros(oracle):
    """src/uav_system_ros.cpp:223-232: two makeStep(0.01) with zero actuators on the ground: x unchanged, v=0,
    rpm = min_rpm*(1-c^2), c=exp(-.01/.03) -> 569.301970732."""
    O = oracle
    p = oracle_params("x500", ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    s = O.OracleSwarm(1)
    s.construct(0, 1, p, [[10, 15, 0]], [3.14])
    s.set_input(0, 1, O.ACTUATOR_CMD, [[0.0] * 4])
    s.step(0.01)
    s.step(0.01)
    st = s.get_state()
    assert np.array_equal(st["x"], [[10, 15, 0]]) and np.array_equal(st["v"], np.zeros((1, 3)))
    c = math.exp(-0.01 / 0.03)
    assert np.allclose(st["motor_rpm"][0, :4], 1170 * (1 - c) * (1 + c), rtol=1e-15)
    assert abs(st["motor_rpm"][0, 0] - 569.301970732) < 1e-8
    # spawn attitude is AngleAxis(-heading, z) (multirotor_model.hpp:174,443)
    assert np.allclose(st["R"][0], [[math.cos(3.14), math.sin(3.14), 0], [-math.sin(3.14), math.cos(3.14), 0], [0, 0, 1]], atol=1e-15)


def test_position_cascade_smoke_trajectory(oracle):
    """BASELINE config 1: x500 at (10,15,0) hdg 3.14, warm-up, POSITION_CMD (12,13,5,hdg 1.0), dt=1 ms.
    Values agree with an independent numpy transcription made during the survey."""
    O = oracle
    p = oracle_params("x500", ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    s = O.OracleSwarm(1)
    s.construct(0, 1, p, [[10, 15, 0]], [3.14])
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(s, nm)(0, 1)
    s.set_input(0, 1, O.ACTUATOR_CMD, [[0.0] * 4])
    s.step(0.01)
    s.step(0.01)
    s.set_input(0, 1, O.POSITION_CMD, [[12, 13, 5, 1.0]])
    s.step(DT)
    st = s.get_state()
    assert np.allclose(st["motor_rpm"][0, :4], [796.932519, 679.666406, 799.055570, 588.995195], atol=1e-5)
    assert np.array_equal(st["x"], [[10, 15, 0]])
    s.step_n(DT, 999)
    assert np.allclose(s.get_state()["x"][0], [11.291111, 14.513544, 1.660905], atol=2e-6)
    s.step_n(DT, 19000)
    st = s.get_state()
    assert np.allclose(st["x"][0], [11.997797, 13.003830, 4.995801], atol=2e-6)
    assert np.allclose(st["motor_rpm"][0, :4], 4255.3767, atol=1e-3)
    assert abs(math.atan2(st["R"][0, 1, 0], st["R"][0, 0, 0]) - 1.000007) < 1e-6


@pytest.mark.parametrize("case", [
    # kp kd ki sat aw  le integ err dt  -> out, le', integ'
    (2.0, 0.15, 0.2, 6.0, 1.0, 0.0, 0.0, 0.25, 0.001, (2.0 * 0.25 + 0.15 * 250.0, 0.25, 0.0)),   # derivative kick 38 -> saturates at 6
    (2.0, 0.0, 0.2, 6.0, 1.0, 0.25, 0.5, 0.25, 0.001, (0.6, 0.25, 0.5 + 0.25e-3)),              # integral used BEFORE update
    (2.0, 0.0, 0.0, 6.0, 1.0, 0.5, 0.0, 0.5, 0.001, (1.0, 0.5, 0.0)),                           # |sum| == aw: strict '<' -> no windup
    (2.0, 0.0, 0.0, 6.0, 1.0, -4.0, 0.0, -4.0, 0.001, (-6.0, -4.0, 0.0)),                        # negative saturation, no windup
    (4.0, 0.04, 0.0, -1.0, 1.0, 0.0, 0.0, 100.0, 0.01, (400.0 + 0.04 * 1e4, 100.0, 0.0)),        # sat<=0: unbounded
])
def test_pid_update(oracle, case):
    import ctypes as C
    kp, kd, ki, sat, aw, le, integ, err, dt, (out, le2, in2) = case
    le_c, in_c = C.c_double(le), C.c_double(integ)
    got = oracle.lib().orc_pid_update(kp, kd, ki, sat, aw, C.byref(le_c), C.byref(in_c), err, dt)
    exp_out = min(max(out, -sat), sat) if sat > 0 else out
    assert got == pytest.approx(exp_out, rel=1e-15) and le_c.value == le2 and in_c.value == pytest.approx(in2, rel=1e-15)


def test_reorth_and_inverse_leaves(oracle):
    import ctypes as C
    rng = np.random.default_rng(0)
    dp = C.POINTER(C.c_double)
    for _ in range(20):
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0] + 1e-3 * rng.normal(size=(3, 3))
        out = np.zeros(9)
        oracle.lib().orc_llt_reorth(np.ascontiguousarray(R).ctypes.data_as(dp), out.ctypes.data_as(dp))
        Rh = out.reshape(3, 3)
        L = np.linalg.cholesky(R.T @ R)
        assert np.allclose(Rh, R @ np.linalg.inv(L), atol=1e-13)  # L^-1, not L^-T: first-order orthonormalisation
        Mx = rng.normal(size=(3, 3))
        oracle.lib().orc_inverse3(np.ascontiguousarray(Mx).ctypes.data_as(dp), out.ctypes.data_as(dp))
        assert np.allclose(out.reshape(3, 3), np.linalg.inv(Mx), rtol=1e-9, atol=1e-11)
        A4 = rng.normal(size=(4, 4))
        o16 = np.zeros(16)
        oracle.lib().orc_inverse_lu(np.ascontiguousarray(A4).ctypes.data_as(dp), 4, o16.ctypes.data_as(dp))
        assert np.allclose(o16.reshape(4, 4), np.linalg.inv(A4), rtol=1e-8, atol=1e-10)


def test_nan_throttle_becomes_zero_motors(oracle):
    """negative thrust_force -> sqrt -> NaN throttle -> NaN motors -> isfinite -> 0 (acceleration_controller.hpp:91-94,
    multirotor_model.hpp:398-400)."""
    O = oracle
    s = O.OracleSwarm(1)
    s.construct(0, 1, x500(O), [[0, 0, 50]], [0.0])
    Rflip = np.diag([1.0, -1.0, -1.0])  # upside down: fd . R e3 < 0
    s.set_state(0, 1, x=[[0, 0, 50]], v=np.zeros((1, 3)), R=Rflip[None], omega=np.zeros((1, 3)), motor_rpm=[[3000.0] * 4 + [0] * 4])
    s.set_input(0, 1, O.ACCELERATION_HDG_CMD, [[0, 0, 0, 0.3]])
    s.step(DT)
    c = math.exp(-DT / 0.03)
    assert np.allclose(s.get_state()["motor_rpm"][0, :4], c * 3000 + (1 - c) * 1170, rtol=1e-15)
    assert np.all(np.isfinite(s.get_state()["x"]))


def test_ground_and_takeoff_patch(oracle):
    O = oracle
    p = oracle_params("x500", ground_enabled=True, ground_z=1.0, takeoff_patch_enabled=True)
    s = O.OracleSwarm(1)
    s.construct(0, 1, p, [[0, 0, 3.0]], [0.0])
    s.set_input(0, 1, O.ACTUATOR_CMD, [[0.0] * 4])
    s.step_n(DT, 50)
    st = s.get_state()
    assert st["x"][0, 2] == 3.0 and np.all(st["v"] == 0)  # held by the take-off patch at spawn height
    assert s.get_params(0).takeoff_patch_enabled == 1
    s.set_input(0, 1, O.ACTUATOR_CMD, [[1.0] * 4])  # mean input > 0.9 hover -> patch disabled forever
    s.step(DT)
    assert s.get_params(0).takeoff_patch_enabled == 0
    s.set_input(0, 1, O.ACTUATOR_CMD, [[0.0] * 4])
    s.step_n(DT, 3000)
    st = s.get_state()
    assert st["x"][0, 2] == 1.0 and np.all(st["v"] == 0) and np.all(st["omega"] == 0)  # now resting on ground_z


def test_crash_zeroes_motors_forever(oracle):
    O = oracle
    s = O.OracleSwarm(2)
    s.construct(0, 2, x500(O), [[0, 0, 50], [5, 0, 50]], [0.0, 0.0])
    s.set_input(0, 2, O.ACTUATOR_CMD, [[0.6] * 4] * 2)
    s.crash(1, 1)
    s.step_n(DT, 200)
    rpm = s.get_state()["motor_rpm"]
    assert rpm[0, 0] > 3000 and rpm[1, 0] < 1171 and list(s.has_crashed()) == [0, 1]


def test_mixer_desaturation_abs_overload(oracle, tmp_path):
    """mixer.hpp:121 `actuators.motors.array() += abs(min)` — UNQUALIFIED abs on a double.  Every TU that holds the line includes
    Eigen, whose Eigen/Core pulls <emmintrin.h> -> <mm_malloc.h> -> <stdlib.h> on x86-64, which brings std::abs(double) into the
    global namespace: the offset is |min| (DESIGN.md §9 records the probe).  The oracle must NOT truncate (the int overload would
    add 0 for any min in (-1, 0)); the truncating reading exists as an oracle-only build (`make -C oracle absint`) and is shown
    here to differ, so a reference-held fixture could tell the two apart."""
    import ctypes as C
    import subprocess
    O = oracle
    s = O.OracleSwarm(1)
    s.construct(0, 1, x500(O), [[0, 0, 0]], [0.0])
    # x500 rows (roll, pitch, yaw, throttle): (-r,-r,-1,1) (r,r,-1,1) (r,-r,1,1) (-r,r,1,1), r = sqrt(1/2)
    cg = np.array([[0.0, 0.0, 0.4, 0.1]])  # motors before desaturation: (-0.3, -0.3, 0.5, 0.5): min = -0.3, max after the offset 0.8
    m = s.debug_component(3, 0, 1, cg)[0, :4]  # MRS_COMP_MIXER
    assert np.allclose(m, [0.0, 0.0, 0.8, 0.8], atol=1e-15), m
    # the truncating variant, built on the spot from the same source with -DORC_MIXER_ABS_INT
    import os
    src = os.path.join(os.path.dirname(O.__file__), "uav_oracle.c")
    so = str(tmp_path / "liboracle_absint.so")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-fPIC", "-ffp-contract=off", "-DORC_MIXER_ABS_INT", "-shared", "-o", so, src, "-lm", "-lpthread"])
    L = C.CDLL(so)
    dp = C.POINTER(C.c_double)
    L.orc_swarm_create.restype = C.c_void_p
    L.orc_swarm_create.argtypes = [C.c_int32]
    L.orc_swarm_destroy.argtypes = [C.c_void_p]
    L.orc_swarm_construct.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(O.ModelParams), dp, dp]
    L.orc_swarm_debug_component.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, dp, C.c_int32, dp, C.c_int32, C.c_double]
    h = L.orc_swarm_create(1)
    p = x500(O)
    pos, hd = np.zeros(3), np.zeros(1)
    L.orc_swarm_construct(h, 0, 1, C.byref(p), pos.ctypes.data_as(dp), hd.ctypes.data_as(dp))
    out = np.zeros(8)
    L.orc_swarm_debug_component(h, 3, 0, 1, cg.ctypes.data_as(dp), 4, out.ctypes.data_as(dp), 8, 0.001)
    L.orc_swarm_destroy(h)
    assert np.allclose(out[:4], [-0.3, -0.3, 0.5, 0.5], atol=1e-15), out  # int overload: abs((int)-0.3) == 0, nothing is added
