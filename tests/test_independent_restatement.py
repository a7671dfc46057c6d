"""oracle/uav_oracle.c against a second, independently written restatement of the same reference functions
(tests/independent_model.py: per-UAV objects, whole-matrix numpy/LAPACK expressions).  The dynamics part of the oracle is
"parity unpinned" (DESIGN.md §2); two restatements written in different styles agreeing to 1e-9 over closed-loop trajectories of
every input mode, with feed-forwards, crashes, pushes, the ground and the take-off patch, is the strongest check available here.
Tolerance: 1e-9 relative (L-inf per field) — LAPACK's Cholesky / inverses and numpy's summation orders differ from the Eigen orders
the oracle spells out in the last bits, and the loop is closed around them for hundreds of steps."""
import numpy as np
import pytest

import helpers
import independent_model as IM

RTOL = 1e-9


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle_swarm as O
    O.build()
    return O


def payload(O, mode, rng, nm, x):
    if mode == O.ACTUATOR_CMD:
        return rng.uniform(0.35, 0.60, nm)
    if mode == O.CONTROL_GROUP_CMD:
        return np.concatenate([rng.uniform(-0.1, 0.1, 3), rng.uniform(0.3, 0.7, 1)])
    if mode == O.ATTITUDE_RATE_CMD:
        return np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(0.3, 0.7, 1)])
    if mode == O.ATTITUDE_CMD:
        return np.concatenate([helpers.tilted_rotations(rng, 1, 0.4).reshape(9), rng.uniform(0.3, 0.7, 1)])
    if mode == O.TILT_HDG_RATE_CMD:
        tilt = helpers.tilted_rotations(rng, 1, 0.4)[0, :, 2] * rng.uniform(0.5, 3.0)
        return np.concatenate([tilt, rng.uniform(-1, 1, 1), rng.uniform(0.3, 0.7, 1)])
    if mode in (O.ACCELERATION_HDG_RATE_CMD, O.ACCELERATION_HDG_CMD):
        return np.concatenate([rng.uniform(-2, 2, 3), rng.uniform(-1, 1, 1)])
    if mode in (O.VELOCITY_HDG_RATE_CMD, O.VELOCITY_HDG_CMD):
        return np.concatenate([rng.uniform(-3, 3, 3), rng.uniform(-1, 1, 1)])
    if mode == O.POSITION_CMD:
        return np.concatenate([x + rng.uniform(-5, 5, 3), rng.uniform(-3.14, 3.14, 1)])
    return None


def compare(o, u, what):
    so = o.get_state()
    nm = u.p["n_motors"]
    for name, a, b in (("x", so["x"][0], u.x), ("v", so["v"][0], u.v), ("R", so["R"][0], u.R), ("omega", so["omega"][0], u.omega),
                       ("motor_rpm", so["motor_rpm"][0][:nm], u.motor_rpm), ("imu", o.get_imu()[0], u.imu)):
        a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
        assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: {name}: NaN pattern"
        err = np.nanmax(np.abs(a - b), initial=0.0) / max(1.0, np.nanmax(np.abs(a), initial=0.0))  # relative, absolute below magnitude 1
        assert err <= RTOL, f"{what}: {name}: error {err:.3e} > {RTOL:.1e}"


def make_pair(O, airframe, rng, spawn=True, **kw):
    po = helpers.oracle_params(airframe, **kw)
    pos = rng.uniform(-20, 20, 3) + [0, 0, 30]
    hdg = rng.uniform(-3, 3)
    o = O.OracleSwarm(1)
    o.construct(0, 1, po, pos[None, :], np.array([hdg]))
    u = IM.Uav(IM.params_from_struct(po), pos, hdg)
    return o, u, po


@pytest.mark.parametrize("airframe", ["x500", "f550", "naki"])
@pytest.mark.parametrize("mode", list(range(0, 11)))
def test_every_input_mode_closed_loop(oracle, airframe, mode):
    """Spawn (AngleAxis(-heading)), the two warm-up steps of UavSystemRos (0.01 s, zero actuators), then 400 steps of 1 ms in the
    given mode from a perturbed flying state; a feed-forward of each kind arrives on the way."""
    O = oracle
    rng = np.random.default_rng(1000 + mode)
    o, u, po = make_pair(O, airframe, rng, ground_enabled=True, ground_z=0.0)
    nm = int(po.n_motors)
    for _ in range(2):
        o.set_input(0, 1, O.ACTUATOR_CMD, np.zeros((1, nm)))
        u.set_input(IM.ACTUATOR_CMD, np.zeros(nm))
        o.step(0.01)
        u.make_step(0.01)
    compare(o, u, "warm-up")
    st = helpers.random_state(rng, 1, nm, tilted=True)
    o.set_state(0, 1, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    u.x, u.v, u.R, u.omega, u.motor_rpm = st["x"][0].copy(), st["v"][0].copy(), st["R"][0].copy(), st["omega"][0].copy(), st["motor_rpm"][0][:nm].copy()
    pl = payload(O, mode, rng, nm, st["x"][0])
    o.set_input(0, 1, mode, None if pl is None else pl[None, :])
    u.set_input(mode, pl)
    for k in range(400):
        if k in (100, 150, 200, 250):
            kind = (k - 100) // 50
            ff = np.concatenate([rng.uniform(-1, 1, 3), rng.uniform(-0.5, 0.5, 1)])
            o.set_feedforward(0, 1, kind, ff[None, :])
            u.set_feedforward(kind, ff)
        o.step(0.001)
        u.make_step(0.001)
        if k % 50 == 49:
            compare(o, u, f"mode {mode} step {k + 1}")


def test_push_crash_ground_and_takeoff_patch(oracle):
    """applyForce latch, crash (zero motors until the ground), the ground plane, and the take-off patch that holds the spawn height
    until the mean motor input exceeds 0.9 hover."""
    O = oracle
    rng = np.random.default_rng(77)
    o, u, po = make_pair(O, "x500", rng, ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=True)
    z0 = u.x[2]
    o.set_input(0, 1, O.ACTUATOR_CMD, np.full((1, 4), 0.2))
    u.set_input(IM.ACTUATOR_CMD, np.full(4, 0.2))
    for k in range(50):
        o.step(0.001)
        u.make_step(0.001)
    compare(o, u, "held by the take-off patch")
    assert u.x[2] == z0 and u.takeoff_patch_enabled
    goal = np.concatenate([u.x + [1.0, -2.0, 3.0], [0.5]])
    o.set_input(0, 1, O.POSITION_CMD, goal[None, :])
    u.set_input(IM.POSITION_CMD, goal)
    for k in range(600):
        if k == 300:
            o.apply_force(0, 1, np.array([[3.0, -2.0, 1.0]]))
            u.apply_force([3.0, -2.0, 1.0])
        o.step(0.001)
        u.make_step(0.001)
        if k % 100 == 99:
            compare(o, u, f"flying, step {k + 1}")
    assert not u.takeoff_patch_enabled and u.x[2] > z0
    o.crash(0, 1)
    u.crash()
    o.set_state(0, 1, x=np.array([[0.0, 0.0, 0.3]]))
    u.x = np.array([0.0, 0.0, 0.3])
    for k in range(800):
        o.step(0.001)
        u.make_step(0.001)
        if k % 100 == 99:
            compare(o, u, f"crashed, step {k + 1}")
    assert u.x[2] == 0.0 and not u.v.any()


def test_mixer_allocation_and_saturation_branches(oracle):
    """CTL/mixer.hpp: the normalised pseudo-inverse and the three desaturation branches, per airframe."""
    O = oracle
    for airframe in ("x500", "f550", "naki"):
        po = helpers.oracle_params(airframe)
        nm = int(po.n_motors)
        o = O.OracleSwarm(1)
        o.construct(0, 1, po, np.zeros((1, 3)), np.zeros(1))
        u = IM.Uav(IM.params_from_struct(po), np.zeros(3), 0.0)
        helpers.assert_close(np.asarray(o.get_mixer_allocation(0))[:nm], u.allocation_inv, 1e-12, airframe + " allocation")
        for cg in ([0.0, 0.0, 0.0, 0.5], [0.9, -0.4, 0.3, 0.6], [0.3, 0.2, -0.9, 0.005], [2.0, 1.5, 0.4, 0.9], [-0.2, 0.1, 0.05, 0.0]):
            o.set_state(0, 1, motor_rpm=np.zeros((1, O.MAX_MOTORS)))
            u.motor_rpm = np.zeros(nm)
            o.set_input(0, 1, O.CONTROL_GROUP_CMD, np.array([cg]))
            u.set_input(IM.CONTROL_GROUP_CMD, cg)
            o.step(0.001)
            u.make_step(0.001)
            # the motor filter from zero rpm exposes the clamped mixer output: rpm = (1 - c) * (min + (max - min) * motors)
            helpers.assert_close(o.get_state()["motor_rpm"][0][:nm], u.motor_rpm, 1e-12, f"{airframe} mixer {cg}")


@pytest.mark.parametrize("mode", [6, 7, 9, 10])
def test_inverted_uav_takes_the_nan_throttle_path(oracle, mode):
    """A UAV flipped beyond 90 degrees in a cascade mode: the desired force has a negative component along body z, the throttle
    sqrt is NaN (CTL/acceleration_controller.hpp:91-94), the mixer output is NaN and MultirotorModel::setInput turns it into zero
    throttle (MM:397-399) while the attitude/rate PIDs keep integrating.  Both restatements must walk the same path."""
    O = oracle
    rng = np.random.default_rng(300 + mode)
    o, u, po = make_pair(O, "x500", rng, ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    R = helpers.tilted_rotations(rng, 1, 0.3)[0] @ np.diag([1.0, -1.0, -1.0])
    rpm = rng.uniform(3000, 5000, 4)
    full = np.zeros((1, O.MAX_MOTORS))
    full[0, :4] = rpm
    o.set_state(0, 1, R=R[None], omega=np.array([[0.3, -0.2, 0.1]]), motor_rpm=full)
    u.R, u.omega, u.motor_rpm = R.copy(), np.array([0.3, -0.2, 0.1]), rpm.copy()
    pl = payload(O, mode, rng, 4, u.x)
    o.set_input(0, 1, mode, pl[None, :])
    u.set_input(mode, pl)
    saw_nan_throttle = False
    for k in range(400):
        o.step(0.001)
        u.make_step(0.001)
        saw_nan_throttle |= bool(np.isnan(u.actuators).any())
        if k % 50 == 49:
            compare(o, u, f"inverted, mode {mode}, step {k + 1}")
    assert saw_nan_throttle


def test_config1_smoke_values_of_the_survey(oracle):
    """BASELINE config 1 (x500 spawned at (10,15,0) heading 3.14, two warm-up steps, POSITION_CMD (12,13,5) heading 1.0, dt 1 ms)
    against the values a third, throw-away transcription gave during the survey (6-7 significant digits)."""
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    u = IM.Uav(IM.params_from_struct(po), [10.0, 15.0, 0.0], 3.14)
    for _ in range(2):
        u.set_input(IM.ACTUATOR_CMD, np.zeros(4))
        u.make_step(0.01)
    assert np.allclose(u.motor_rpm, 569.301970732, rtol=1e-11) and np.array_equal(u.x, [10.0, 15.0, 0.0])
    u.set_input(IM.POSITION_CMD, [12.0, 13.0, 5.0, 1.0])
    u.make_step(0.001)
    assert np.allclose(u.motor_rpm, [796.932519, 679.666406, 799.055570, 588.995195], rtol=2e-9)
    for _ in range(999):
        u.make_step(0.001)
    assert np.allclose(u.x, [11.291111, 14.513544, 1.660905], atol=2e-6)
