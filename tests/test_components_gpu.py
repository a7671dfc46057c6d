"""The components of the hot path ONE AT A TIME (rows C2-C8 / M3-M6 could only be tested through the cascade).

mrs_swarm_debug_component runs a single device function of the step kernels — re-orthonormalisation, the ODE right-hand side, the
mixer, each controller — for every UAV of a swarm on that UAV's own state, constants and PID columns; orc_swarm_debug_component
runs the oracle's function of the same name on the same inputs.  Inputs are adversarial on purpose: near-singular rotation
matrices, all three desaturation branches of the mixer (mixer.hpp:116-141), NaN throttles, the fabs(heading_rate) < 1e-3 switch
(attitude_controller.hpp:216), degenerate attitudes that trip the three warnings.  LITERAL: reference operation order (agreement
to rounding); FAST: FMA / reciprocal forms (1e-9).  A third opinion comes from tests/independent_model.py where it has the function."""
import numpy as np
import pytest

import helpers
import independent_model as IM

pytestmark = pytest.mark.gpu
DT = 0.001
AIRFRAMES = ("x500", "f550", "naki")


def make_pair(M, n, airframe, rng, arith, tilted=True):
    p = helpers.Pair(M, n, arith=arith)
    po = p.construct(0, n, airframe, ground_enabled=True)
    st = helpers.random_state(rng, n, po.n_motors, tilted=tilted)
    p.set_state(0, n, st)
    return p, po, st


def rows_close(a, b, rtol, what):
    """row-wise relative L-inf: every UAV's output vector on its own scale (floor 1); NaN / inf patterns identical"""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN pattern differs"
    fin = np.isfinite(b)
    assert np.array_equal(a[~fin & ~np.isnan(b)], b[~fin & ~np.isnan(b)]), f"{what}: inf pattern differs"
    d = np.where(fin, np.abs(np.where(fin, a, 0) - np.where(fin, b, 0)), 0).max(axis=1)
    sc = np.maximum(np.where(fin, np.abs(b), 0).max(axis=1), 1.0)
    k = int(np.argmax(d / sc))
    assert (d / sc)[k] <= rtol, f"{what}: row {k} off by {(d / sc)[k]:.3e} > {rtol:.1e}\n gpu {a[k]}\n ref {b[k]}"
    return float((d / sc).max())


def both(p, comp, rows, dt=DT):
    n = len(rows)
    return p.g.debug_component(comp, 0, n, rows, dt), p.o.debug_component(comp, 0, n, rows, dt)


@pytest.mark.parametrize("fast", [False, True])
def test_reorthonormalisation(mrs, oracle, fast):
    M = mrs
    rng = np.random.default_rng(1)
    n = 4096
    p, _, _ = make_pair(M, n, "x500", rng, M.ARITH_FAST if fast else M.ARITH_LITERAL)
    R = helpers.random_rotations(rng, n)
    R[:1000] += rng.normal(0, 1e-3, (1000, 3, 3))            # what an RK stage hands in
    R[1000:2000] *= rng.uniform(0.2, 5.0, (1000, 1, 1))      # scaled
    R[2000:3000] += rng.normal(0, 0.3, (1000, 3, 3))         # badly skewed
    if not fast:  # nearly and exactly singular: Eigen's llt stops at the non-positive pivot; LITERAL follows it entry by entry
        R[3000:3500, :, 2] = R[3000:3500, :, 0] * rng.uniform(0.5, 2, (500, 1)) + rng.normal(0, 1e-7, (500, 3))
        R[3500:3600, :, 1] = 0.0
        R[3600:3650] = 0.0
    g, o = both(p, M.swarm.COMP_REORTH, R.reshape(n, 9))
    rows_close(g[:3000], o[:3000], 1e-9 if fast else 1e-13, "reorth")
    if not fast:
        # ill-conditioned inputs amplify the last-bit differences of sqrt / division: hold them to the conditioning, not to 1e-13
        assert np.array_equal(np.isnan(g), np.isnan(o)) and np.array_equal(np.isinf(g), np.isinf(o))
        fin = np.isfinite(o)
        scale = np.maximum(np.abs(np.where(fin, o, 0)).max(axis=1, keepdims=True), 1.0)
        assert (np.abs(np.where(fin, g - o, 0)) / scale).max() < 1e-6
    Rh = o[:1000].reshape(-1, 3, 3)
    assert np.abs(np.einsum("nij,nik->njk", Rh, Rh) - np.eye(3)).max() < 3e-4  # R L^-1 only orthonormalises to first order (multirotor_model.hpp:249-253)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("airframe", AIRFRAMES)
def test_model_rhs(mrs, oracle, fast, airframe):
    M = mrs
    rng = np.random.default_rng(2)
    n = 2048
    p, po, st = make_pair(M, n, airframe, rng, M.ARITH_FAST if fast else M.ARITH_LITERAL, tilted=False)
    p.both("apply_force", 0, n, rng.normal(0, 5, (n, 3)))
    y = np.concatenate([st["x"], rng.normal(0, 8, (n, 3)), (st["R"] + rng.normal(0, 2e-3, (n, 3, 3))).reshape(n, 9), rng.normal(0, 3, (n, 3))], axis=1)
    y[:50, 3:6] = 0.0        # |v| == 0: the normalisation branch of multirotor_model.hpp:339-342
    y[50:60, 15] = np.nan    # a NaN body rate poisons omega_dot and R_dot: components -> 0 (:361-365)
    y[60:70, 4] = np.inf
    g, o = both(p, M.swarm.COMP_MODEL_RHS, y)
    rows_close(g, o, 1e-9 if fast else 1e-12, f"model_rhs {airframe}")
    # third opinion
    u = IM.Uav(IM.params_from_struct(po))
    for k in (0, 100, 777):
        u.motor_rpm = st["motor_rpm"][k, :po.n_motors].copy()
        u.external_force = p.o.get_external_force(k, 1)[0]
        yy = np.concatenate([y[k, 0:6], y[k, 6:15].reshape(3, 3).T.ravel(), y[k, 15:18]])
        d = u.rhs(yy)
        back = np.concatenate([d[0:6], d[6:15].reshape(3, 3).T.ravel(), d[15:18]])
        assert np.allclose(back, o[k], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("fast", [False, True])
@pytest.mark.parametrize("airframe", AIRFRAMES)
def test_mixer_every_desaturation_branch(mrs, oracle, fast, airframe):
    M = mrs
    rng = np.random.default_rng(3)
    n = 3000
    p, po, _ = make_pair(M, n, airframe, rng, M.ARITH_FAST if fast else M.ARITH_LITERAL)
    cg = np.concatenate([rng.uniform(-0.2, 0.2, (n, 3)), rng.uniform(0.2, 0.8, (n, 1))], axis=1)      # no saturation
    cg[500:1000, :3] = rng.uniform(-1.5, 1.5, (500, 3))                                                # some motor below zero
    cg[1000:1500] = np.concatenate([rng.uniform(-1, 1, (500, 3)), rng.uniform(0.7, 1.3, (500, 1))], axis=1)  # above one, throttle > 1e-2
    cg[1500:2000] = np.concatenate([rng.uniform(-3, 3, (500, 3)), rng.uniform(-0.5, 0.01, (500, 1))], axis=1)  # above one, throttle <= 1e-2
    cg[2000:2010, 0] = np.nan
    g, o = both(p, M.swarm.COMP_MIXER, cg)
    rows_close(g, o, 1e-9 if fast else 1e-13, f"mixer {airframe}")
    m = o[:, :po.n_motors]
    assert (m[500:1000].min(axis=1) >= -1e-12).all() and (m[1000:1500].max(axis=1) > 1.0).any()  # the branches were really taken
    u = IM.Uav(IM.params_from_struct(po))
    for k in (0, 600, 1200, 1700):
        assert np.allclose(u.mixer(cg[k]), m[k], rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("fast", [False, True])
def test_position_velocity_and_rate_controllers_keep_their_pid_state(mrs, oracle, fast):
    M = mrs
    rng = np.random.default_rng(4)
    n = 1024
    p, _, st = make_pair(M, n, "f550", rng, M.ARITH_FAST if fast else M.ARITH_LITERAL)
    tol = 1e-9 if fast else 1e-13
    for call in range(6):  # the PIDs integrate and differentiate over the calls; large errors hit the saturation and the anti-windup
        big = 50.0 if call % 2 else 1.0
        g, o = both(p, M.swarm.COMP_POSITION, st["x"] + rng.normal(0, big, (n, 3)))
        rows_close(g, o, tol, f"position controller call {call}")
        g, o = both(p, M.swarm.COMP_VELOCITY, rng.normal(0, big, (n, 3)))
        rows_close(g, o, tol, f"velocity controller call {call}")
        g, o = both(p, M.swarm.COMP_RATE, np.concatenate([rng.normal(0, big, (n, 3)), rng.uniform(0, 1, (n, 1))], axis=1))
        rows_close(g, o, tol, f"rate controller call {call}")
        helpers.assert_close(p.g.get_pid(), p.o.get_pid(), tol, f"PID state after call {call}")


@pytest.mark.parametrize("fast", [False, True])
def test_acceleration_controllers(mrs, oracle, fast):
    M = mrs
    rng = np.random.default_rng(5)
    n = 4096
    p, po, st = make_pair(M, n, "x500", rng, M.ARITH_FAST if fast else M.ARITH_LITERAL, tilted=False)  # any attitude, upside down included
    acc = np.concatenate([rng.normal(0, 3, (n, 3)), rng.uniform(-np.pi, np.pi, (n, 1))], axis=1)
    acc[:300, 2] = rng.uniform(-30, -9.9, 300)       # desired force points down: sqrt of a negative thrust -> NaN throttle (:91-94)
    acc[300:600, :3] = rng.normal(0, 40, (300, 3))   # violent manoeuvres: z_d far from vertical
    g, o = both(p, M.swarm.COMP_ACCELERATION_HDG, acc)
    rows_close(g, o, 1e-8 if fast else 1e-11, "acceleration controller (heading)")
    assert np.isnan(o[:, 9]).sum() > 100, "the NaN-throttle path must be exercised"
    g, o = both(p, M.swarm.COMP_ACCELERATION_HDG_RATE, acc)
    rows_close(g, o, 1e-9 if fast else 1e-13, "acceleration controller (heading rate)")
    if not fast:  # z_d exactly horizontal along x: the 2x2 block of I - z z^T is singular, Eigen's LU divides by zero — same NaN/inf pattern
        sing = np.tile([7.0, 0.0, -po.g, 0.3], (64, 1))
        g, o = p.g.debug_component(M.swarm.COMP_ACCELERATION_HDG, 0, 64, sing), p.o.debug_component(M.swarm.COMP_ACCELERATION_HDG, 0, 64, sing)
        assert np.array_equal(np.isnan(g), np.isnan(o))
    u = IM.Uav(IM.params_from_struct(po))
    for k in (5, 400, 2000):
        u.R = st["R"][k].copy()
        Rd, thr = u.acceleration_to_attitude(acc[k, :3], acc[k, 3])
        ref = p.o.debug_component(M.swarm.COMP_ACCELERATION_HDG, k, 1, acc[k:k + 1])[0]
        assert np.allclose(np.append(Rd.ravel(), thr), ref, rtol=1e-9, atol=1e-9, equal_nan=True)


@pytest.mark.parametrize("fast", [False, True])
def test_attitude_controllers_and_the_heading_rate_switch(mrs, oracle, fast):
    M = mrs
    rng = np.random.default_rng(6)
    n = 4096
    p, _, st = make_pair(M, n, "x500", rng, M.ARITH_FAST if fast else M.ARITH_LITERAL, tilted=False)
    tol = 1e-8 if fast else 1e-11
    # upright, on its side (body x vertical: denominator of the heading-rate map <= 1e-5, attitude_controller.hpp:195) and random UAVs
    R = st["R"].copy()
    R[:64] = np.array([[0.0, 0, 1], [0, 1, 0], [-1.0, 0, 0]]) + rng.normal(0, 1e-4, (64, 3, 3))
    R[64:128] = np.eye(3)
    st["R"] = R
    p.set_state(0, n, st)
    # ATTITUDE: Rd + throttle
    Rd = helpers.random_rotations(rng, n).reshape(n, 9)
    g, o = both(p, M.swarm.COMP_ATTITUDE, np.concatenate([Rd, rng.uniform(0, 1, (n, 1))], axis=1))
    rows_close(g, o, tol, "attitude controller")
    # TILT + heading rate: tilt = the UAV's own body z (no attitude error apart from the PIDs' memory) and heading rates on both
    # sides of the 1e-3 threshold (attitude_controller.hpp:216), plus random tilts and rates
    tilt = np.concatenate([R[:, :, 2] * rng.uniform(0.5, 2, (n, 1)), rng.uniform(-2, 2, (n, 1)), rng.uniform(0, 1, (n, 1))], axis=1)
    tilt[128:1128, 3] = rng.choice([9e-4, 1.1e-3, -9e-4, -1.1e-3, 0.0], 1000)
    tilt[2000:3000, :3] = rng.normal(0, 1, (1000, 3))
    for call in range(3):
        g, o = both(p, M.swarm.COMP_TILT_HDG_RATE, tilt)
        rows_close(g, o, tol, f"tilt / heading-rate controller call {call}")
        helpers.assert_close(p.g.get_pid(), p.o.get_pid(), tol, "attitude PID state")
    dg, do = p.g.get_diag(), p.o.get_diag()
    assert dg == do and do["hdg_rate_denom_small"] >= 64, (dg, do)  # the warnings of :196,236,245, counted alike
