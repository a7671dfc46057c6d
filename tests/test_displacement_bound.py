"""The displacement bound behind the announced stall indices of the split sharded tick (step_device.inc epilogue, TypeParams
pred_a0 / pred_drag in host_api.hip derive_type), checked against the ORACLE's dynamics on the CPU: over h steps of RK4 a UAV moves at most
    h dt |v| + (h dt)^2 / 2 * A,   A = A0 + (|resist_k| / m) (|v| + h dt A0)^2,   A0 = g + 1.5 (sum_m alloc[3][m] max_rpm^2 + thrust now) / m + a_ext
(a_ext: the collision forces' share, listed partners x |rebounce| in the kernel — here an applied force of that size) as long as the
motor speeds stay within max_rpm (the low-pass keeps them there) and R is near a rotation.  Thousands of random states of three
airframes, actuator and position commands, forces in random directions, dt of 1 and 10 ms, horizons 1..4 — the kernel uses h = 4."""
import numpy as np
import pytest

import helpers
from oracle import oracle_swarm as O


@pytest.mark.parametrize("airframe", ["x500", "f550", "naki"])
@pytest.mark.parametrize("dt", [0.001, 0.01])
@pytest.mark.parametrize("overspeed", [False, True])
def test_rk4_displacement_stays_within_the_announced_bound(airframe, dt, overspeed):
    """overspeed: motor speeds the host SET beyond max_rpm (set_state does not clamp; or max_rpm lowered under running motors) — the
    kernel's bound carries the thrust of the current step next to the max_rpm cap (pred_thr), so it holds there too."""
    rng = np.random.default_rng({"x500": 1, "f550": 2, "naki": 3}[airframe] * 1000 + int(dt * 1e4) + (7 if overspeed else 0))
    n, hmax = 3000, 4
    po = helpers.oracle_params(airframe, ground_enabled=True, ground_z=-1e6)  # (the ground only ever shortens a displacement)
    nm = po.n_motors
    st = helpers.random_state(rng, n, nm, tilted=True)
    st["x"] = rng.uniform(-50, 50, (n, 3)) + [0, 0, 200.0]
    st["v"] = rng.normal(0, 1, (n, 3)) * rng.uniform(0, 25, (n, 1))           # up to tens of m/s
    st["omega"] = rng.normal(0, 1, (n, 3)) * rng.uniform(0, 6, (n, 1))
    st["motor_rpm"][:, :nm] = rng.uniform(po.min_rpm, po.max_rpm, (n, nm))    # anywhere in the admissible range, full throttle included
    if overspeed:
        st["motor_rpm"][: n // 2, :nm] *= rng.uniform(1.0, 1.8, (n // 2, nm))
    o = O.OracleSwarm(n)
    o.construct(0, n, po)
    for nmf in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(o, nmf)(0, n)
    o.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    half = n // 2
    o.set_input(0, half, O.ACTUATOR_CMD, rng.uniform(0.0, 1.0, (half, nm)))    # any throttles
    goals = np.concatenate([st["x"][half:] + rng.uniform(-300, 300, (n - half, 3)), rng.uniform(-3, 3, (n - half, 1))], axis=1)
    o.set_input(half, n - half, O.POSITION_CMD, goals)                          # far goals: saturated controllers
    a_ext = rng.choice([0.0, 100.0, 300.0, 2400.0], n)                          # 0, 1, 3, 24 listed partners at rebounce 100
    dirs = rng.normal(0, 1, (n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    o.apply_force(0, n, dirs * (a_ext * po.mass)[:, None])
    alloc3 = np.array([po.allocation_matrix[3 * O.MAX_MOTORS + m] for m in range(nm)])
    assert (alloc3 >= 0).all()
    thrust_now = (st["motor_rpm"][:, :nm] ** 2) @ alloc3  # what the kernel's motor stage computed for the step that led here, at most
    a0 = abs(po.g) + 1.5 * (np.abs(alloc3).sum() * po.max_rpm ** 2 + thrust_now) / po.mass + a_ext
    drag = abs(po.air_resistance_coeff * np.pi * po.arm_length * po.arm_length) / po.mass
    x0, vn = st["x"].copy(), np.linalg.norm(st["v"], axis=1)
    worst = 0.0
    for h in range(1, hmax + 1):
        o.step_n(dt, 1, 8)
        moved = np.linalg.norm(o.get_state()["x"] - x0, axis=1)
        hdt = h * dt
        bound = hdt * vn + 0.5 * hdt * hdt * (a0 + drag * (vn + hdt * a0) ** 2)
        assert np.all(moved <= bound), (airframe, dt, h, float((moved - bound).max()))
        worst = max(worst, float((moved / bound).max()))
    assert 0.2 < worst <= 1.0  # the bound is not vacuous: some UAV uses a good part of it
