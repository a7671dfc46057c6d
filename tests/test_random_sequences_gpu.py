"""Differential test of the whole ABI surface: random sequences of calls (commands in every mode, feed-forwards, parameter
changes, crashes, forces, teleports, hold, timeouts, steps with and without fused sub-steps, collision ticks in both modes) are
applied to the oracle and to the product; state, PIDs, IMU, forces and crash flags must agree after every few calls.
Catches host-side bookkeeping errors (kernel-variant choice, type interning, neighbour-list invalidation, flag words)."""
import numpy as np
import pytest

import helpers
from helpers import RTOL_LITERAL, Pair

DT = 0.001
WIDTH = {1: 4, 2: 4, 3: 4, 4: 10, 5: 5, 6: 4, 7: 4, 8: 4, 9: 4, 10: 4}  # payload width per input mode (x500: 4 motors)


def rng_range(rng, n):
    a, b = sorted(rng.integers(0, n + 1, 2))
    if a == b:
        b = min(n, a + 1)
        a = b - 1
    return int(a), int(b - a)


def payload(rng, mode, count, x):
    if mode == 1:
        return rng.uniform(0.2, 0.8, (count, 8))  # 8 columns serve every airframe (the first n_motors are used)
    if mode == 2:  # control group: roll pitch yaw throttle
        return np.concatenate([rng.uniform(-0.2, 0.2, (count, 3)), rng.uniform(0.3, 0.7, (count, 1))], axis=1)
    if mode == 3:  # attitude rate + throttle
        return np.concatenate([rng.uniform(-1, 1, (count, 3)), rng.uniform(0.3, 0.7, (count, 1))], axis=1)
    if mode == 4:  # attitude matrix + throttle
        return np.concatenate([helpers.tilted_rotations(rng, count).reshape(count, 9), rng.uniform(0.3, 0.7, (count, 1))], axis=1)
    if mode == 5:  # tilt vector, heading rate, throttle
        t = rng.normal(0, 0.2, (count, 3)) + [0, 0, 1]
        return np.concatenate([t, rng.uniform(-1, 1, (count, 1)), rng.uniform(0.3, 0.7, (count, 1))], axis=1)
    if mode in (6, 7):  # acceleration + heading (rate)
        return np.concatenate([rng.uniform(-2, 2, (count, 3)), rng.uniform(-1, 1, (count, 1))], axis=1)
    if mode in (8, 9):  # velocity + heading (rate)
        return np.concatenate([rng.uniform(-3, 3, (count, 3)), rng.uniform(-1, 1, (count, 1))], axis=1)
    return np.concatenate([x + rng.uniform(-3, 3, (count, 3)), rng.uniform(-3, 3, (count, 1))], axis=1)  # position


@pytest.mark.gpu
@pytest.mark.parametrize("seed,fast,fleet", [(s, s % 4 == 3, "mixed" if s % 2 else "x500") for s in range(24)])
def test_random_call_sequences_match_oracle(mrs, oracle, seed, fast, fleet):
    rng = np.random.default_rng(1000 + seed)
    n = 150
    p = Pair(mrs, n, arith=mrs.ARITH_FAST if fast else mrs.ARITH_LITERAL)
    rtol = 1e-7 if fast else RTOL_LITERAL  # FAST: per-step 1e-13, amplified by PID derivative terms and collisions over ~200 steps
    pos = rng.uniform(0, 9, (n, 3)) + [0, 0, 0.5]  # dense enough for collisions to happen all the time
    if fleet == "x500":
        p.construct(0, n, "x500", pos=pos, heading=rng.uniform(-3, 3, n), ground_enabled=True, ground_z=0.0,
                    takeoff_patch_enabled=bool(seed % 2))
    else:  # three airframes (4 and 6 motors), boundaries inside 64-UAV blocks: uniform and mixed blocks side by side
        for (lo, hi), frame in (((0, 70), "x500"), ((70, 110), "f550"), ((110, 150), "t650")):
            p.construct(lo, hi - lo, frame, pos=pos[lo:hi], heading=rng.uniform(-3, 3, hi - lo), ground_enabled=True, ground_z=0.0,
                        takeoff_patch_enabled=bool(seed % 2))
    p.both("set_input", 0, n, oracle.POSITION_CMD, payload(rng, 10, n, pos))
    ops = 0
    for it in range(70):
        op = rng.integers(0, 18)
        first, count = rng_range(rng, n)
        x = p.o.get_state(first, count)["x"]
        if op <= 2:
            mode = int(rng.integers(0, 11))
            if mode == 0:
                p.both("set_input", first, min(count, 5), 0, None)  # INPUT_UNKNOWN: zero actuators
            else:
                p.both("set_input", first, count, mode, payload(rng, mode, count, x))
        elif op == 3:
            kind = int(rng.integers(0, 4))
            p.both("set_feedforward", first, count, kind, np.concatenate([rng.uniform(-0.5, 0.5, (count, 3)), rng.uniform(-0.2, 0.2, (count, 1))], axis=1))
        elif op == 4:
            p.both("crash", first, min(count, 3))
        elif op == 5:
            p.both("apply_force", first, count, rng.normal(0, 3, (count, 3)))
        elif op == 6:
            st = p.o.get_state(first, count)
            st["x"] = st["x"] + rng.normal(0, 0.5, (count, 3))
            st["v"] = st["v"] + rng.normal(0, 0.5, (count, 3))
            p.both("set_state", first, count, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        elif op == 7:
            p.both("set_mass", first, min(count, 10), float(rng.uniform(1.5, 3.0)))
        elif op == 8:
            p.both("set_ground_z", first, count, float(rng.uniform(-0.5, 0.3)))
        elif op == 9:
            p.both("timeout_input", first, count)
        elif op == 10:
            p.both("set_hold", first, count, bool(rng.integers(0, 2)))
        elif op == 11:
            which = ("set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params")[int(rng.integers(0, 4))]
            p.both(which, first, count, kp=float(rng.uniform(1.0, 5.0)))
        elif op == 12:
            k, sub = int(rng.integers(1, 6)), int(rng.integers(1, 3))
            p.o.step_n(DT, k * sub)
            p.g.step_n(DT, k * sub, sub)  # n_steps is the total; sub of them are fused per launch
        elif op == 13:
            crash = bool(rng.integers(0, 4) == 0)
            p.both("handle_collisions", not crash or bool(rng.integers(0, 2)), crash, 100.0)
        elif op == 14:
            k = int(rng.integers(1, 8))
            for _ in range(k):
                p.o.step(DT)
                p.o.handle_collisions(True, False, 60.0)
            p.g.tick_n(DT, k, True, False, 60.0)
        elif op == 15:  # the staged upload path against the plain one
            mode = int(rng.integers(1, 11))
            pl = payload(rng, mode, count, x)
            p.o.set_input(first, count, mode, pl)
            rows = p.g.input_staging(count, pl.shape[1])
            rows[:] = pl
            p.g.commit_input(first, count, mode, pl.shape[1])
        elif op == 16:  # publisher payloads
            a, b = p.g.get_outputs_view(first, count), p.o.get_outputs(first, count)
            for f in a.dtype.names:
                av, bv = a[f].copy(), b[f].copy()
                if f == "range" and fast:
                    # acos(R22) of a level UAV is NaN or 0 depending on the last bit of R22 (the reference's formula, kept):
                    # FAST states differ from the oracle's in exactly those bits
                    both = np.isfinite(av) & np.isfinite(bv)
                    av, bv = av[both], bv[both]
                helpers.assert_close(av, bv, max(rtol, 1e-9), f"seed {seed}: output {f}")
        else:  # UavSystem::setParams with another airframe of the same motor count (the reference keeps motor_rpm's size, :374-377)
            for uav in range(first, first + min(count, 5)):
                nm = p.o.get_params(uav).n_motors
                frame = {4: ("x500", "t650", "a300", "f450"), 6: ("f550",), 8: ("naki",)}[nm]
                po = helpers.oracle_params(frame[int(rng.integers(0, len(frame)))], ground_enabled=True, ground_z=0.0)
                p.o.set_params(uav, 1, po)
                p.g.set_params(uav, 1, helpers.to_product_params(mrs, po))
                xs = p.o.get_state(uav, 1)["x"]
                p.both("set_input", uav, 1, oracle.POSITION_CMD, payload(rng, 10, 1, xs))
        ops += 1
        if it % 5 == 4:
            p.step(DT, 2)
            p.compare(rtol, f"seed {seed}, after {ops} calls")
            helpers.assert_close(p.g.get_external_force(), p.o.get_external_force(), max(rtol, 1e-11), f"seed {seed}: forces after {ops} calls")
            assert np.array_equal(p.g.has_crashed(), p.o.has_crashed()), f"seed {seed}: crash flags after {ops} calls"
    assert p.g.get_diag() == p.o.get_diag()
