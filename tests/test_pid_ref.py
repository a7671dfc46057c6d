"""Row C1 of SURVEY §8(a) pinned against the reference ITSELF: the reference's PIDController (controllers/pid.hpp needs nothing
but <math.h>) is compiled where it lies into oracle/_ref/libref_pid.so (oracle/Makefile), and

* the committed golden vectors tests/golden/pid_reference_vectors.npz (outputs of that class for seeded stimulus, made by
  tests/golden/make_golden.py:pid_reference) must be reproduced bit for bit by the oracle's pid_update (oracle/uav_oracle.c) —
  this part needs neither the reference tree nor the _ref library;
* when the library is there (built here, or travelled to the GPU box as a prebuilt file) it must still produce the golden vectors,
  and fresh random sequences must agree with the oracle bit for bit as well.
Bit-exact: the PID is a handful of IEEE additions, multiplications and one division in a fixed order (pid.hpp:67-96)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle_swarm as O
    O.build()
    return O


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "pid_reference_vectors.npz"))


def oracle_run(O, params, err, dt, event, new_sat):
    """the oracle's PID over one sequence; its state lives in two doubles like PIDController's members"""
    kp, kd, ki, sat, aw = (float(v) for v in params)
    le, integ = C.c_double(0.0), C.c_double(0.0)  # PIDController(): reset()
    out = np.zeros(len(err))
    for k in range(len(err)):
        if event[k] == 1:  # reset(): last_error = integral = 0 (pid.hpp:61-65)
            le.value, integ.value = 0.0, 0.0
        elif event[k] == 2:  # setSaturation (pid.hpp:56-59)
            sat = float(new_sat[k])
        out[k] = O.lib().orc_pid_update(kp, kd, ki, sat, aw, C.byref(le), C.byref(integ), float(err[k]), float(dt[k]))
    return out


def assert_same_bits(got, exp, what):
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    same = (got.view(np.uint64) == exp.view(np.uint64)) | (np.isnan(got) & np.isnan(exp))
    assert same.all(), f"{what}: first difference at step {int(np.flatnonzero(~same)[0])}: {got[~same][0]!r} vs {exp[~same][0]!r}"


def test_oracle_pid_reproduces_the_reference_vectors(oracle, golden):
    g = golden
    assert g["out"].shape == (96, 160)
    for q in range(len(g["params"])):
        got = oracle_run(oracle, g["params"][q], g["err"][q], g["dt"][q], g["event"][q], g["new_sat"][q])
        assert_same_bits(got, g["out"][q], f"sequence {q} (kp, kd, ki, saturation, antiwindup = {g['params'][q]})")


def test_vectors_exercise_every_branch(golden):
    """the stimulus reaches both saturation sides, the unsaturated range, the no-saturation and no-anti-windup
    parameterisations, NaN and infinite errors"""
    g = golden
    sat = g["params"][:, 3]
    hit_hi = hit_lo = inside = 0
    for q in range(len(sat)):
        if sat[q] > 0 and not g["event"][q].any():
            hit_hi += int((g["out"][q] == sat[q]).sum())
            hit_lo += int((g["out"][q] == -sat[q]).sum())
            inside += int((np.abs(g["out"][q]) < sat[q]).sum())
    assert hit_hi > 50 and hit_lo > 50 and inside > 500, (hit_hi, hit_lo, inside)
    assert (sat <= 0).sum() >= 10 and (g["params"][:, 4] <= 0).sum() >= 10
    assert np.isnan(g["err"]).sum() >= 5 and np.isinf(g["err"]).sum() >= 5 and np.isnan(g["out"]).sum() >= 10


def test_reference_library_still_produces_the_vectors_and_matches_fresh_sequences(oracle, golden):
    L = oracle.ref_pid_lib()
    if L is None:
        pytest.skip("oracle/_ref/libref_pid.so not built (reference tree absent and no prebuilt library)")
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden

    def ref_run(params, err, dt, event, new_sat):
        h = L.ref_pid_create()
        L.ref_pid_set_params(h, *[float(v) for v in params])
        out = np.zeros(len(err))
        for k in range(len(err)):
            if event[k] == 1:
                L.ref_pid_reset(h)
            elif event[k] == 2:
                L.ref_pid_set_saturation(h, float(new_sat[k]))
            out[k] = L.ref_pid_update(h, float(err[k]), float(dt[k]))
        L.ref_pid_destroy(h)
        return out

    g = golden
    for q in range(0, len(g["params"]), 7):
        assert_same_bits(ref_run(g["params"][q], g["err"][q], g["dt"][q], g["event"][q], g["new_sat"][q]), g["out"][q], f"golden {q}")
    params, err, dt, event, new_sat = make_golden.pid_sequences(seed=4242, n_seq=40, n_steps=120)
    for q in range(len(params)):
        assert_same_bits(oracle_run(oracle, params[q], err[q], dt[q], event[q], new_sat[q]),
                         ref_run(params[q], err[q], dt[q], event[q], new_sat[q]), f"fresh sequence {q}")
