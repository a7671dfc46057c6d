"""Committed fixtures (tests/golden/, made by tests/golden/make_golden.py):
nanoflann_radius_sets.npz holds outputs of the REFERENCE's own kd-tree; oracle_trajectories.npz freezes the oracle."""
import os

import numpy as np
import pytest

import helpers

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CLOUDS = ["dense", "sparse", "grid12", "tmux400"]


def expected_forces(pts, off, idx, d2, arm, prop, mass, rebounce=100.0):
    n = len(pts)
    f = np.zeros((n, 3))
    crashed = np.zeros(n, dtype=np.int32)
    for i in range(n):
        for j, d in zip(idx[off[i]:off[i + 1]], d2[off[i]:off[i + 1]]):
            if j == i:
                continue
            if d < ((arm + prop) + arm) + prop:
                crashed[j] = 1
                rel = pts[i] - pts[j]
                nn = np.sqrt((rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2])
                if nn > 0:
                    rel = rel / nn
                f[i] += ((rebounce * rel) * mass) * (mass / (mass + mass))
    return f, crashed


@pytest.mark.parametrize("name", CLOUDS)
def test_oracle_collisions_against_reference_kdtree_fixture(oracle, name):
    g = np.load(os.path.join(G, "nanoflann_radius_sets.npz"))
    pts, off, idx, d2 = (g[f"{name}_{k}"] for k in ("points", "offsets", "indices", "d2"))
    p = helpers.oracle_params("x500")
    f_exp, c_exp = expected_forces(pts, off, idx, d2, p.arm_length, p.prop_radius, p.mass)
    s = oracle.OracleSwarm(len(pts))
    s.construct(0, len(pts), p, pts, np.zeros(len(pts)))
    s.handle_collisions(True, False, 100.0)
    helpers.assert_close(s.get_external_force(), f_exp, 1e-14, name)
    s.handle_collisions(True, True, 100.0)
    assert np.array_equal(s.has_crashed(), c_exp)


def test_oracle_regression_trajectories(oracle):
    g = np.load(os.path.join(G, "oracle_trajectories.npz"))
    O = oracle
    p = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    s = O.OracleSwarm(1)
    s.construct(0, 1, p, [[10, 15, 0]], [3.14])
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(s, nm)(0, 1)
    s.set_input(0, 1, O.ACTUATOR_CMD, [[0.0] * 4])
    s.step_n(0.01, 2)
    s.set_input(0, 1, O.POSITION_CMD, [[12, 13, 5, 1.0]])
    for k in range(20):
        s.step_n(0.001, 1000)
        st = s.get_state()
        row = np.concatenate([st["x"][0], st["v"][0], st["R"][0].ravel(), st["omega"][0], st["motor_rpm"][0, :4], s.get_imu()[0]])
        helpers.assert_close(row, g["config1_every_1000_steps"][k], 1e-9, f"config 1 step {1000 * (k + 1)}")


@pytest.mark.gpu
@pytest.mark.parametrize("name", CLOUDS)
def test_gpu_collisions_against_reference_kdtree_fixture(mrs, name):
    g = np.load(os.path.join(G, "nanoflann_radius_sets.npz"))
    pts, off, idx, d2 = (g[f"{name}_{k}"] for k in ("points", "offsets", "indices", "d2"))
    p = mrs.model_params("x500")
    f_exp, c_exp = expected_forces(pts, off, idx, d2, p.arm_length, p.prop_radius, p.mass)
    s = mrs.Swarm(len(pts))
    s.construct(0, len(pts), p, pts, np.zeros(len(pts)))
    s.handle_collisions(True, False, 100.0)
    helpers.assert_close(s.get_external_force(), f_exp, 1e-13, name)
    s.handle_collisions(True, True, 100.0)
    assert np.array_equal(s.has_crashed(), c_exp)


@pytest.mark.gpu
@pytest.mark.parametrize("arith,rtol", [(0, 1e-9), (1, 1e-6)])
def test_gpu_mixed_swarm_against_frozen_oracle_vectors(mrs, arith, rtol):
    """64 UAVs, four airframes, five input modes, 200 steps — inputs and expected outputs from the fixture only."""
    g = np.load(os.path.join(G, "oracle_trajectories.npz"))
    n = 64
    names = ["x500", "f550", "naki", "t650"]
    s = mrs.Swarm(n, arith=arith)
    for i in range(n):
        s.construct(i, 1, mrs.model_params(names[i % 4], ground_enabled=True), [g["mixed_x0"][i]], [0.0])
    s.set_state(0, n, g["mixed_x0"], g["mixed_v0"], g["mixed_R0"], g["mixed_w0"], g["mixed_rpm0"])
    for i in range(n):
        s.set_input(i, 1, int(g["mixed_modes"][i]), g["mixed_payloads"][i][None, :])
    s.step_n(0.001, 200)
    st = s.get_state()
    for k, key in (("x", "mixed_x"), ("v", "mixed_v"), ("R", "mixed_R"), ("omega", "mixed_w"), ("motor_rpm", "mixed_rpm")):
        helpers.assert_close(st[k], g[key], rtol, key)
    helpers.assert_close(s.get_imu(), g["mixed_imu"], rtol * 10, "imu")


def _run_all_modes(make_swarm, params_of, g):
    """replays tests/golden/all_modes_trajectories.npz on an oracle or product swarm: inputs from the fixture only"""
    n = len(g["modes"])
    s = make_swarm(n)
    for i in range(n):
        s.construct(i, 1, params_of(str(g["frames"][i])), [g["x0"][i]], [0.0])
        for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
            getattr(s, nm)(i, 1)
    s.set_state(0, n, g["x0"], g["v0"], g["R0"], g["w0"], g["rpm0"])
    for i in range(n):
        s.set_input(i, 1, int(g["modes"][i]), g["payloads"][i][None, :])
        if int(g["ff_kind"][i]) >= 0:
            s.set_feedforward(i, 1, int(g["ff_kind"][i]), g["ff_payload"][i][None, :])
    s.step_n(0.001, 150)
    return s


def test_oracle_all_modes_regression(oracle):
    """every input mode x {x500, f550, naki}: the oracle against its own frozen vectors (SURVEY 7, step 2 iv)"""
    g = np.load(os.path.join(G, "all_modes_trajectories.npz"))
    assert sorted(set(g["modes"].tolist())) == list(range(oracle.ACTUATOR_CMD, oracle.POSITION_CMD + 1)) and set(g["frames"].tolist()) == {"x500", "f550", "naki"}
    s = _run_all_modes(oracle.OracleSwarm, lambda f: helpers.oracle_params(f, ground_enabled=True, ground_z=0.0), g)
    st = s.get_state()
    for k, key in (("x", "x"), ("v", "v"), ("R", "R"), ("omega", "w"), ("motor_rpm", "rpm")):
        helpers.assert_close(st[k], g[key], 1e-12, key)
    helpers.assert_close(s.get_pid(), g["pid"], 1e-12, "pid")


@pytest.mark.gpu
@pytest.mark.parametrize("arith,rtol", [(0, 1e-10), (1, 1e-6)])
def test_gpu_all_modes_against_frozen_oracle_vectors(mrs, arith, rtol):
    """every input mode x {4, 6, 8 motors} x feed-forward slots, 150 steps — inputs and expected outputs from the fixture only"""
    g = np.load(os.path.join(G, "all_modes_trajectories.npz"))
    s = _run_all_modes(lambda n: mrs.Swarm(n, arith=arith), lambda f: mrs.model_params(f, ground_enabled=True, ground_z=0.0), g)
    st = s.get_state()
    for k, key in (("x", "x"), ("v", "v"), ("R", "R"), ("omega", "w"), ("motor_rpm", "rpm")):
        helpers.assert_close(st[k], g[key], rtol, key)
    helpers.assert_close(s.get_imu(), g["imu"], rtol * 10, "imu")
    helpers.assert_close(s.get_pid(), g["pid"], rtol * 100, "pid")
