"""GPU parity tests proper: the HIP path (through the C ABI, via mrs_multirotor_simulator_amd.Swarm) against the
CPU oracle on identical seeded inputs.  Tolerance of BASELINE.json's north_star: state L-inf <= 1e-6 relative (FP64);
the LITERAL kernel is additionally held to 1e-11 (it repeats the reference's operation order without FMA contraction)."""
import os

import numpy as np
import pytest

import helpers
from helpers import RTOL_FAST, RTOL_LITERAL, RTOL_NORTH_STAR, Pair, random_state

pytestmark = pytest.mark.gpu
DT = 0.001


@pytest.fixture(scope="module")
def M(mrs):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return mrs


def goals(rng, n):
    return np.concatenate([rng.uniform(-40, 40, (n, 2)), rng.uniform(2, 20, (n, 1)), rng.uniform(-3.14, 3.14, (n, 1))], axis=1)


def payload_for(O, mode, rng, n, n_motors, st):
    if mode == O.ACTUATOR_CMD:
        return rng.uniform(0.35, 0.60, (n, n_motors))
    if mode == O.CONTROL_GROUP_CMD:
        return np.concatenate([rng.uniform(-0.1, 0.1, (n, 3)), rng.uniform(0.3, 0.7, (n, 1))], axis=1)
    if mode == O.ATTITUDE_RATE_CMD:
        return np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.uniform(0.3, 0.7, (n, 1))], axis=1)
    if mode == O.ATTITUDE_CMD:
        Rd = helpers.tilted_rotations(rng, n, 0.4).reshape(n, 9)
        return np.concatenate([Rd, rng.uniform(0.3, 0.7, (n, 1))], axis=1)
    if mode == O.TILT_HDG_RATE_CMD:
        tilt = helpers.tilted_rotations(rng, n, 0.4)[:, :, 2] * rng.uniform(0.5, 3.0, (n, 1))
        return np.concatenate([tilt, rng.uniform(-1, 1, (n, 1)), rng.uniform(0.3, 0.7, (n, 1))], axis=1)
    if mode in (O.ACCELERATION_HDG_RATE_CMD, O.ACCELERATION_HDG_CMD):
        return np.concatenate([rng.uniform(-2, 2, (n, 3)), rng.uniform(-1, 1, (n, 1))], axis=1)
    if mode in (O.VELOCITY_HDG_RATE_CMD, O.VELOCITY_HDG_CMD):
        return np.concatenate([rng.uniform(-3, 3, (n, 3)), rng.uniform(-1, 1, (n, 1))], axis=1)
    if mode == O.POSITION_CMD:
        return np.concatenate([st["x"] + rng.uniform(-5, 5, (n, 3)), rng.uniform(-3.14, 3.14, (n, 1))], axis=1)
    return None


# ------------------------------------------------------------------------------------------------------------------
def test_config3_actuator_single_step_random_states(M, oracle):
    """BASELINE config 3 generator at an oracle-sized N: one step from random states, actuator references."""
    rng = np.random.default_rng(3)
    n = 4096
    p = Pair(M, n)
    p.construct(0, n, "x500")
    p.set_state(0, n, random_state(rng, n, 4))
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, rng.uniform(0.35, 0.60, (n, 4)))
    p.step(DT)
    worst = p.compare(RTOL_LITERAL, "1 step")
    p.step(DT, 100)
    worst = max(worst, p.compare(RTOL_LITERAL, "101 steps"))
    assert worst <= RTOL_NORTH_STAR


@pytest.mark.parametrize("airframe", ["x500", "f550", "naki"])
@pytest.mark.parametrize("mode", list(range(0, 11)))
def test_every_input_mode(M, oracle, mode, airframe):
    """All 11 INPUT_MODEs x {4,6,8} motors: 40 steps from random near-upright states."""
    rng = np.random.default_rng(100 + mode)
    n = 320  # five wavefronts
    nm = M.AIRFRAMES[airframe]["n_motors"]
    p = Pair(M, n)
    p.construct(0, n, airframe)
    st = random_state(rng, n, nm, tilted=True)
    p.set_state(0, n, st)
    p.both("set_input", 0, n, mode, payload_for(oracle, mode, rng, n, nm, st))
    p.step(DT)
    p.compare(RTOL_LITERAL, f"mode {mode} step 1")
    p.step(DT, 39)
    p.compare(RTOL_LITERAL, f"mode {mode} step 40")
    assert p.g.get_diag() == p.o.get_diag()


@pytest.mark.parametrize("mode,kinds", [(10, (1,)), (10, (0,)), (10, (0, 1, 2, 3)), (9, (3,)), (9, (2,)), (8, (2,)), (8, (3,)), (8, (2, 3))])
def test_feedforward_slots(M, oracle, mode, kinds):
    """std::optional feed-forwards and their priority order (uav_system.hpp:318-346)."""
    rng = np.random.default_rng(7 + mode + sum(kinds))
    n = 128
    p = Pair(M, n)
    p.construct(0, n, "x500")
    st = random_state(rng, n, 4, tilted=True)
    p.set_state(0, n, st)
    for k in kinds:
        p.both("set_feedforward", 0, n, k, rng.uniform(-1, 1, (n, 4)))
    p.both("set_input", 0, n, mode, payload_for(oracle, mode, rng, n, 4, st))
    p.step(DT, 25)
    p.compare(RTOL_LITERAL, f"ff {kinds} mode {mode}")


@pytest.mark.parametrize("fast", [False, True])
def test_config1_single_uav_trajectory(M, oracle, fast):
    """BASELINE config 1: 1 x500, spawn (10,15,0) hdg 3.14, two warm-up steps of 0.01 s, POSITION_CMD (12,13,5,1.0),
    20 000 steps of 1 ms.  A swarm of one goes through the same kernel; both arithmetic flavours stay inside the north-star
    tolerance over the whole closed-loop flight."""
    p = Pair(M, 1, arith=M.ARITH_FAST if fast else M.ARITH_LITERAL)
    p.construct(0, 1, "x500", pos=[[10, 15, 0]], heading=[3.14], ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    p.both("set_input", 0, 1, oracle.ACTUATOR_CMD, [[0.0] * 4])
    p.step(0.01, 2)
    p.compare(RTOL_FAST if fast else RTOL_LITERAL, "warm-up")
    p.both("set_input", 0, 1, oracle.POSITION_CMD, [[12, 13, 5, 1.0]])
    for k in range(20):
        p.step(DT, 1000)
        p.compare(RTOL_NORTH_STAR, f"step {1000 * (k + 1)}")
    assert np.allclose(p.g.get_state()["x"][0], [11.997797, 13.003830, 4.995801], atol=2e-6)


def test_config2_400_hexarotors(M, oracle):
    """BASELINE config 2: 400 f550 on a 20x20 grid (4 m pitch), POSITION_CMD to goto.py-style goals, ground on."""
    rng = np.random.default_rng(400)
    n = 400
    gx, gy = np.meshgrid(np.arange(20) * 4.0, np.arange(20) * 4.0, indexing="ij")
    pos = np.stack([gx.ravel(), gy.ravel(), np.zeros(n)], axis=1)
    p = Pair(M, n)
    p.construct(0, n, "f550", pos=pos, heading=np.zeros(n), ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, np.zeros((n, 6)))
    p.step(0.01, 2)
    p.both("set_input", 0, n, oracle.POSITION_CMD, goals(rng, n))
    for k in range(4):
        p.step(DT, 500)
        p.compare(RTOL_NORTH_STAR, f"step {500 * (k + 1)}")


def test_heterogeneous_swarm_and_ragged_tail(M, oracle):
    """Mixed airframes inside one wavefront (type waterfall) and a swarm size that is not a multiple of 64."""
    rng = np.random.default_rng(11)
    n = 203
    p = Pair(M, n)
    names = ["x500", "f550", "naki", "t650", "a300"]
    st_all = {}
    for i in range(n):
        af = names[i % len(names)] if i < 150 else "robofly"
        p.construct(i, 1, af, pos=[rng.uniform(-50, 50, 3) + [0, 0, 100]], heading=[rng.uniform(-3, 3)])
        nm = M.AIRFRAMES[af]["n_motors"]
        st = random_state(rng, 1, nm, tilted=True)
        p.set_state(i, 1, st)
        st_all[i] = st
        mode = [oracle.POSITION_CMD, oracle.ACTUATOR_CMD, oracle.VELOCITY_HDG_RATE_CMD, oracle.ATTITUDE_RATE_CMD][i % 4]
        p.both("set_input", i, 1, mode, payload_for(oracle, mode, rng, 1, nm, st))
    p.step(DT, 30)
    p.compare(RTOL_LITERAL, "heterogeneous")


def test_substep_fusion_is_bit_identical(M, oracle):
    rng = np.random.default_rng(5)
    n = 512
    sw = [M.Swarm(n), M.Swarm(n)]
    st = random_state(rng, n, 4, tilted=True)
    g = goals(rng, n)
    for s in sw:
        s.construct(0, n, M.model_params("x500"))
        s.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        s.set_input(0, n, M.POSITION_CMD, g)
    sw[0].step_n(DT, 24, 1)
    sw[1].step_n(DT, 24, 8)
    a, b = sw[0].get_state(), sw[1].get_state()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(sw[0].get_pid(), sw[1].get_pid()) and np.array_equal(sw[0].get_imu(), sw[1].get_imu())


def test_fast_arithmetic_within_tolerance(M, oracle):
    rng = np.random.default_rng(9)
    n = 2048
    p = Pair(M, n, arith=M.ARITH_FAST)
    p.construct(0, n, "x500")
    st = random_state(rng, n, 4, tilted=True)
    p.set_state(0, n, st)
    p.both("set_input", 0, n, oracle.POSITION_CMD, payload_for(oracle, oracle.POSITION_CMD, rng, n, 4, st))
    p.step(DT)
    p.compare(RTOL_FAST, "fast 1 step")
    p.step(DT, 99)
    p.compare(RTOL_NORTH_STAR, "fast 100 steps")


def test_crash_force_ground_takeoff_and_nan_guards(M, oracle):
    rng = np.random.default_rng(21)
    n = 256
    p = Pair(M, n)
    p.construct(0, n, "x500", pos=np.concatenate([rng.uniform(-20, 20, (n, 2)), rng.uniform(0.0, 0.05, (n, 1))], axis=1),
                heading=rng.uniform(-3, 3, n), ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=True)
    thr = rng.uniform(0.0, 1.0, (n, 1)) * np.ones((1, 4))
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, thr)
    p.both("crash", 10, 20)
    p.both("apply_force", 40, 50, rng.normal(0, 5, (50, 3)))
    p.step(DT, 300)  # some lift off (patch disabled), some stay clamped
    p.compare(RTOL_LITERAL, "ground/takeoff")
    assert np.array_equal(p.g.has_crashed(), p.o.has_crashed())
    assert [p.g.get_params(i).takeoff_patch_enabled for i in range(n)] == [p.o.get_params(i).takeoff_patch_enabled for i in range(n)]
    assert np.array_equal(p.g.get_external_force(), p.o.get_external_force())
    # NaN guards: non-finite actuators -> 0; NaN throttle from an inverted attitude; NaN/inf state -> rollback
    p.both("set_input", 0, 8, oracle.ACTUATOR_CMD, np.full((8, 4), np.nan))
    p.both("set_input", 8, 8, oracle.ACTUATOR_CMD, np.full((8, 4), np.inf))
    st = p.o.get_state(16, 8)
    st["R"][:] = np.diag([1.0, -1.0, -1.0])
    p.both("set_state", 16, 8, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    p.both("set_input", 16, 8, oracle.ACCELERATION_HDG_CMD, np.tile([0, 0, 0, 0.3], (8, 1)))
    st = p.o.get_state(24, 4)
    st["v"][:, 0] = [np.nan, np.inf, 1e200, -np.inf]
    p.both("set_state", 24, 4, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    p.step(DT, 5)
    p.compare(RTOL_LITERAL, "nan guards")
    assert p.g.get_diag() == p.o.get_diag() and p.o.get_diag()["nan_rollback"] > 0


def test_set_params_resets_controllers(M, oracle):
    """set_mass semantics (src/uav_system_ros.cpp:1028-1053): new ModelParams -> default gains, fresh PIDs."""
    rng = np.random.default_rng(33)
    n = 64
    p = Pair(M, n)
    po = p.construct(0, n, "x500")
    p.both("set_position_params", 0, n, 3.0, 0.2, 0.1, 5.0)
    p.both("set_rate_params", 0, n, 5.0, 0.05, 0.01)
    st = random_state(rng, n, 4, tilted=True)
    p.set_state(0, n, st)
    p.both("set_input", 0, n, oracle.POSITION_CMD, payload_for(oracle, oracle.POSITION_CMD, rng, n, 4, st))
    p.step(DT, 20)
    p.compare(RTOL_LITERAL, "custom gains")
    # mass change on half of the swarm, the way callbackSetMass does it
    po2 = oracle.ModelParams.from_buffer_copy(bytes(po))
    old = po2.mass
    po2.mass = 2.6
    for m in range(4):
        po2.allocation_matrix[2 * 8 + m] = po2.mass * (po2.allocation_matrix[2 * 8 + m] / old)
    oracle.lib().orc_calculate_inertia(po2)
    p.o.set_params(0, n // 2, po2)
    p.g.set_params(0, n // 2, helpers.to_product_params(M, po2))
    assert np.all(p.g.get_pid(0, n // 2) == 0)
    p.step(DT, 20)
    p.compare(RTOL_LITERAL, "after set_params")
    assert p.g.get_params(0).mass == 2.6 and p.g.get_params(n - 1).mass == 2.0


@pytest.mark.parametrize("crash", [False, True])
def test_collisions_match_oracle(M, oracle, crash):
    """handleCollisions on a dense random cloud: forces / crash flags against the oracle, including the
    squared-distance-vs-metres criterion and heterogeneous airframes."""
    rng = np.random.default_rng(77)
    n = 3000
    p = Pair(M, n)
    pos = rng.uniform(0, 14, (n, 3))  # ~1 UAV per m^3: many pairs inside crit (~0.8 m^2)
    p.construct(0, n // 2, "x500", pos=pos[: n // 2], heading=np.zeros(n // 2))
    p.construct(n // 2, n - n // 2, "t650", pos=pos[n // 2:], heading=np.zeros(n - n // 2))
    p.both("handle_collisions", True, crash, 100.0)
    fo, fg = p.o.get_external_force(), p.g.get_external_force()
    assert np.array_equal(p.g.has_crashed(), p.o.has_crashed())
    if crash:
        assert p.o.has_crashed().sum() > 100 and np.all(fg == 0)
    else:
        assert (np.abs(fo).sum(axis=1) > 0).sum() > 100
        helpers.assert_close(fg, fo, 1e-14, "forces")
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, np.full((n, 4), 0.5))
    p.step(DT, 3)
    p.compare(RTOL_LITERAL, "step after collisions")
    # disabled: early return leaves forces untouched (src/multirotor_simulator.cpp:299-301)
    p.both("handle_collisions", False, False, 100.0)
    assert np.array_equal(p.g.get_external_force(), fg)


def test_collision_neighbour_set_vs_reference_kdtree(M, oracle):
    """Pairs found by the GPU hash == pairs found by the reference's own nanoflann (oracle/_ref), via the force support."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    n = 5000
    pos = rng.uniform(0, 30, (n, 3))
    pos[:50] = pos[50:100] + rng.normal(0, 0.05, (50, 3))  # guaranteed close pairs
    pos[100] = pos[101]  # coincident pair: normalized(0) = 0 -> zero force, still a neighbour
    g = M.Swarm(n)
    g.construct(0, n, M.model_params("x500"), pos, np.zeros(n))
    g.handle_collisions(True, False, 100.0)
    f = g.get_external_force()
    off, idx, d2 = oracle.ref_radius_neighbours(pos, 3.0, 10)
    crit = 2 * (0.25 + 0.15)
    expect = np.zeros((n, 3))
    for i in range(n):
        js = idx[off[i]:off[i + 1]]
        dd = d2[off[i]:off[i + 1]]
        for j, d in sorted(zip(js, dd)):
            if j != i and d < crit:
                rel = pos[i] - pos[j]
                nn = np.sqrt((rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2])
                if nn > 0:
                    rel = rel / nn
                expect[i] += 100.0 * rel * 2.0 * (2.0 / (2.0 + 2.0))
    assert (np.abs(expect).sum(axis=1) > 0).sum() >= 50
    helpers.assert_close(f, expect, 1e-13, "forces vs reference kd-tree neighbours")


def test_collisions_three_body_clusters_in_the_reference_traversal_order(M, oracle):
    """UAVs with THREE OR MORE partners: the reference adds the rebounce forces in the order its kd-tree hands the neighbours out
    (src/multirotor_simulator.cpp:330-353), this library in ascending partner index.  The two sums differ by rounding only: held
    here against forces accumulated in the reference's OWN traversal order (oracle/_ref: the reference's nanoflann, unsorted
    results), at a few ulps of the force scale.  Mixed airframes so that the masses in the force expression differ."""
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(15)
    n_clusters, per = 300, 5
    n = n_clusters * per
    centres = rng.uniform(0, 120, (n_clusters, 3)) + [0, 0, 10]
    pos = (centres[:, None, :] + rng.normal(0, 0.22, (n_clusters, per, 3))).reshape(n, 3)  # five UAVs within ~0.5 m of each other
    kinds = ["x500", "f450", "t650"]
    p = Pair(M, n)
    params = {}
    for k in range(n):
        a = kinds[k % 3]
        params[k] = p.construct(k, 1, a, pos=pos[k:k + 1], heading=np.zeros(1))
    p.both("handle_collisions", True, False, 100.0)
    fg, fo = p.g.get_external_force(), p.o.get_external_force()
    off, idx, d2 = oracle.ref_radius_neighbours(pos, 3.0, 10)
    expect = np.zeros((n, 3))
    multi = 0
    for i in range(n):
        hits = 0
        for j, d in zip(idx[off[i]:off[i + 1]], d2[off[i]:off[i + 1]]):  # the reference's order, as returned
            if j == i:
                continue
            pi, pj = params[i], params[int(j)]
            if d < ((pi.arm_length + pi.prop_radius) + pj.arm_length) + pj.prop_radius:
                rel = pos[i] - pos[j]
                z = (rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2]
                if z > 0:
                    rel = rel / np.sqrt(z)
                expect[i] += ((100.0 * rel) * pi.mass) * (pj.mass / (pi.mass + pj.mass))
                hits += 1
        multi += hits >= 3
    assert multi > 200, "the scenario must hold many UAVs with three or more partners"
    scale = np.abs(expect).max()
    assert np.abs(fg - expect).max() <= 8 * np.finfo(float).eps * scale, "summation order: more than a few ulps"
    helpers.assert_close(fg, fo, 1e-15, "GPU vs oracle (both ascending index)")


def test_timer_main_tick_order(M, oracle):
    """tick_n == {makeStep for all; handleCollisions} repeated (src/multirotor_simulator.cpp:211-217): forces act next tick."""
    rng = np.random.default_rng(8)
    n = 1500
    p = Pair(M, n)
    pos = rng.uniform(0, 12, (n, 3)) + [0, 0, 30]
    p.construct(0, n, "x500", pos=pos, heading=rng.uniform(-3, 3, n))
    p.both("set_input", 0, n, oracle.POSITION_CMD, np.concatenate([pos + rng.uniform(-2, 2, (n, 3)), np.zeros((n, 1))], axis=1))
    for _ in range(25):
        p.o.step(DT)
        p.o.handle_collisions(True, False, 100.0)
    p.g.tick_n(DT, 25, True, False, 100.0)
    p.compare(RTOL_LITERAL, "25 ticks with elastic collisions")
    helpers.assert_close(p.g.get_external_force(), p.o.get_external_force(), 1e-12, "forces")


def test_full_size_100k_sample_against_oracle(M, oracle):
    """BASELINE config 3 at its full size (100 000 UAVs): every UAV is independent, so a random sample of lanes
    stepped alone by the oracle must agree with the same lanes inside the big launch (size-independence property)."""
    rng = np.random.default_rng(3)
    n = 100_000
    g = M.Swarm(n)
    g.construct(0, n, M.model_params("x500", ground_enabled=True))
    st = random_state(rng, n, 4)
    cmd = rng.uniform(0.35, 0.60, (n, 4))
    g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    g.set_input(0, n, M.ACTUATOR_CMD, cmd)
    g.step_n(DT, 50)
    pick = np.sort(rng.choice(n, 2000, replace=False))
    pick[:3] = [0, 63, n - 1]
    o = oracle.OracleSwarm(len(pick))
    o.construct(0, len(pick), helpers.oracle_params("x500", ground_enabled=True))
    o.set_state(0, len(pick), st["x"][pick], st["v"][pick], st["R"][pick], st["omega"][pick], st["motor_rpm"][pick])
    o.set_input(0, len(pick), oracle.ACTUATOR_CMD, cmd[pick])
    o.step_n(DT, 50)
    a, b = g.get_state(), o.get_state()
    for k in b:
        helpers.assert_close(a[k][pick], b[k], RTOL_LITERAL, k)
    assert np.all(np.isfinite(a["x"])) and np.allclose(np.einsum("nij,nik->njk", a["R"], a["R"]), np.eye(3), atol=1e-12)


# ------------------------------------------------------------------------------------------------------------------
# FAST arithmetic (the production flavour bench.py measures): same scenarios, north-star tolerance
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("airframe", ["x500", "f550", "naki"])
@pytest.mark.parametrize("mode", list(range(0, 11)))
def test_every_input_mode_fast_arithmetic(M, oracle, mode, airframe):
    rng = np.random.default_rng(100 + mode)
    n = 192
    nm = M.AIRFRAMES[airframe]["n_motors"]
    p = Pair(M, n, arith=M.ARITH_FAST)
    p.construct(0, n, airframe)
    st = random_state(rng, n, nm, tilted=True)
    p.set_state(0, n, st)
    p.both("set_input", 0, n, mode, payload_for(oracle, mode, rng, n, nm, st))
    p.step(DT)
    p.compare(RTOL_FAST, f"fast mode {mode} step 1")
    p.step(DT, 39)
    p.compare(RTOL_NORTH_STAR, f"fast mode {mode} step 40")


def test_config3_fast_arithmetic_1000_steps(M, oracle):
    """the bench workload itself (config 3 generator), 1000 steps, FAST kernel vs oracle at the north-star tolerance."""
    rng = np.random.default_rng(3)
    n = 2048
    p = Pair(M, n, arith=M.ARITH_FAST)
    p.construct(0, n, "x500", ground_enabled=True)
    p.set_state(0, n, random_state(rng, n, 4))
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, rng.uniform(0.35, 0.60, (n, 4)))
    p.step(DT, 1000)
    p.compare(RTOL_NORTH_STAR, "config 3, 1000 steps, fast")


def test_config2_fast_arithmetic(M, oracle):
    rng = np.random.default_rng(400)
    n = 400
    gx, gy = np.meshgrid(np.arange(20) * 4.0, np.arange(20) * 4.0, indexing="ij")
    pos = np.stack([gx.ravel(), gy.ravel(), np.zeros(n)], axis=1)
    p = Pair(M, n, arith=M.ARITH_FAST)
    p.construct(0, n, "f550", pos=pos, heading=np.zeros(n), ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, np.zeros((n, 6)))
    p.step(0.01, 2)
    p.both("set_input", 0, n, oracle.POSITION_CMD, goals(rng, n))
    p.step(DT, 2000)
    p.compare(RTOL_NORTH_STAR, "config 2, 2000 steps, fast")


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU collision exchange, exercised on one GPU
# ------------------------------------------------------------------------------------------------------------------
def test_gathered_collisions_from_two_virtual_shards(M, oracle):
    """Two shards (two Swarm objects on this GPU) pack their 48-B records, the records are concatenated with NaN padding the
    way ShardedSwarm's all-gather lays them out, and each shard collides against the gathered buffer with its offset."""
    import torch
    from mrs_multirotor_simulator_amd.sharded import max_shard, shard_range
    rng = np.random.default_rng(99)
    n_total, world = 2001, 2
    pos = rng.uniform(0, 13, (n_total, 3))
    o = oracle.OracleSwarm(n_total)
    po = helpers.oracle_params("x500")
    pt = helpers.oracle_params("t650")
    half = shard_range(n_total, world, 0)[1]
    o.construct(0, half, po, pos[:half], np.zeros(half))
    o.construct(half, n_total - half, pt, pos[half:], np.zeros(n_total - half))
    o.handle_collisions(True, False, 100.0)
    n_max = max_shard(n_total, world)
    recv = torch.full((world * n_max, 6), float("nan"), dtype=torch.float64, device="cuda")
    shards = []
    for r in range(world):
        lo, hi = shard_range(n_total, world, r)
        g = M.Swarm(hi - lo)
        g.construct(0, hi - lo, helpers.to_product_params(M, po if r == 0 else pt), pos[lo:hi], np.zeros(hi - lo))
        g.pack_positions_to(recv[r * n_max:].data_ptr())
        g.synchronize()
        shards.append((g, lo, hi))
    torch.cuda.synchronize()
    for r, (g, lo, hi) in enumerate(shards):
        g.handle_collisions_gathered(recv.data_ptr(), world * n_max, r * n_max, True, False, 100.0)
        helpers.assert_close(g.get_external_force(), o.get_external_force(lo, hi - lo), 1e-13, f"shard {r} forces")
    assert (np.abs(o.get_external_force()).sum(axis=1) > 0).sum() > 100


def test_sharded_swarm_world1_on_gpu(M, oracle):
    import torch
    from mrs_multirotor_simulator_amd.sharded import GpuEngine, ShardedSwarm
    rng = np.random.default_rng(17)
    n = 1000
    pos = rng.uniform(0, 10, (n, 3)) + [0, 0, 20]
    p = Pair(M, n)
    p.construct(0, n, "x500", pos=pos, heading=np.zeros(n))
    cmd = np.concatenate([pos + rng.uniform(-2, 2, (n, 3)), np.zeros((n, 1))], axis=1)
    p.both("set_input", 0, n, oracle.POSITION_CMD, cmd)
    sh = ShardedSwarm(n, GpuEngine(p.g, torch.device("cuda", 0)), torch.device("cuda", 0))
    sh.tick_n(DT, 10, True, False, 100.0)
    for _ in range(10):
        p.o.step(DT)
        p.o.handle_collisions(True, False, 100.0)
    p.compare(RTOL_LITERAL, "sharded world=1")


def test_set_state_keeps_v_prev_like_the_reference(M, oracle):
    """MultirotorModel::setState (multirotor_model.hpp:424-433) overwrites v but not v_prev; the step kernel normally
    elides the v_prev column (v_prev == v after every step), so this is the one path where the column matters."""
    rng = np.random.default_rng(123)
    n = 200
    p = Pair(M, n)
    p.construct(0, n, "x500")
    p.set_state(0, n, random_state(rng, n, 4))
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, rng.uniform(0.35, 0.6, (n, 4)))
    p.compare(RTOL_LITERAL, "after first set_state (v_prev = 0)")
    p.step(DT, 3)
    p.compare(RTOL_LITERAL, "stepped")
    assert np.array_equal(p.g.get_state()["v_prev"], p.g.get_state()["v"])
    st2 = random_state(rng, 50, 4)
    p.set_state(40, 50, st2)       # v replaced on a sub-range, v_prev must stay the pre-set_state velocity
    p.compare(RTOL_LITERAL, "after second set_state")
    p.set_state(60, 10, random_state(rng, 10, 4))  # twice in a row: v_prev still the value from before the first one
    p.compare(RTOL_LITERAL, "after third set_state")
    p.step(DT)
    p.compare(RTOL_LITERAL, "IMU uses the kept v_prev")
    p.both("apply_force", 0, 20, rng.normal(0, 3, (20, 3)))  # first force ever applied to this swarm
    p.step(DT, 2)
    p.compare(RTOL_LITERAL, "external force switched on")


def test_c_abi_error_codes(M):
    """The reference's API cannot fail; the C ABI reports misuse through return codes (surfaced as MrsError)."""
    s = M.Swarm(10)
    with pytest.raises(M.MrsError, match="range"):
        s.step_n(DT, 1) or s.get_state(8, 5)
    with pytest.raises(M.MrsError, match="range"):
        s.set_input(5, 6, M.POSITION_CMD, np.zeros((6, 4)))
    with pytest.raises(M.MrsError, match="mode"):
        s.set_input(0, 1, 11, np.zeros((1, 4)))
    with pytest.raises(M.MrsError, match="stride|payload"):
        s.set_input(0, 1, M.ATTITUDE_CMD, np.zeros((1, 4)))
    with pytest.raises(M.MrsError, match="n_motors|narrower"):
        s.construct(0, 1, M.model_params("naki"))
        s.set_input(0, 1, M.ACTUATOR_CMD, np.zeros((1, 4)))
    with pytest.raises(M.MrsError):
        s.step_n(-1.0, 1)
    p = M.model_params("x500")
    p.n_motors = 9
    with pytest.raises(M.MrsError, match="n_motors"):
        s.construct(0, 1, p)
    e = M.Swarm(0)  # empty swarm: everything is a no-op
    e.step_n(DT, 3)
    e.handle_collisions(True, False, 100.0)
    assert e.get_state()["x"].shape == (0, 3)


def test_concurrent_commands_and_steps(M):
    """Subscriber callbacks run concurrently with timerMain in the reference (MT node handle + mutex_uav_system_); here every
    C-ABI call on a swarm is serialised by the library.  ctypes releases the GIL, so the two threads really overlap."""
    import threading
    rng = np.random.default_rng(4)
    n = 2048
    s = M.Swarm(n, arith=M.ARITH_FAST)
    s.construct(0, n, M.model_params("x500", ground_enabled=True), np.concatenate([rng.uniform(-50, 50, (n, 2)), np.full((n, 1), 10.0)], axis=1),
                np.zeros(n))
    goal = np.concatenate([rng.uniform(-50, 50, (n, 2)), np.full((n, 1), 12.0), np.zeros((n, 1))], axis=1)
    s.set_input(0, n, M.POSITION_CMD, goal)
    stop = threading.Event()
    errors = []

    def commander():
        k = 0
        try:
            while not stop.is_set():
                lo = (k * 37) % (n - 64)
                s.set_input(lo, 64, M.POSITION_CMD, goal[lo:lo + 64])
                s.get_outputs(lo, 64)
                k += 1
        except Exception as e:  # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=commander)
    t.start()
    for _ in range(300):
        s.step_n(DT, 10)
    stop.set()
    t.join()
    assert not errors
    st = s.get_state()
    assert np.all(np.isfinite(st["x"])) and np.all(np.abs(st["x"][:, 2] - 12.0) < 3.0)


def test_collisions_dense_clusters_take_the_fallback_paths(M, oracle):
    """More than 63 UAVs in one cell (saturated descriptor count), > 6 qualifying partners per UAV (hit-list overflow) and
    > 1024 candidate pairs per wavefront: the reference-order fallback must give the same forces / crash flags."""
    rng = np.random.default_rng(2024)
    n = 700
    pos = np.concatenate([rng.uniform(0.0, 0.6, (200, 3)) + [5, 5, 5],          # 200 UAVs inside one 1.75 m cell
                          rng.uniform(0.0, 3.0, (300, 3)) + [20, 20, 20],        # dense blob over a few cells
                          rng.uniform(-30, 30, (200, 3))])                       # background
    for crash in (False, True):
        p = Pair(M, n)
        p.construct(0, n, "x500", pos=pos, heading=np.zeros(n))
        p.both("handle_collisions", True, crash, 100.0)
        assert np.array_equal(p.g.has_crashed(), p.o.has_crashed())
        helpers.assert_close(p.g.get_external_force(), p.o.get_external_force(), 1e-12, "dense forces")
        if not crash:
            assert (np.abs(p.o.get_external_force()).sum(axis=1) > 0).sum() > 400


def test_full_size_100k_collision_tick_against_oracle(M, oracle):
    """BASELINE config 4 at its full size: 100 000 UAVs at 64 m^3 per UAV, elastic collisions.  Whole-swarm comparison of
    the forces with the oracle plus the size-independent property that equal-mass pair forces cancel (sum F = 0)."""
    rng = np.random.default_rng(4)
    n = 100_000
    side = (64.0 * n) ** (1.0 / 3.0)
    pos = rng.uniform(0, 1, (n, 3)) * [side * 2, side * 2, side / 4] + [0, 0, 5]
    pos[:2000] = pos[2000:4000] + rng.normal(0, 0.3, (2000, 3))  # make sure a few thousand pairs really touch
    p = Pair(M, n)
    p.construct(0, n, "x500", pos=pos, heading=np.zeros(n))
    p.both("handle_collisions", True, False, 100.0)
    fg, fo = p.g.get_external_force(), p.o.get_external_force()
    touched = (np.abs(fo).sum(axis=1) > 0).sum()
    assert touched > 2000
    helpers.assert_close(fg, fo, 1e-13, "forces at 100k")
    assert np.abs(fg.sum(axis=0)).max() < 1e-9 * np.abs(fg).sum()
    p.both("handle_collisions", False, True, 100.0)
    assert np.array_equal(p.g.has_crashed(), p.o.has_crashed()) and p.o.has_crashed().sum() == touched


def test_neighbour_lists_are_reused_and_rebuilt_with_identical_results(M, oracle):
    """Collision ticks between two neighbour searches work from the stored lists (collide.hip).  300 ticks of a swarm whose UAVs
    fly up to ~20 m/s, so the lists go stale every few dozen ticks: forces, crash-free state and PIDs must follow the oracle
    (which searches on every tick) the whole way, and the statistics must show that most ticks did NOT search."""
    rng = np.random.default_rng(77)
    n = 4000
    side = (64.0 * n) ** (1.0 / 3.0)
    p = Pair(M, n)
    pos = rng.uniform(0, side, (n, 3)) + [0, 0, 50]
    pos[:300] = pos[300:600] + rng.normal(0, 0.35, (300, 3))  # touching pairs from the first tick on
    p.construct(0, n, "x500", pos=pos, heading=rng.uniform(-3, 3, n))
    st = helpers.random_state(rng, n, 4, tilted=True)
    st["x"] = pos
    st["v"] = rng.normal(0, 6.0, (n, 3))
    p.set_state(0, n, st)
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, rng.uniform(0.4, 0.55, (n, 4)))
    touched = 0
    for chunk in range(6):
        for _ in range(50):
            p.o.step(DT)
            p.o.handle_collisions(True, False, 100.0)
        p.g.tick_n(DT, 50, True, False, 100.0)
        fo = p.o.get_external_force()
        helpers.assert_close(p.g.get_external_force(), fo, 1e-12, f"forces after {50 * (chunk + 1)} ticks")
        p.compare(RTOL_LITERAL, f"state after {50 * (chunk + 1)} ticks")
        touched = max(touched, int((np.abs(fo).sum(axis=1) > 0).sum()))
    assert touched > 100
    ticks, rebuilds = p.g.collision_stats()
    assert ticks == 300
    if os.environ.get("MRS_NEIGHBOUR_LISTS", "1") != "0":  # (the tuning switch that searches on every tick)
        assert 3 <= rebuilds <= 100, f"{rebuilds} neighbour searches in {ticks} ticks"
        if os.environ.get("MRS_FUSED_COLLISIONS", "1") != "0":
            fused, stalls, replayed, ahead = p.g.fused_stats()
            # all but the ticks that searched were evaluated inside the next step launch; the searches were queued ahead of time
            # (a UAV near the edge of its skin) or followed a stall (a UAV over the edge before the host had reacted)
            assert fused >= ticks - rebuilds - 6 and stalls + ahead >= rebuilds - 6 and ahead >= 1, (fused, stalls, replayed, ahead, rebuilds)
            print(f"{ticks} ticks: {fused} fused launches, {rebuilds} searches ({ahead} queued ahead, {stalls} after a stall), {replayed} launches replayed")


def test_neighbour_lists_follow_host_writes(M, oracle):
    """set_state / set_mass between collision ticks invalidate the stored lists (positions at the last search, airframe constants)."""
    rng = np.random.default_rng(78)
    n = 600
    p = Pair(M, n)
    pos = rng.uniform(0, 60, (n, 3)) + [0, 0, 20]
    p.construct(0, n, "x500", pos=pos, heading=np.zeros(n))
    p.both("handle_collisions", True, False, 100.0)
    p.both("handle_collisions", True, False, 100.0)  # list tick
    helpers.assert_close(p.g.get_external_force(), p.o.get_external_force(), 1e-12, "before the writes")
    # teleport UAV 5 next to UAV 400 (they were nowhere near each other when the lists were built)
    st = {k: v[5:6].copy() for k, v in p.o.get_state().items()}
    st["x"][0] = p.o.get_state()["x"][400] + [0.3, 0.1, -0.2]
    p.both("set_state", 5, 1, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    p.both("handle_collisions", True, False, 100.0)
    fo = p.o.get_external_force()
    assert np.abs(fo[5]).sum() > 0 and np.abs(fo[400]).sum() > 0
    helpers.assert_close(p.g.get_external_force(), fo, 1e-12, "after set_state")
    # the force depends on both masses (src/multirotor_simulator.cpp:350)
    p.both("set_mass", 400, 1, 5.0)
    p.both("handle_collisions", True, False, 100.0)
    helpers.assert_close(p.g.get_external_force(), p.o.get_external_force(), 1e-12, "after set_mass")
    # crash mode from the lists
    p.both("handle_collisions", False, True, 100.0)
    assert np.array_equal(p.g.has_crashed(), p.o.has_crashed()) and p.o.has_crashed()[[5, 400]].all()


def test_full_size_1M_uavs_replica_property_and_oracle_sample(M, oracle):
    """BASELINE config 5's size on one GPU: 1 000 000 UAVs, built as 244 copies of a 4096-UAV swarm (+ remainder), position
    cascade with ground contact.  Size-independent properties: every copy must end bit-identical to the first one (no
    cross-talk between lanes, blocks or the tail block), and the first copy must follow the oracle."""
    rng = np.random.default_rng(55)
    m, n = 4096, 1_000_000
    reps = -(-n // m)
    st = random_state(rng, m, 4, box=50.0, zlo=0.2, zhi=30.0, tilted=True)
    st["x"][:300, 2] = rng.uniform(0.0, 0.05, 300)  # these come down on the ground plane within the first steps
    st["v"][:300, 2] = -3.0
    cmd = np.concatenate([st["x"] + rng.uniform(-4, 4, (m, 3)), rng.uniform(-3, 3, (m, 1))], axis=1)
    cmd[:, 2] = np.abs(cmd[:, 2])
    tile = lambda a: np.concatenate([a] * reps, axis=0)[:n]
    g = M.Swarm(n, arith=M.ARITH_FAST)
    g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), tile(st["x"]), np.zeros(n))
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(g, nm)(0, n)
    g.set_state(0, n, tile(st["x"]), tile(st["v"]), tile(st["R"]), tile(st["omega"]), tile(st["motor_rpm"]))
    g.set_input(0, n, oracle.POSITION_CMD, tile(cmd))
    g.step_n(DT, 200)
    out = g.get_state()
    for k in ("x", "v", "R", "omega", "motor_rpm"):
        a = out[k]
        first = a[:m]
        for r in range(1, reps):
            blk = a[r * m:(r + 1) * m]
            assert np.array_equal(blk, first[:len(blk)]), f"{k}: copy {r} differs from copy 0"
    o = oracle.OracleSwarm(m)
    o.construct(0, m, helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0), st["x"], np.zeros(m))
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(o, nm)(0, m)
    o.set_state(0, m, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, m, oracle.POSITION_CMD, cmd)
    o.step_n(DT, 200)
    ref = o.get_state()
    for k in ("x", "v", "R", "omega", "motor_rpm"):
        helpers.assert_close(out[k][:m], ref[k], RTOL_NORTH_STAR, f"1M swarm, first copy vs oracle: {k}")
    assert (ref["x"][:, 2] <= 1e-9).any(), "some UAVs should be sitting on the ground plane"


def test_pointer_addressed_kernels_match_buffer_addressed_ones(M):
    """Swarms below 4 GiB of state use buffer-addressed columns (step_device.inc, SwarmAcc<true>); larger ones the 64-bit pointer
    form.  The choice is made once per process, so both forms run in child processes: LITERAL results must agree bit for bit,
    FAST ones to the last few bits."""
    import subprocess, sys, os, json
    code = r'''
import json, numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import mrs_multirotor_simulator_amd as M
import helpers
rng = np.random.default_rng(12)
n = 700
out = {}
for arith in (M.ARITH_LITERAL, M.ARITH_FAST):
    g = M.Swarm(n, arith=arith)
    st = helpers.random_state(rng, n, 4, box=20.0, zlo=0.1, zhi=8.0, tilted=True)
    g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), st["x"], np.zeros(n))
    g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    g.set_input(0, n, M.POSITION_CMD, np.concatenate([st["x"] + 1.0, np.zeros((n, 1))], axis=1))
    g.set_input(0, 100, M.ACTUATOR_CMD, rng.uniform(0.3, 0.6, (100, 4)))
    g.step_n(0.001, 60)
    g.step_n(0.001, 4, 5)
    s = g.get_state()
    out[str(arith)] = {k: s[k].tolist() for k in ("x", "R", "motor_rpm")}
    out[str(arith)]["pid"] = g.get_pid().tolist()
print("RESULT" + json.dumps(out))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    res = []
    for env_extra in ({}, {"MRS_NO_BUFFER_ADDRESSING": "1"}):
        env = dict(os.environ, **env_extra)
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][0][6:]))
    for arith in res[0]:
        for k in res[0][arith]:
            a, b = np.array(res[0][arith][k]), np.array(res[1][arith][k])
            if int(arith) == M.ARITH_LITERAL:
                assert np.array_equal(a, b), f"literal: {k}"
            else:  # FMA contraction is the compiler's choice per instantiation: last-bit differences are allowed here
                helpers.assert_close(a, b, RTOL_FAST, f"fast: {k}")


@pytest.mark.parametrize("n", [3_200_000, 6_500_000])
def test_state_array_beyond_2_GiB_still_addresses_every_column(M, oracle, n):
    """3.2 M UAVs = 2.2 GiB of state: the 32-bit buffer offsets of the step kernel run past 2^31.  6.5 M UAVs = 4.5 GiB: past the
    reach of a buffer resource, the launcher switches to the pointer-addressed kernels with 64-bit column offsets.  Replica property
    (all copies of a 4096-UAV swarm stay bit-identical, the last column of the last copy included) + the first copy against the oracle."""
    rng = np.random.default_rng(56)
    m = 4096
    reps = -(-n // m)
    st = random_state(rng, m, 4)
    cmd = rng.uniform(0.35, 0.6, (m, 4))
    tile = lambda a: np.concatenate([a] * reps, axis=0)[:n]
    g = M.Swarm(n, arith=M.ARITH_FAST)
    g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), tile(st["x"]), np.zeros(n))
    g.set_state(0, n, tile(st["x"]), tile(st["v"]), tile(st["R"]), tile(st["omega"]), tile(st["motor_rpm"]))
    g.set_input(0, n, oracle.ACTUATOR_CMD, tile(cmd))
    g.step_n(DT, 50)
    out = g.get_state()
    imu = g.get_imu()
    del g
    for k, a in list(out.items()) + [("imu", imu)]:
        first = a[:m]
        for r in range(1, reps):
            blk = a[r * m:(r + 1) * m]
            assert np.array_equal(blk, first[:len(blk)]), f"{k}: copy {r} differs from copy 0"
    o = oracle.OracleSwarm(m)
    o.construct(0, m, helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0), st["x"], np.zeros(m))
    o.set_state(0, m, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, m, oracle.ACTUATOR_CMD, cmd)
    o.step_n(DT, 50)
    ref = o.get_state()
    for k in ("x", "v", "R", "omega", "motor_rpm"):
        helpers.assert_close(out[k][:m], ref[k], RTOL_NORTH_STAR, f"{n} UAVs, first copy vs oracle: {k}")


def test_fast_kernel_nan_lane_does_not_disturb_its_wave(M, oracle):
    """FAST runs the RK4 stages without the per-component NaN guards and repeats a wave's step with them when a lane ends up with a
    NaN (step_device.inc).  The finite lanes of such a wave must get exactly the bits they get without the NaN neighbour, the NaN
    lane must be rolled back like the reference does (multirotor_model.hpp:228-233), and the counters must show it."""
    rng = np.random.default_rng(88)
    n = 256
    st = random_state(rng, n, 4)
    cmd = rng.uniform(0.35, 0.6, (n, 4))
    sws = []
    for poison in (False, True):
        g = M.Swarm(n, arith=M.ARITH_FAST)
        g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), st["x"], np.zeros(n))
        s2 = {k: v.copy() for k, v in st.items()}
        if poison:
            s2["v"][5, 1] = np.nan          # wave 0
            s2["omega"][130, 2] = np.inf    # wave 2: an inf rate makes NaNs inside the stages
        g.set_state(0, n, s2["x"], s2["v"], s2["R"], s2["omega"], s2["motor_rpm"])
        g.set_input(0, n, oracle.ACTUATOR_CMD, cmd)
        g.step_n(DT, 3)
        sws.append((g.get_state(), g.get_diag()))
    (a, da), (b, db) = sws
    clean = np.ones(n, bool)
    clean[[5, 130]] = False
    for k in ("x", "v", "R", "omega", "motor_rpm"):
        assert np.array_equal(a[k][clean], b[k][clean]), k
    assert da["nan_rollback"] == 0 and db["nan_rollback"] >= 3
    # the rolled-back lane keeps its (non-finite) start state for x/v/R/omega; the motor filter still runs (:244-246)
    assert np.isnan(b["v"][5, 1]) and np.array_equal(b["x"][5], st["x"][5])
    assert np.array_equal(b["motor_rpm"][5], a["motor_rpm"][5])


def test_nan_rollback_inside_a_fused_launch(M, oracle):
    """A UAV whose step produces a NaN is rolled back to the start of THAT step (multirotor_model.hpp:228-233).  Inside a launch that
    fuses several sub-steps the state of the launch start is all HBM holds, so the kernel keeps the sub-step's start state itself:
    a UAV that turns non-finite in the middle of a launch must end exactly where the unfused sequence leaves it."""
    rng = np.random.default_rng(61)
    n = 192
    wild = [7, 70, 130]
    for arith, tol in ((M.ARITH_LITERAL, RTOL_LITERAL), (M.ARITH_FAST, RTOL_FAST)):
        st = random_state(rng, n, 4)
        # body rates of 1e150 rad/s: R grows by ~1e147 per step, R^T R overflows a few steps in -> NaN derivatives -> rollbacks
        st["omega"][7] = [1e150, 0.0, 0.0]
        st["omega"][70] = [0.0, 3e151, 1e150]
        st["omega"][130] = [1e100, 1e100, 1e100]
        cmd = rng.uniform(0.35, 0.6, (n, 4))
        runs, diags = [], []
        for sub in (1, 6, 4):
            p = Pair(M, n, arith=arith)
            p.construct(0, n, "x500", ground_enabled=True, ground_z=0.0)
            p.set_state(0, n, st)
            p.both("set_input", 0, n, oracle.ACTUATOR_CMD, cmd)
            p.o.step_n(DT, 12)
            p.g.step_n(DT, 12, sub)
            a, b = p.g.get_state(), p.o.get_state()
            clean = np.ones(n, bool)
            clean[wild] = False
            for k in a:  # the oracle comparison leaves the overflowing UAVs out: inf arithmetic is not part of the parity contract
                helpers.assert_close(a[k][clean], b[k][clean], tol, f"arith {arith}, {sub} sub-steps: {k}")
            runs.append(a)
            diags.append(p.g.get_diag()["nan_rollback"])
        assert diags[0] > 0 and diags[0] == diags[1] == diags[2], diags
        for r in runs[1:]:
            for k in runs[0]:
                if arith == M.ARITH_LITERAL:
                    assert np.array_equal(runs[0][k], r[k], equal_nan=True), f"literal: fused != unfused in {k}"
                else:  # FMA contraction differs between the kernel instantiations: last bits only, and only finite lanes compared
                    helpers.assert_close(runs[0][k][clean], r[k][clean], 1e-11, f"fast: fused vs unfused, {k}")


@pytest.mark.parametrize("fast", [False, True])
def test_two_stream_step_runs_equal_single_stream_ones(M, oracle, fast, monkeypatch):
    """A run of steps is issued as two half-swarm launches per step on two streams (tick_single.hip); same kernels, same blocks:
    the results must be bit-identical to the single-stream order, also when other calls are interleaved between the runs."""
    rng = np.random.default_rng(97)
    n = 70_001  # 1094 blocks: above the split threshold, odd block count, ragged tail
    st = random_state(rng, n, 4, tilted=True)
    goals = np.concatenate([st["x"] + rng.uniform(-3, 3, (n, 3)), rng.uniform(-3, 3, (n, 1))], axis=1)
    act = rng.uniform(0.35, 0.6, (n // 2, 4))
    results = []
    for split in ("0", "1"):
        monkeypatch.setenv("MRS_SPLIT_STREAMS", split)
        g = M.Swarm(n, arith=M.ARITH_FAST if fast else M.ARITH_LITERAL)
        g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), st["x"], np.zeros(n))
        g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        g.set_input(0, n, oracle.POSITION_CMD, goals)
        g.step_n(DT, 12)
        g.set_input(0, n // 2, oracle.ACTUATOR_CMD, act)     # ordered after the run on both streams
        g.apply_force(n - 100, 100, rng.normal(0, 2, (100, 3)) * 0 + 1.5)
        g.step_n(DT, 9)
        g.step_n(DT, 8, 2)                                   # fused sub-steps, four launches
        g.handle_collisions(True, False, 100.0)
        g.step_n(DT, 5)
        s = g.get_state()
        s["pid"], s["imu"], s["force"] = g.get_pid(), g.get_imu(), g.get_external_force()
        results.append(s)
        del g
    for k in results[0]:
        assert np.array_equal(results[0][k], results[1][k]), k


def test_step_runs_right_after_a_synchronize_skip_the_fork_event_with_identical_results(M, oracle):
    """A split run that is the first call after mrs_swarm_synchronize starts its second stream without a fork event (both streams are
    idle); any other call in between brings the event back.  Same results either way, and the profile region (start event, one end
    event per stream) reports a plausible time."""
    rng = np.random.default_rng(98)
    n = 70_001
    st = random_state(rng, n, 4, tilted=True)
    goals = np.concatenate([st["x"] + rng.uniform(-3, 3, (n, 3)), rng.uniform(-3, 3, (n, 1))], axis=1)
    results = []
    for quiet in (False, True):
        g = M.Swarm(n, arith=M.ARITH_FAST)
        g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), st["x"], np.zeros(n))
        g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        g.set_input(0, n, oracle.POSITION_CMD, goals)
        g.set_profiling(1)
        times = []
        for _ in range(4):
            g.synchronize()
            if not quiet:
                g.has_crashed(0, 1)  # any call between the synchronize and the run: the run forks with an event again
            g.step_n(DT, 6)
            times.append(g.last_step_kernel_ms())
            g.set_input(n - 64, 64, oracle.POSITION_CMD, goals[n - 64:])  # host write right behind a split run (ordered after both streams)
        g.set_profiling(0)
        s = g.get_state()
        s["pid"], s["imu"] = g.get_pid(), g.get_imu()
        results.append(s)
        for ms, launches in times:
            assert launches == 6 and 2e-3 < ms < 1.0, (ms, launches)  # per-step device time of a 70 k cascade step: microseconds, not zero
        del g
    for k in results[0]:
        assert np.array_equal(results[0][k], results[1][k]), k


def test_gathered_collisions_reuse_neighbour_lists_over_many_ticks(M, oracle):
    """The multi-GPU collision path on two virtual shards of one GPU, 200 ticks of a moving swarm: per tick every shard steps,
    packs its records into the (NaN-padded) gathered buffer and collides against it.  The shards keep neighbour lists over the
    gathered records between searches; forces, crash flags and states must follow the single-swarm oracle, and most ticks must not
    search."""
    import torch
    from mrs_multirotor_simulator_amd.sharded import max_shard, shard_range
    rng = np.random.default_rng(123)
    n_total, world = 3001, 2
    side = (64.0 * n_total) ** (1.0 / 3.0)
    pos = rng.uniform(0, side, (n_total, 3)) + [0, 0, 30]
    pos[:200] = pos[200:400] + rng.normal(0, 0.3, (200, 3))
    st = helpers.random_state(rng, n_total, 4, tilted=True)
    st["x"] = pos
    st["v"] = rng.normal(0, 5.0, (n_total, 3))
    cmd = rng.uniform(0.4, 0.55, (n_total, 4))
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    n_max = max_shard(n_total, world)
    recv = torch.full((world * n_max, 6), float("nan"), dtype=torch.float64, device="cuda")
    shards = []
    for r in range(world):
        lo, hi = shard_range(n_total, world, r)
        g = M.Swarm(hi - lo)
        g.construct(0, hi - lo, helpers.to_product_params(M, po), pos[lo:hi], np.zeros(hi - lo))
        g.set_state(0, hi - lo, st["x"][lo:hi], st["v"][lo:hi], st["R"][lo:hi], st["omega"][lo:hi], st["motor_rpm"][lo:hi])
        g.set_input(0, hi - lo, oracle.ACTUATOR_CMD, cmd[lo:hi])
        shards.append((g, lo, hi))
    for tick in range(200):
        crash = tick == 150
        o.step(DT)
        o.handle_collisions(True, crash, 100.0)
        for r, (g, lo, hi) in enumerate(shards):
            g.step(DT)
            g.pack_positions_to(recv[r * n_max:].data_ptr())
        for g, _, _ in shards:
            g.synchronize()
        for r, (g, lo, hi) in enumerate(shards):
            g.handle_collisions_gathered(recv.data_ptr(), world * n_max, r * n_max, True, crash, 100.0)
        if tick % 40 == 39 or tick == 150:
            so = o.get_state()
            for r, (g, lo, hi) in enumerate(shards):
                helpers.assert_close(g.get_external_force(), o.get_external_force(lo, hi - lo), 1e-11, f"tick {tick} shard {r} forces")
                assert np.array_equal(g.has_crashed(), o.has_crashed(lo, hi - lo)), f"tick {tick} shard {r} crash flags"
                sg = g.get_state()
                for k in ("x", "v", "R", "omega"):
                    helpers.assert_close(sg[k], so[k][lo:hi], RTOL_LITERAL, f"tick {tick} shard {r} {k}")
    assert o.has_crashed().sum() > 0 and (np.abs(o.get_external_force()).sum(axis=1) > 0).sum() > 50
    for g, _, _ in shards:
        ticks, searches = g.collision_stats()
        assert ticks == 200
        if os.environ.get("MRS_NEIGHBOUR_LISTS", "1") != "0":
            assert 2 <= searches <= 80, f"{searches} searches in {ticks} gathered ticks"


def test_library_driven_sharded_tick_with_a_one_rank_communicator(M, oracle):
    """mrs_swarm_tick_sharded_n: the library issues the all-gather itself (RCCL bound at run time, on the swarm's stream).  A
    one-rank communicator exercises the whole path on the single-GPU box — unique id, communicator, NaN-padded send buffer,
    all-gather, gathered collision pass with neighbour lists — and must reproduce mrs_swarm_tick_n on a twin swarm and the oracle.
    A shard that is not the one of its rank is refused."""
    from mrs_multirotor_simulator_amd import swarm as S
    rng = np.random.default_rng(77)
    n = 2500
    side = (64.0 * n) ** (1.0 / 3.0)
    pos = rng.uniform(0, side, (n, 3)) + [0, 0, 30]
    pos[:150] = pos[150:300] + rng.normal(0, 0.3, (150, 3))
    st = helpers.random_state(rng, n, 4, tilted=True)
    st["x"] = pos
    st["v"] = rng.normal(0, 5.0, (n, 3))
    cmd = rng.uniform(0.4, 0.55, (n, 4))
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n)
    o.construct(0, n, po, pos, np.zeros(n))
    o.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n, oracle.ACTUATOR_CMD, cmd)
    twins = []
    for _ in range(2):
        g = M.Swarm(n)
        g.construct(0, n, helpers.to_product_params(M, po), pos, np.zeros(n))
        g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        g.set_input(0, n, oracle.ACTUATOR_CMD, cmd)
        twins.append(g)
    a, b = twins
    uid = S.rccl_unique_id()
    assert len(uid) == 128
    with pytest.raises(M.MrsError):
        a.tick_sharded_n(DT, 1, True, False, 100.0)  # no communicator yet
    with pytest.raises(M.MrsError):
        a.comm_init(1, 0, uid, n + 5)  # n_total does not match the shard of rank 0
    a.comm_init(1, 0, uid, n)
    for block, crash in ((60, False), (1, True), (59, False)):
        a.tick_sharded_n(DT, block, True, crash, 100.0)
        b.tick_n(DT, block, True, crash, 100.0)
        for _ in range(block):
            o.step(DT)
            o.handle_collisions(True, crash, 100.0)
        sa, sb, so = a.get_state(), b.get_state(), o.get_state()
        for k in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(sa[k], sb[k], RTOL_LITERAL, "twin " + k)
            helpers.assert_close(sa[k], so[k], RTOL_LITERAL, k)
        helpers.assert_close(a.get_external_force(), b.get_external_force(), 1e-11, "twin forces")
        helpers.assert_close(a.get_external_force(), o.get_external_force(), 1e-11, "forces")
        assert np.array_equal(a.has_crashed(), o.has_crashed())
    assert o.has_crashed().sum() > 0
    assert a.collision_stats()[0] == 120
    a.comm_destroy()
    a.comm_destroy()  # idempotent
    with pytest.raises(M.MrsError):
        a.tick_sharded_n(DT, 1, True, False, 100.0)


def test_device_pid_reproduces_the_reference_pid_vectors(M):
    """The cascade kernels' PID device function against the golden vectors recorded from the REFERENCE's own PIDController
    (tests/golden/pid_reference_vectors.npz, made by make_golden.py:pid_reference from controllers/pid.hpp compiled where it lies).
    LITERAL: bit for bit, NaN pattern included.  FAST multiplies by 1/dt instead of dividing and may contract a*b+c: within 1e-12
    of the output scale per update while the loop state stays finite."""
    from mrs_multirotor_simulator_amd import swarm as S
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pid_reference_vectors.npz"))
    lit = S.debug_pid_sequences(M.ARITH_LITERAL, g["params"], g["err"], g["dt"], g["event"], g["new_sat"])
    exp = g["out"]
    same = (lit.view(np.uint64) == exp.view(np.uint64)) | (np.isnan(lit) & np.isnan(exp))
    assert same.all(), f"LITERAL: {int((~same).sum())} of {same.size} updates differ, first at {np.argwhere(~same)[0]}"
    fast = S.debug_pid_sequences(M.ARITH_FAST, g["params"], g["err"], g["dt"], g["event"], g["new_sat"])
    assert np.array_equal(np.isnan(fast), np.isnan(exp))
    inf = np.isinf(exp)
    assert np.array_equal(fast[inf], exp[inf])  # an infinite error without saturation: the same infinity
    ok = np.isfinite(exp)
    scale = np.maximum(1.0, np.max(np.where(ok, np.abs(exp), 0.0), axis=1, keepdims=True))
    assert np.max(np.abs(fast[ok] - exp[ok]) / np.broadcast_to(scale, exp.shape)[ok]) < 1e-12


def test_non_temporal_kernel_matches_the_ordinary_one(M):
    """Model-only steps of small swarms run the `*_buf_nt` kernel (non-temporal state accesses, step_device.inc); the launcher's
    choice can be forced either way per process (MRS_NT_ACCESSES), so both run in child processes on the same actuator-level
    swarm, single- and two-stream runs included.  A cache hint must not change a bit in LITERAL; in FAST the two instantiations
    may contract differently (last bits)."""
    import subprocess, sys, json
    code = r'''
import json, numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import mrs_multirotor_simulator_amd as M
import helpers
rng = np.random.default_rng(31)
n = 70000  # 1094 blocks: enough for the two-stream form of a long run
out = {}
for arith in (M.ARITH_LITERAL, M.ARITH_FAST):
    g = M.Swarm(n, arith=arith)
    st = helpers.random_state(rng, n, 4, tilted=True)
    g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), st["x"], np.zeros(n))
    g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    g.set_input(0, n, M.ACTUATOR_CMD, rng.uniform(0.3, 0.6, (n, 4)))
    g.step_n(0.001, 3)    # single stream
    g.step_n(0.001, 40)   # two streams
    s = g.get_state(0, 4096)
    t = g.get_state(n - 4096, 4096)
    out[str(arith)] = {k: np.concatenate([s[k], t[k]]).tolist() for k in ("x", "v", "R", "omega", "motor_rpm")}
print("RESULT" + json.dumps(out))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    res = []
    for nt in ("0", "1"):
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, MRS_NT_ACCESSES=nt), timeout=300)
        assert p.returncode == 0, p.stderr[-2000:]
        res.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][0][6:]))
    for arith in res[0]:
        for k in res[0][arith]:
            a, b = np.array(res[0][arith][k]), np.array(res[1][arith][k])
            if int(arith) == M.ARITH_LITERAL:
                assert np.array_equal(a, b), f"literal: {k}"
            else:
                helpers.assert_close(a, b, RTOL_FAST, f"fast: {k}")
