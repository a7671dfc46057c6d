"""What the pool of stand-alone UavSystem objects (include/.../uav_system.hpp UavPool) is made of, each on its own against the oracle:
mrs_swarm_step_range (uavs_[i]->makeStep(dt) for a sub-range only, src/multirotor_simulator.cpp:212), mrs_swarm_copy_uavs
(UavSystem copy assignment between batches, src/uav_system_ros.cpp:105), mrs_swarm_clone_resized (a pool that grows) and
mrs_swarm_get_states (MultirotorModel::State + IMU + crash flag of a range in one download, multirotor_model.hpp:90-98)."""
import numpy as np
import pytest

import helpers
from helpers import Pair, RTOL_LITERAL

pytestmark = pytest.mark.gpu
DT = 0.001


def mixed_pair(M, O, n, rng, arith=0):
    """x500 and f550 UAVs in alternating runs (blocks of mixed airframes), position and actuator commands"""
    p = Pair(M, n, arith=arith)
    cut = [0, 70, 130, 200, n]
    for k in range(4):
        a, b = cut[k], cut[k + 1]
        af = "x500" if k % 2 == 0 else "f550"
        pos = rng.uniform(-20, 20, (b - a, 3)) + [0, 0, 30]
        p.construct(a, b - a, af, pos=pos, heading=rng.uniform(-3, 3, b - a), ground_enabled=True, ground_z=0.0)
        nm = 4 if af == "x500" else 6
        st = helpers.random_state(rng, b - a, nm, tilted=True)
        st["x"] = pos
        p.set_state(a, b - a, st)
        if k < 2:
            p.both("set_input", a, b - a, O.POSITION_CMD, np.concatenate([pos + rng.uniform(-3, 3, (b - a, 3)), rng.uniform(-3, 3, (b - a, 1))], axis=1))
        else:
            p.both("set_input", a, b - a, O.ACTUATOR_CMD, rng.uniform(0.3, 0.6, (b - a, nm)))
    return p


@pytest.mark.parametrize("arith", [0, 1])
@pytest.mark.parametrize("first,count", [(0, 1), (63, 2), (65, 140), (199, 57), (255, 1), (100, 156)])
def test_step_range_steps_exactly_the_range(mrs, oracle, first, count, arith):
    M, O = mrs, oracle
    n = 256
    p = mixed_pair(M, O, n, np.random.default_rng(first * 1000 + count), arith)
    p.step(DT, 3)
    for _ in range(4):  # the oracle: everybody else on hold
        p.g.step_range(first, count, DT)
        if first > 0:
            p.o.set_hold(0, first, True)
        if first + count < n:
            p.o.set_hold(first + count, n - first - count, True)
        p.o.step(DT)
        p.o.set_hold(0, n, False)
    p.compare(RTOL_LITERAL if arith == 0 else helpers.RTOL_FAST, f"range [{first}, {first + count})")
    p.step(DT, 2)  # whole-swarm steps go on from there
    p.compare(RTOL_LITERAL if arith == 0 else helpers.RTOL_FAST, "whole steps after the partial ones")


def test_step_range_with_collisions_pending(mrs, oracle):
    """a collision tick requested before the partial step acts on it (applyForce happens inside handleCollisions, :356-358); the next
    collision tick searches again (the partial step ran no skin test)"""
    M, O = mrs, oracle
    rng = np.random.default_rng(5)
    n = 192
    p = Pair(M, n)
    pos = rng.uniform(0, 6, (n, 3)) + [0, 0, 20]  # dense: many pairs in contact
    p.construct(0, n, "x500", pos=pos, heading=np.zeros(n), ground_enabled=True, ground_z=0.0)
    p.both("set_input", 0, n, O.ACTUATOR_CMD, rng.uniform(0.4, 0.5, (n, 4)))
    for _ in range(3):
        p.step(DT, 1)
        p.both("handle_collisions", True, False, 100.0)
    p.g.step_range(40, 100, DT)
    p.o.set_hold(0, 40, True)
    p.o.set_hold(140, n - 140, True)
    p.o.step(DT)
    p.o.set_hold(0, n, False)
    p.both("handle_collisions", True, False, 100.0)
    p.step(DT, 1)
    p.compare(RTOL_LITERAL, "partial step between collision ticks")
    helpers.assert_close(p.g.get_external_force(), p.o.get_external_force(), 1e-12, "forces")
    assert (np.abs(p.o.get_external_force()).sum(axis=1) > 0).sum() > 10


def test_copy_uavs_and_resized_clone(mrs, oracle):
    M, O = mrs, oracle
    rng = np.random.default_rng(11)
    n = 256
    p = mixed_pair(M, O, n, rng)
    p.step(DT, 5)
    big = p.g.clone_resized(512)  # the pool grows: the first 256 are copies, the rest UavSystem()
    a, b = p.g.get_state(), big.get_state(0, n)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(big.get_pid(0, n), p.g.get_pid())
    fresh = big.get_state(n, 256)
    assert np.all(fresh["x"] == 0) and np.all(fresh["R"] == np.eye(3)) and np.all(fresh["motor_rpm"] == 0)
    # the copy lives on: same steps, same results as the original (and the oracle)
    big.step_n(DT, 4)
    p.step(DT, 4)
    b = big.get_state(0, n)
    for k, v in p.g.get_state().items():
        assert np.array_equal(v, b[k]), k
    p.compare(RTOL_LITERAL, "original after the clone")
    # copy assignment between the two batches: f550 UAVs 80..99 of the original over x500 UAVs 10..29 of the copy (airframe, state,
    # command, PIDs travel), and inside one swarm
    big.copy_uavs(10, p.g, 80, 20)
    big.copy_uavs(300, big, 130, 40)
    big.step_n(DT, 3)
    p.step(DT, 3)
    so = p.o.get_state()
    got = big.get_state()
    for k in ("x", "v", "R", "omega", "motor_rpm"):
        helpers.assert_close(got[k][10:30], so[k][80:100], RTOL_LITERAL, f"copied from the other swarm: {k}")
        helpers.assert_close(got[k][300:340], so[k][130:170], RTOL_LITERAL, f"copied inside the swarm: {k}")
        helpers.assert_close(got[k][30:80], so[k][30:80], RTOL_LITERAL, f"untouched neighbours: {k}")
    assert big.get_params(12).n_motors == 6 and big.get_params(9).n_motors == 4
    with pytest.raises(M.MrsError):
        big.copy_uavs(0, big, 10, 20)  # overlapping ranges of one swarm
    other = M.Swarm(64)  # not a clone: index 1 of its parameter table is an a300, index 1 of the pair's tables an x500
    other.construct(0, 64, M.model_params("a300"))
    with pytest.raises(M.MrsError):
        big.copy_uavs(0, other, 0, 10)


def test_get_states_packs_state_imu_and_crash_flag(mrs, oracle):
    M, O = mrs, oracle
    rng = np.random.default_rng(13)
    n = 256
    p = mixed_pair(M, O, n, rng)
    p.step(DT, 7)
    p.both("crash", 33, 2)
    st = p.g.get_states(20, 200)
    ref = p.g.get_state(20, 200)
    for k in ("x", "v", "v_prev", "omega", "motor_rpm"):
        assert np.array_equal(st[k], ref[k]), k
    assert np.array_equal(st["R"], ref["R"].reshape(-1, 3, 3))
    assert np.array_equal(st["imu_acceleration"], p.g.get_imu(20, 200))
    assert np.array_equal(st["crashed"], p.g.has_crashed(20, 200))
    assert st["crashed"].sum() == 2 and set(st["n_motors"]) == {4, 6}
    # v_prev of a UAV whose velocity the host changed (MultirotorModel::setState leaves v_prev alone, multirotor_model.hpp:424-433)
    old_v = p.g.get_state(50, 1)["v"].copy()
    p.g.set_state(50, 1, v=[[1.0, 2.0, 3.0]])
    one = p.g.get_states(50, 1)
    assert np.array_equal(one["v"], [[1.0, 2.0, 3.0]]) and np.array_equal(one["v_prev"], old_v)
    helpers.assert_close(st["x"], p.o.get_state(20, 200)["x"], RTOL_LITERAL, "x vs oracle")
