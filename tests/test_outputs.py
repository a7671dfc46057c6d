"""Publisher payloads (odometry / IMU / rangefinder / poses) — SURVEY §8f rank 2.  CPU: the oracle's restatement of
Eigen::Quaterniond(R) (what mrs_lib::AttitudeConverter stores) and of publishRangefinder against independent maths;
GPU: the packed device derivation against the oracle."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

import helpers


def special_rotations():
    rs = [np.eye(3), np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0]), np.diag([-1.0, -1.0, 1.0])]
    for ax, ang in (([1, 0, 0], 3.0), ([0, 1, 0], 3.1), ([0, 0, 1], -3.05), ([1, 1, 0], 2.9), ([0.2, 1, 1], 3.13), ([1, 0.1, 1], np.pi)):
        rs.append(Rotation.from_rotvec(np.array(ax, float) / np.linalg.norm(ax) * ang).as_matrix())
    return np.array(rs)


def scenario(O, rng, n):
    R = np.concatenate([special_rotations(), helpers.random_rotations(rng, n - len(special_rotations()))])
    st = helpers.random_state(rng, n, 4)
    st["R"] = R
    st["x"][:, 2] = rng.uniform(0.0, 60.0, n)  # some beyond the 40 m range limit once tilted
    return st


def test_oracle_outputs_known_answers(oracle):
    O = oracle
    rng = np.random.default_rng(5)
    n = 300
    st = scenario(O, rng, n)
    s = O.OracleSwarm(n)
    p = helpers.oracle_params("x500", ground_enabled=True, ground_z=1.5)
    s.construct(0, n, p)
    s.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    s.set_input(0, n, O.ACTUATOR_CMD, np.full((n, 4), 0.5))
    s.step(0.001)  # gives a non-trivial IMU value
    st = s.get_state()
    o = s.get_outputs()
    assert np.array_equal(o["position"], st["x"]) and np.array_equal(o["angular_velocity"], st["omega"])
    assert np.array_equal(o["linear_acceleration"], s.get_imu())
    assert np.allclose(o["velocity_body"], np.einsum("nji,nj->ni", st["R"], st["v"]), rtol=1e-14, atol=1e-14)
    q_ref = Rotation.from_matrix(st["R"]).as_quat()  # x y z w
    sign = np.sign(np.sum(q_ref * o["orientation"], axis=1, keepdims=True))
    assert np.allclose(o["orientation"] * sign, q_ref, atol=1e-9)
    assert np.allclose(np.linalg.norm(o["orientation"], axis=1), 1.0, atol=1e-9)
    # Eigen's branch: w >= 0 when trace > 0
    tr = np.trace(st["R"], axis1=1, axis2=2)
    assert np.all(o["orientation"][tr > 0, 3] > 0)
    bz = st["R"][:, 2, 2]
    exp = np.where(bz > 0, (st["x"][:, 2] - 1.5) / np.where(bz > 0, bz, 1.0) + 0.01, np.inf)
    exp = np.where(exp > 40.0, 41.0, exp)
    assert np.allclose(o["range"], exp, rtol=1e-12)
    assert (o["range"] == 41.0).sum() > 10 and (o["range"] < 40).sum() > 10


def test_identity_and_half_turns(oracle):
    s = oracle.OracleSwarm(4)
    R = np.array([np.eye(3), np.diag([1.0, -1, -1]), np.diag([-1.0, 1, -1]), np.diag([-1.0, -1, 1])])
    s.set_state(0, 4, np.zeros((4, 3)), np.zeros((4, 3)), R, np.zeros((4, 3)), np.zeros((4, 8)))
    q = s.get_outputs()["orientation"]
    assert np.array_equal(q, [[0, 0, 0, 1], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]])


@pytest.mark.gpu
def test_gpu_outputs_match_oracle(mrs, oracle):
    rng = np.random.default_rng(6)
    n = 3000
    p = helpers.Pair(mrs, n)
    p.construct(0, 1500, "x500", ground_enabled=True, ground_z=1.5)
    p.construct(1500, 1500, "f550", ground_enabled=True, ground_z=-2.0)
    p.set_state(0, n, scenario(oracle, rng, n))
    p.both("set_input", 0, 1500, oracle.ACTUATOR_CMD, np.full((1500, 4), 0.5))
    p.both("set_input", 1500, 1500, oracle.ACTUATOR_CMD, np.full((1500, 6), 0.5))
    p.step(0.001, 2)
    a, b = p.g.get_outputs(), p.o.get_outputs()
    for k in ("position", "orientation", "velocity_body", "angular_velocity", "linear_acceleration", "range"):
        helpers.assert_close(a[k], b[k], 1e-12, k)
    sub = p.g.get_outputs(1490, 20)
    assert np.array_equal(sub["position"], a["position"][1490:1510]) and np.array_equal(sub["range"], a["range"][1490:1510])


@pytest.mark.gpu
def test_outputs_view_and_staged_input_equal_the_copying_calls(mrs, oracle):
    """mrs_swarm_get_outputs_view == mrs_swarm_get_outputs; mrs_swarm_input_staging + commit_input == mrs_swarm_set_input, for every
    payload width, on a sub-range, and re-used across ticks."""
    import helpers
    rng = np.random.default_rng(91)
    n = 333
    a, b = mrs.Swarm(n), mrs.Swarm(n)
    pos = rng.uniform(-20, 20, (n, 3)) + [0, 0, 30]
    for g in (a, b):
        g.construct(0, n, mrs.model_params("x500", ground_enabled=True, ground_z=0.0), pos, rng.uniform(-3, 3, n) * 0)
    cases = [(mrs.POSITION_CMD, 4, 4), (mrs.ACTUATOR_CMD, 4, 4), (mrs.ACTUATOR_CMD, 8, 8), (mrs.ATTITUDE_CMD, 10, 10),
             (mrs.TILT_HDG_RATE_CMD, 5, 7), (mrs.VELOCITY_HDG_CMD, 4, 6)]
    first, count = 17, 300
    for tick, (mode, width, stride) in enumerate(cases):
        payload = rng.uniform(0.3, 0.6, (count, stride))
        if mode == mrs.ATTITUDE_CMD:
            payload[:, :9] = helpers.tilted_rotations(rng, count).reshape(count, 9)
        a.set_input(first, count, mode, payload)
        rows = b.input_staging(count, stride)
        rows[:] = payload
        b.commit_input(first, count, mode, stride)
        a.step_n(0.001, 5)
        b.step_n(0.001, 5)
        sa, sb = a.get_state(), b.get_state()
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), f"case {tick}: {k}"
        oa = a.get_outputs(first, count)
        ov = b.get_outputs_view(first, count)
        assert oa.dtype == ov.dtype and oa.shape == ov.shape
        for f in oa.dtype.names:
            assert np.array_equal(oa[f], ov[f]), f"case {tick}: output {f}"
    with pytest.raises(Exception):
        b.commit_input(0, n, mrs.ATTITUDE_CMD, 4)   # stride too small for the mode


@pytest.mark.gpu
def test_pipelined_outputs_equal_the_synchronous_ones_while_steps_run(mrs, oracle):
    """mrs_swarm_get_outputs_async / mrs_swarm_outputs_wait: the download of tick t is waited for AFTER tick t + 1 has
    been queued — staged commands every tick through the two row blocks, two tickets in flight — and must be the payload a
    synchronous mrs_swarm_get_outputs returns right after tick t on a twin swarm; the twin's final state equals the pipelined one."""
    rng = np.random.default_rng(17)
    n = 20_000
    a, b = mrs.Swarm(n, arith=mrs.ARITH_FAST), mrs.Swarm(n, arith=mrs.ARITH_FAST)
    st = scenario(oracle, rng, n)
    for g in (a, b):
        g.construct(0, n, mrs.model_params("x500", ground_enabled=True, ground_z=0.0))
        g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    ticks = 12
    cmds = rng.uniform(0.35, 0.6, (ticks, n, 4))
    want = []
    for t in range(ticks):  # the serial loop: upload, step, download, one after the other
        a.set_input(0, n, mrs.ACTUATOR_CMD, cmds[t])
        a.step(0.001)
        want.append(a.get_outputs().copy())
    pending = None
    for t in range(ticks):  # the pipelined loop
        rows = b.input_staging(n, 4)
        rows[:] = cmds[t]
        b.commit_input(0, n, mrs.ACTUATOR_CMD, 4)
        b.step(0.001)
        ticket = b.get_outputs_async()
        if pending is not None:  # tick t is in flight while tick t - 1 is read
            got = b.outputs_wait(pending[1])
            for f in got.dtype.names:
                assert np.array_equal(got[f], want[pending[0]][f]), f"tick {pending[0]}: {f}"
        pending = (t, ticket)
    got = b.outputs_wait(pending[1])
    for f in got.dtype.names:
        assert np.array_equal(got[f], want[ticks - 1][f]), f"last tick: {f}"
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    sub = b.outputs_wait(b.get_outputs_async(100, 50))  # a sub-range, waited for at once
    assert np.array_equal(sub["position"], want[-1]["position"][100:150])
    with pytest.raises(Exception):
        b.outputs_wait(pending[1] - 1)  # that block has long been handed to a newer download


@pytest.mark.gpu
def test_pipelined_outputs_survive_a_stall_of_the_lazy_collision_ticks(mrs, oracle):
    """With collisions on, mrs_swarm_tick_n leaves every tick to the NEXT launch and never synchronises; a launch whose lists went
    stale turns the launches behind it into no-ops — and a pack queued behind a no-op has packed an older state.  The wait must
    notice, replay and pack again: every pipelined payload equals the synchronous one of a twin swarm, stalls included."""
    import bench
    n, per_call = 20_000, 3
    st, cmd = bench.make_inputs(n, "position+collisions", seed=11, volume_per_uav=16.0)
    # a few UAVs that cross the last quarter of their skin (0.0625 m) within ONE step while their speed lasts (air drag): no warning
    # can come in time, the launch finds its lists stale and the launches queued behind it in the same call are no-ops
    st["v"][:8] = [0.0, 170.0, 0.0]

    def make():
        g = mrs.Swarm(n, arith=mrs.ARITH_FAST)
        g.construct(0, n, mrs.model_params("x500", ground_enabled=True, ground_z=0.0))
        g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        g.set_input(0, n, mrs.POSITION_CMD, cmd)
        return g

    a, calls, want = make(), 25, []
    for t in range(calls):  # the twin: every call settled by a synchronous download
        a.tick_n(0.001, per_call, True, False, 100.0)
        want.append(a.get_outputs().copy())
    b, pending, worst = make(), None, 0.0
    for t in range(calls):  # nothing but ticks and pipelined downloads: the host never looks at the swarm
        b.tick_n(0.001, per_call, True, False, 100.0)
        ticket = b.get_outputs_async()
        if pending is not None:
            got = b.outputs_wait(pending[1])
            for f in got.dtype.names:
                worst = max(worst, helpers.rel_linf(got[f], want[pending[0]][f]))
                helpers.assert_close(got[f], want[pending[0]][f], 1e-9, f"call {pending[0]}: {f}")
        pending = (t, ticket)
    got = b.outputs_wait(pending[1])
    for f in got.dtype.names:
        helpers.assert_close(got[f], want[pending[0]][f], 1e-9, f"last call: {f}")
    fused, stalls, replayed, ahead = b.fused_stats()
    print(f"pipelined outputs with lazy collision ticks: {fused} fused launches, {stalls} stalls, {replayed} replayed launches, worst relative difference {worst:.2e}")
    assert fused >= calls and stalls >= 1 and replayed >= 1, (fused, stalls, replayed)  # the case the test is about has happened


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_random_pipelined_loops_equal_their_synchronous_twins(mrs, oracle, seed):
    """Random interleavings of what a publisher loop does — ticks with collisions (1-3 per call), commands for random ranges (staged rows or
    mrs_swarm_set_input: both only DRAIN the queued launches, a pending collision tick stays with the next fused launch), input timeouts,
    forces from outside (these settle), pipelined downloads waited for at random later moments (up to two in flight) — on one swarm,
    against a twin that does the same with synchronous calls only.  Fast UAVs make launches stall under the packs.  Every payload and
    the final state must agree."""
    import bench
    rng = np.random.default_rng(100 + seed)
    n = 6_000
    st, cmd = bench.make_inputs(n, "position+collisions", seed=40 + seed, volume_per_uav=14.0)
    st["v"][:6] = [0.0, 160.0, 0.0]

    def make():
        g = mrs.Swarm(n, arith=mrs.ARITH_FAST)
        g.construct(0, n, mrs.model_params("x500", ground_enabled=True, ground_z=0.0))
        g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        g.set_input(0, n, mrs.POSITION_CMD, cmd)
        return g

    a, b = make(), make()
    in_flight, checked, worst = [], 0, 0.0  # (ticket, expected payload)

    def check(k):
        nonlocal checked, worst
        ticket, want = in_flight.pop(k)
        got = b.outputs_wait(ticket)
        for f in got.dtype.names:
            worst = max(worst, helpers.rel_linf(got[f], want[f]))
            helpers.assert_close(got[f], want[f], 1e-9, f"seed {seed}, download {checked}: {f}")
        checked += 1

    for it in range(70):
        op = rng.integers(0, 10)
        if op <= 3:  # ticks
            k = int(rng.integers(1, 4))
            a.tick_n(0.001, k, True, False, 100.0)
            b.tick_n(0.001, k, True, False, 100.0)
        elif op <= 5:  # commands for a range
            first = int(rng.integers(0, n - 200))
            count = int(rng.integers(1, 200))
            goal = np.concatenate([st["x"][first:first + count] + rng.uniform(-5, 5, (count, 3)), rng.uniform(-3, 3, (count, 1))], axis=1)
            a.set_input(first, count, mrs.POSITION_CMD, goal)
            if rng.integers(0, 2):
                rows = b.input_staging(count, 4)
                rows[:] = goal
                b.commit_input(first, count, mrs.POSITION_CMD, 4)
            else:
                b.set_input(first, count, mrs.POSITION_CMD, goal)
        elif op == 6:  # a download, pipelined on b
            if len(in_flight) == 2:
                check(0)
            want = a.get_outputs().copy()
            in_flight.append((b.get_outputs_async(), want))
        elif op == 7 and in_flight:
            check(int(rng.integers(0, len(in_flight))) if len(in_flight) == 1 else 0)
        elif op == 8:
            first = int(rng.integers(0, n - 50))
            if rng.integers(0, 2):
                a.timeout_input(first, 50)
                b.timeout_input(first, 50)
            else:
                f = rng.normal(0, 1.0, (50, 3))
                a.apply_force(first, 50, f)
                b.apply_force(first, 50, f)
        else:
            hold = bool(rng.integers(0, 2))
            first = int(rng.integers(0, n - 30))
            a.set_hold(first, 30, hold)
            b.set_hold(first, 30, hold)
    while in_flight:
        check(0)
    sa, sb = a.get_state(), b.get_state()
    for k in sa:
        helpers.assert_close(sb[k], sa[k], 1e-9, f"seed {seed}: final {k}")
    helpers.assert_close(b.get_pid(), a.get_pid(), 1e-9, "final PID state")
    fused, stalls, replayed, ahead = b.fused_stats()
    print(f"seed {seed}: {checked} pipelined downloads checked (worst {worst:.1e}); pipelined swarm: {fused} fused launches, {stalls} stalls, {replayed} replayed")
    assert checked >= 3 and fused >= 20
