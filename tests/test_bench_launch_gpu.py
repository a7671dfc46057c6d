"""The launches bench.py times, put against the oracle DIRECTLY (they were only checked transitively).

bench.py's default workload is BASELINE configs[2] — 100 000 x500 UAVs, ARITH_FAST, mrs_swarm_step_n with many steps — which the
library issues as `mrs_uav_model_step_buf_nt_fast` on TWO streams (half the blocks each).  UAVs are independent inside makeStep,
so a seeded sample of lanes stepped alone by the oracle must agree with the same lanes of the big launch; tolerance = north_star's
1e-6, per UAV (helpers.per_uav_linf) and per field over the swarm."""
import numpy as np
import pytest

import helpers
from helpers import RTOL_NORTH_STAR

pytestmark = pytest.mark.gpu
DT = 0.001


def _bench_inputs(n, workload, seed):
    import bench  # the bench's own generator: same states and commands as the timed run
    return bench.make_inputs(n, workload, seed)


def _oracle_sample(oracle, st, cmd, pick, mode, n_steps):
    m = len(pick)
    o = oracle.OracleSwarm(m)
    o.construct(0, m, helpers.oracle_params("x500", ground_enabled=True))
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(o, nm)(0, m)
    o.set_state(0, m, st["x"][pick], st["v"][pick], st["R"][pick], st["omega"][pick], st["motor_rpm"][pick])
    o.set_input(0, m, mode, cmd[pick])
    o.step_n(DT, n_steps, 8)
    out = o.get_state()
    out["imu"], out["pid"] = o.get_imu(), o.get_pid()
    return out


def _gpu_sample(g, pick):
    full = g.get_state()
    out = {k: v[pick] for k, v in full.items()}
    out["imu"], out["pid"] = g.get_imu()[pick], g.get_pid()[pick]
    return out, full


@pytest.mark.parametrize("workload,n,steps", [("actuator", 100_000, 120), ("position", 100_000, 120), ("actuator", 1_000_000, 40),
                                              ("position", 50_000, 120), ("actuator", 4_000_000, 24), ("position", 2_000_000, 24)])
def test_bench_launch_against_oracle(mrs, oracle, workload, n, steps):
    """100 k actuator: *_model_step_buf_nt_fast x 2 streams; 100 k position: mrs_uav_step_buf_fast x 2 streams; 1 M actuator:
    *_model_step_buf_w3_fast x 2 streams; 50 k position: mrs_uav_step_buf_nt_fast on one stream (< 1024 blocks); 4 M actuator and
    2 M position (>= 1.4 GB moved per step, the HBM-streaming sizes of profiles/): the non-temporal two-wave kernels again."""
    M = mrs
    rng = np.random.default_rng(11)
    st, cmd = _bench_inputs(n, workload, seed=3)
    mode = M.ACTUATOR_CMD if workload == "actuator" else M.POSITION_CMD
    g = M.Swarm(n, arith=M.ARITH_FAST)
    g.construct(0, n, M.model_params("x500", ground_enabled=True))
    g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    g.set_input(0, n, mode, cmd)
    g.step_n(DT, steps)  # >= 4 launches and >= 1024 blocks: the split-stream form bench.py times
    pick = np.sort(rng.choice(n, 2048, replace=False))
    pick[:2], pick[-1] = [0, 63], n - 1
    nb = (n + 63) // 64
    pick[2:6] = [(nb // 2) * 64 - 1, (nb // 2) * 64, (nb // 2) * 64 + 63, n - 2]  # both sides of the two streams' block boundary
    pick = np.unique(pick)
    ref = _oracle_sample(oracle, st, cmd, pick, oracle.ACTUATOR_CMD if workload == "actuator" else oracle.POSITION_CMD, steps)
    got, full = _gpu_sample(g, pick)
    for k in ("x", "v", "R", "omega", "motor_rpm", "imu") + (("pid",) if workload == "position" else ()):
        helpers.assert_close(got[k], ref[k], RTOL_NORTH_STAR, f"{workload} {n}: {k}")
    worst, err = helpers.assert_close_per_uav(got, ref, RTOL_NORTH_STAR, f"{workload} {n}")
    print(f"{workload} {n} UAVs x {steps} steps (FAST, bench launch form): worst UAV {pick[worst]} per-UAV error {err:.2e}")
    assert np.all(np.isfinite(full["x"]))
    assert np.allclose(np.einsum("nij,nik->njk", full["R"], full["R"]), np.eye(3), atol=1e-9)


@pytest.mark.parametrize("exported", [False, True])
def test_bench_regions_of_twenty_steps(mrs, oracle, exported):
    """The driver's `bench.py --steps 20 --warmup 5`: regions of 20 steps, each behind mrs_swarm_synchronize.  A run of steps that
    follows a synchronize starts its second stream WITHOUT the fork event (both streams are idle); once the caller holds the
    stream's handle (mrs_swarm_stream) the library asks the stream first.  Both ways against the oracle."""
    M = mrs
    n, regions = 100_000, 6
    rng = np.random.default_rng(12)
    st, cmd = _bench_inputs(n, "position", seed=3)
    g = M.Swarm(n, arith=M.ARITH_FAST)
    g.construct(0, n, M.model_params("x500", ground_enabled=True))
    g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    g.set_input(0, n, M.POSITION_CMD, cmd)
    if exported:
        assert g.stream()
    g.step_n(DT, 5)
    for _ in range(regions):
        g.synchronize()
        g.step_n(DT, 20)
    g.synchronize()
    nb = (n + 63) // 64
    pick = np.unique(np.concatenate([np.sort(rng.choice(n, 1024, replace=False)), [0, (nb // 2) * 64 - 1, (nb // 2) * 64, n - 1]]))
    ref = _oracle_sample(oracle, st, cmd, pick, oracle.POSITION_CMD, 5 + 20 * regions)
    got, _ = _gpu_sample(g, pick)
    for k in ("x", "v", "R", "omega", "motor_rpm", "pid"):
        helpers.assert_close(got[k], ref[k], RTOL_NORTH_STAR, f"regions of 20, exported={exported}: {k}")
    helpers.assert_close_per_uav(got, ref, RTOL_NORTH_STAR, f"regions of 20, exported={exported}")


# ---- BASELINE config 4 in the form `bench.py --workload position+collisions` (and its `config4` sub-record) times ----------------
def _tainted(x_all, in_sample, tainted, reach):
    """One checkpoint of the closure argument: sample UAVs with an OUTSIDE UAV within `reach` become tainted, and taint spreads
    along the `reach`-graph among the sample UAVs (to its fixed point).  Two UAVs farther apart than `reach` at a checkpoint cannot
    touch before the next one (reach = contact range + both UAVs' largest displacement in between)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree
    s_idx = np.flatnonzero(in_sample)
    ts, to = cKDTree(x_all[s_idx]), cKDTree(x_all[~in_sample])
    exposed = np.array([len(h) > 0 for h in ts.query_ball_tree(to, reach)])
    t = tainted[s_idx] | exposed
    pairs = ts.query_pairs(reach, output_type="ndarray")
    g = coo_matrix((np.ones(len(pairs), dtype=np.int8), (pairs[:, 0], pairs[:, 1])), shape=(len(s_idx), len(s_idx)))
    _, lab = connected_components(g, directed=False)
    bad = np.zeros(lab.max() + 1, dtype=bool)
    bad[lab[t]] = True
    out = tainted.copy()
    out[s_idx] = bad[lab]
    return out


@pytest.mark.parametrize("volume", [64.0, 16.0])
def test_config4_bench_form_against_oracle(mrs, oracle, volume):
    """100 000 x500 UAVs, FAST, mrs_swarm_tick_n over 300 ticks: `mrs_uav_step_coll_buf_fast` launches, neighbour searches queued
    ahead of time and — thanks to a few fast UAVs — at least one stall with replayed launches.  Checked against
      (1) the oracle on the UAVs of a sub-box, of which only those are compared that provably never felt a UAV outside the sample:
          the run is cut into chunks of 25 (10) ticks; at every cut the positions of ALL UAVs are read back and a sample UAV closer than
          contact range + 2 x (largest displacement of a chunk) to an outside UAV, or to a tainted sample UAV, is tainted from then on;
      (2) the same 300 ticks in ONE tick_n call on a second swarm (the form the bench times; the chunked run settles at every cut)."""
    M = mrs
    n, ticks, vcap = 100_000, 300, 14.0
    chunk = 25 if volume == 64.0 else 10  # the denser swarm needs the shorter reach (contact range + 2 * vcap * chunk * dt) to keep taint local
    import bench
    st, cmd = bench.make_inputs(n, "position+collisions", seed=3, volume_per_uav=volume)
    rng = np.random.default_rng(44)
    side = (volume * n) ** (1.0 / 3.0)
    # a few UAVs far faster than the warning threshold allows for: the lists go stale before the queued search runs (stall + replay)
    fast = np.flatnonzero(st["x"][:, 0] > 1.8 * side)[:12]
    st["v"][fast] = rng.normal(0, 1, (len(fast), 3)) + [0.0, 28.0, 0.0]
    lo = np.array([0.15 * side, 0.15 * side, 5.0])
    want = 14_000 if volume == 64.0 else 9_000
    edge = 2.0 * side * (want / n) ** 0.5  # the air space is 2 side x 2 side x side/4: a column of the full height holding ~`want` UAVs
    in_sample = np.all((st["x"][:, :2] >= lo[:2]) & (st["x"][:, :2] < lo[:2] + [edge, edge]), axis=1)
    pick = np.flatnonzero(in_sample)
    assert 4000 < len(pick) < 40_000 and not in_sample[fast].any()

    def make():
        g = M.Swarm(n, arith=M.ARITH_FAST)
        g.construct(0, n, M.model_params("x500", ground_enabled=True))
        g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        g.set_input(0, n, M.POSITION_CMD, cmd)
        return g

    # (2) first: one call, the bench's form
    one = make()
    one.tick_n(DT, ticks, True, False, 100.0)
    fused, stalls, replayed, ahead = one.fused_stats()
    so = one.get_state()
    so["f"], so["pid"], so["imu"], so["crashed"] = one.get_external_force(), one.get_pid(), one.get_imu(), one.has_crashed()
    _, searches = one.collision_stats()
    del one
    assert fused >= 0.8 * ticks and ahead >= 3, (fused, stalls, replayed, ahead, searches)
    assert stalls >= 1 and replayed >= 1, (fused, stalls, replayed, ahead)
    # (1) chunked, with the closure bookkeeping
    g = make()
    tainted = np.zeros(n, dtype=bool)
    reach = 0.95 + 2.0 * vcap * chunk * DT
    for c in range(ticks // chunk):
        s = g.get_state()
        slow = np.ones(n, dtype=bool)
        slow[fast] = False
        assert np.linalg.norm(s["v"][slow], axis=1).max() < vcap - 3.0, "the closure argument needs the ordinary UAVs below the speed cap"
        # the fast ones: nowhere near the sample (they fly 28 m/s * 0.3 s = 9 m in the whole run)
        d = np.linalg.norm(s["x"][fast][:, None, :2] - (lo[:2] + edge / 2)[None, None, :], axis=2).min()
        assert d > edge + 20.0
        tainted = _tainted(s["x"], in_sample, tainted, reach)
        g.tick_n(DT, chunk, True, False, 100.0)
    sg = g.get_state()
    sg["f"], sg["pid"], sg["imu"] = g.get_external_force(), g.get_pid(), g.get_imu()
    del g
    for k in ("x", "v", "R", "omega", "motor_rpm", "f", "pid", "imu"):  # two call patterns, different kernel instantiations on the way
        helpers.assert_close(sg[k], so[k], 1e-9, f"chunked vs one call: {k}")
    so = {k: v for k, v in so.items() if k != "v_prev"}
    helpers.assert_close_per_uav(sg, so, 1e-9, "chunked vs one call", fields=("x", "v", "R", "omega", "motor_rpm", "f"))
    m = len(pick)
    o = oracle.OracleSwarm(m)
    o.construct(0, m, helpers.oracle_params("x500", ground_enabled=True))
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(o, nm)(0, m)
    o.set_state(0, m, st["x"][pick], st["v"][pick], st["R"][pick], st["omega"][pick], st["motor_rpm"][pick])
    o.set_input(0, m, oracle.POSITION_CMD, cmd[pick])
    for _ in range(ticks):
        o.step_n(DT, 1, 8)
        o.handle_collisions(True, False, 100.0)
    ref = o.get_state()
    ref["f"], ref["pid"], ref["imu"], ref["crashed"] = o.get_external_force(), o.get_pid(), o.get_imu(), o.has_crashed()
    ref = {k: v for k, v in ref.items() if k != "v_prev"}
    clean = ~tainted[pick]
    assert clean.sum() > 2500, (clean.sum(), m)
    got = {k: v[pick][clean] for k, v in so.items()}
    ref = {k: v[clean] for k, v in ref.items()}
    touched = int((np.abs(ref["f"]).sum(axis=1) > 0).sum())
    assert touched >= (3 if volume == 64.0 else 20), touched
    for k in ("x", "v", "R", "omega", "motor_rpm", "f", "pid", "imu"):
        helpers.assert_close(got[k], ref[k], RTOL_NORTH_STAR, f"config 4 at {volume} m^3: {k}")
    worst, err = helpers.assert_close_per_uav(got, ref, RTOL_NORTH_STAR, f"config 4 at {volume} m^3", fields=("x", "v", "R", "omega", "motor_rpm", "f"))
    assert np.array_equal(so["crashed"][pick][clean], ref["crashed"])
    print(f"config 4 bench form, {volume:g} m^3 per UAV: {ticks} ticks in one call = {fused} fused launches, {searches} searches ({ahead} queued ahead), "
          f"{stalls} stalls / {replayed} launches replayed; oracle sample {m} UAVs, {int(clean.sum())} provably closed, {touched} of them under a "
          f"collision force at the end, worst per-UAV error {err:.2e}")
