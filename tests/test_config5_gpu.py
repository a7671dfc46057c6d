"""BASELINE configs[4] ("config 5") AT ITS SIZE on one GPU: 1 000 000 UAVs with mutual collisions, sharded 8 x 125 000.

The eight ranks of the real run are eight `Swarm` objects on the one device of the test box ("virtual shards"), x-sorted slabs of
the swarm, each driven by its own host thread through mrs_swarm_tick_sharded_n; an in-process loopback group stands in for RCCL's
all-gather (device-to-device copies).  Everything else — fused step + collision launches, export-set exchange, searches over the
full gather, stall / replay protocol, order of the collectives — is the code path of the 8-GPU run.  Checked against
  (1) a single 1 000 000-UAV swarm driven by mrs_swarm_tick_n (whole-swarm comparison), and
  (2) the CPU oracle on a sample that is CLOSED under interaction: whole connected components of the "closer than 2.5 m at
      t = 0" graph.  UAVs of different components start >= 2.5 m apart and move < 0.6 m in the test's ticks, so they never come
      within the 0.9 m collision range of each other: a component evolves exactly as it would alone, and the oracle (which
      needs seconds per tick for the whole million) only has to run the sampled components."""
import numpy as np
import pytest

import helpers
from helpers import RTOL_NORTH_STAR

pytestmark = pytest.mark.gpu
DT = 0.001
N_TOTAL, WORLD = 1_000_000, 8


def config5_scenario(n_total, seed=5, planted=4000):
    """bench.py's `position+collisions` generator at 64 m^3 per UAV, plus `planted` pairs that really touch."""
    import bench
    st, cmd = bench.make_inputs(n_total, "position+collisions", seed)
    rng = np.random.default_rng(seed + 1000)
    a = rng.choice(n_total, 2 * planted, replace=False)
    st["x"][a[:planted]] = st["x"][a[planted:]] + rng.normal(0, 0.3, (planted, 3))
    cmd[a[:planted], :3] = st["x"][a[:planted]] + rng.uniform(-5, 5, (planted, 3))
    return st, cmd, a[:planted]


def closed_sample(x, must_have, want, rng, link=2.5):
    """indices of whole connected components of the `link`-metre graph: those of `must_have` plus random ones up to ~`want`"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree
    n = len(x)
    pairs = cKDTree(x).query_pairs(link, output_type="ndarray")
    g = coo_matrix((np.ones(len(pairs), dtype=np.int8), (pairs[:, 0], pairs[:, 1])), shape=(n, n))
    _, lab = connected_components(g, directed=False)
    keep = np.zeros(lab.max() + 1, dtype=bool)
    keep[lab[must_have]] = True
    extra = rng.permutation(lab.max() + 1)
    sizes = np.bincount(lab)
    have = int(sizes[keep].sum())
    for c in extra:
        if have >= want:
            break
        if not keep[c]:
            keep[c] = True
            have += int(sizes[c])
    return np.flatnonzero(keep[lab])


def make_swarm(M, n, st, cmd, sl, arith):
    g = M.Swarm(n, arith=arith)
    g.construct(0, n, M.model_params("x500", ground_enabled=True))
    g.set_state(0, n, st["x"][sl], st["v"][sl], st["R"][sl], st["omega"][sl], st["motor_rpm"][sl])
    g.set_input(0, n, M.POSITION_CMD, cmd[sl])
    return g


def test_config5_one_million_uavs_eight_virtual_shards(mrs, oracle):
    from mrs_multirotor_simulator_amd.sharded import max_shard
    M = mrs
    ticks = 60
    rng = np.random.default_rng(50)
    st, cmd, planted = config5_scenario(N_TOTAL)
    pick = closed_sample(st["x"], planted[:1500], 24_000, rng)

    # (a) eight virtual shards (x-sorted slabs, public index kept through the permutation), the export-set exchange over an
    #     in-process loopback group: one host thread per rank, as eight processes would run it over RCCL
    from test_export_sets_gpu import VirtualShards
    order = M.slab_partition(st["x"], WORLD)
    vs = VirtualShards(M, WORLD, order, M.model_params("x500", ground_enabled=True), None, None, st, M.POSITION_CMD, cmd, M.ARITH_FAST,
                       M.EXCHANGE_EXPORT_SETS)
    vs.tick_n(ticks, True, False, 100.0)
    sh = vs.gather()
    stats = vs.info()
    vs.close()
    del vs
    # (b) the same million UAVs as ONE swarm
    one = make_swarm(M, N_TOTAL, st, cmd, slice(0, N_TOTAL), M.ARITH_FAST)
    one.tick_n(DT, ticks, True, False, 100.0)
    so = one.get_state()
    so["f"], so["pid"], so["imu"] = one.get_external_force(), one.get_pid(), one.get_imu()
    del one
    touched = int((np.abs(so["f"]).sum(axis=1) > 0).sum())
    assert touched > 2000, touched
    # different kernel instantiations (three-wave build at 1 M, two-wave at 125 k) choose their FMAs differently: ~1e-12 per step
    for k in ("x", "v", "R", "omega", "motor_rpm", "f", "pid", "imu"):
        helpers.assert_close(sh[k], so[k], 1e-9, f"8 shards vs one swarm: {k}")
    assert np.array_equal(sh["crashed"], np.zeros(N_TOTAL, dtype=np.int32))
    helpers.assert_close_per_uav(sh, so, 1e-9, "8 shards vs one swarm", fields=("x", "v", "R", "omega", "motor_rpm", "f"))
    assert np.abs(so["v"]).max() * ticks * DT < 0.6, "the closed-sample argument needs slow UAVs"

    # (c) the oracle on the closed sample
    m = len(pick)
    o = oracle.OracleSwarm(m)
    o.construct(0, m, helpers.oracle_params("x500", ground_enabled=True))
    o.set_state(0, m, st["x"][pick], st["v"][pick], st["R"][pick], st["omega"][pick], st["motor_rpm"][pick])
    o.set_input(0, m, oracle.POSITION_CMD, cmd[pick])
    for _ in range(ticks):
        o.step_n(DT, 1, 8)
        o.handle_collisions(True, False, 100.0)
    ref = o.get_state()
    ref["f"], ref["pid"], ref["imu"] = o.get_external_force(), o.get_pid(), o.get_imu()
    got = {k: v[pick] for k, v in sh.items()}
    assert (np.abs(ref["f"]).sum(axis=1) > 0).sum() > 500
    for k in ("x", "v", "R", "omega", "motor_rpm", "f", "pid", "imu"):
        helpers.assert_close(got[k], ref[k], RTOL_NORTH_STAR, f"shards vs oracle sample: {k}")
    worst, err = helpers.assert_close_per_uav(got, ref, RTOL_NORTH_STAR, "shards vs oracle sample",
                                              fields=("x", "v", "R", "omega", "motor_rpm", "f"))
    n_max = max_shard(N_TOTAL, WORLD)
    rank_of = np.empty(N_TOTAL, dtype=np.int64)
    rank_of[order] = np.minimum(np.arange(N_TOTAL) // n_max, WORLD - 1)
    per_shard = np.bincount(rank_of[pick], minlength=WORLD)
    assert per_shard.min() >= 2000, per_shard
    print(f"config 5: {N_TOTAL} UAVs, {WORLD} virtual shards, {ticks} ticks; {touched} UAVs under a collision force at the end; oracle sample "
          f"{m} UAVs ({per_shard.min()}..{per_shard.max()} per shard), worst per-UAV error {err:.2e} (UAV {pick[worst]}); "
          f"rank 0: {stats[0]}")
    for ci in stats:
        assert ci["ticks"] == ticks and ci["searches"] < ticks // 3, ci
        assert ci["bytes_per_tick"] * 8 < ci["bytes_per_rebuild"], ci  # slab shards: the boundary sets are a small part of a shard
