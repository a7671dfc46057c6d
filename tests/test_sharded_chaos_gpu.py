"""The lock-free multi-rank protocol of the sharded collision tick under HOST SKEW (every earlier multi-rank test forced
the hosts into lock-step).  `world` virtual shards on the one GPU, each driven by its own host thread through
mrs_swarm_tick_sharded_n, ticks in the SPLIT form (interior and boundary launches on two streams, tied inside the kernels), and
  * mrs_swarm_debug_chaos: every rank's host sleeps a random 0..300 us before every launch and, at random, decides on the stall and
    warning words as it read them one launch earlier — what a slow PCIe read or a descheduled thread does to a real rank;
  * the loopback group in RENDEZVOUS mode: no host barrier inside the all-gather (a rank only waits for its peers to ARRIVE at the
    same collective), and in its ordinary two-barrier mode;
  * UAVs fast enough for announced stalls, warnings and searches every few ticks.
500 ticks in calls of uneven length, the whole swarm on the oracle, LITERAL arithmetic held to 1e-11.  A protocol slip shows as a
collective mismatch (an error or a time-out of the group), a missed stall as a wrong force."""
import numpy as np
import pytest

import helpers
from helpers import RTOL_LITERAL
from test_export_sets_gpu import VirtualShards, moving_swarm

pytestmark = pytest.mark.gpu
DT = 0.001


@pytest.mark.parametrize("world,rendezvous,n_total", [(4, True, 5000), (8, True, 7200), (4, False, 5000), (8, False, 7200)])
def test_sharded_ticks_under_host_skew(mrs, oracle, monkeypatch, world, rendezvous, n_total):
    M = mrs
    monkeypatch.setenv("MRS_SHARD_SPLIT_MIN_BLOCKS", "1")  # (read when a swarm is created: small shards take the split form too,
    monkeypatch.setenv("MRS_SHARD_SPLIT_MAX_FRACTION", "0.95")  #  even when most of their blocks hold a boundary UAV)
    rng = np.random.default_rng(4000 + world + int(rendezvous))
    pos, st, cmd = moving_swarm(rng, n_total, speed=6.0)
    hot = rng.choice(n_total, 40, replace=False)          # skin used up within a dozen ticks: announcements and searches all the time
    st["v"][hot] = rng.normal(0, 1, (40, 3)) * [12.0, 12.0, 4.0]
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world)
    vs = VirtualShards(M, world, order, helpers.to_product_params(M, po), pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_LITERAL,
                       M.EXCHANGE_EXPORT_SETS, rendezvous=rendezvous)
    for r, (g, _) in enumerate(vs.shards):
        g.debug_chaos(300, seed=17 * world + r)
    done = 0
    for n in (37, 1, 110, 5, 160, 62, 125):  # 500 ticks
        vs.tick_n(n, True, False, 100.0)
        for _ in range(n):
            o.step_n(DT, 1, 8)
            o.handle_collisions(True, False, 100.0)
        done += n
        a, so, fo = vs.gather(), o.get_state(), o.get_external_force()
        helpers.assert_close(a["f"], fo, 1e-11, f"forces after {done} ticks")
        for k in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(a[k], so[k], RTOL_LITERAL, f"{k} after {done} ticks")
        helpers.assert_close_per_uav(a, so, RTOL_LITERAL, f"after {done} ticks")
    assert done == 500 and (np.abs(fo).sum(axis=1) > 0).sum() > 30
    info, split = vs.info(), [g.split_stats() for g, _ in vs.shards]
    vs.close()
    for ci, (ticks_split, nbnd) in zip(info, split):
        assert ci["ticks"] == 500 and 5 <= ci["searches"] <= 250, ci
    assert sum(t for t, _ in split) > 40 * world, split  # the split form really ran (4 ticks after every search and call are serial)
    print(f"chaos, world {world}, {'rendezvous' if rendezvous else 'barrier'} loopback: searches {[ci['searches'] for ci in info]}, "
          f"ticks replayed {[ci['noop_ticks'] for ci in info]}, split ticks / boundary blocks {split}")


def test_split_ticks_on_reserved_compute_units(mrs, oracle, monkeypatch):
    """MRS_SPLIT_CU_RESERVE (boundary chain and collective on a CU-masked stream of their own, interior launch on the other CUs —
    a switch, off by default: MEASUREMENTS §5.6): two more streams, same results."""
    M = mrs
    monkeypatch.setenv("MRS_SHARD_SPLIT_MIN_BLOCKS", "1")
    monkeypatch.setenv("MRS_SHARD_SPLIT_MAX_FRACTION", "0.95")
    monkeypatch.setenv("MRS_SPLIT_CU_RESERVE", "32")
    world, n_total = 2, 3000
    rng = np.random.default_rng(78)
    pos, st, cmd = moving_swarm(rng, n_total, speed=5.0)
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world)
    vs = VirtualShards(M, world, order, helpers.to_product_params(M, po), pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_LITERAL,
                       M.EXCHANGE_EXPORT_SETS, rendezvous=True)
    done = 0
    for n in (50, 3, 97):
        vs.tick_n(n, True, False, 100.0)
        for _ in range(n):
            o.step(DT)
            o.handle_collisions(True, False, 100.0)
        done += n
        a, so = vs.gather(), o.get_state()
        helpers.assert_close(a["f"], o.get_external_force(), 1e-11, f"forces after {done} ticks")
        for k in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(a[k], so[k], RTOL_LITERAL, f"{k} after {done} ticks")
    split = [g.split_stats()[0] for g, _ in vs.shards]
    vs.close()
    assert min(split) > 60, split


def test_split_ticks_with_motor_speeds_beyond_max_rpm(mrs, oracle, monkeypatch):
    """Motor speeds beyond the airframe's max_rpm (set through set_state; or max_rpm lowered under running motors): the displacement
    bound behind the announcements carries the thrust of the current step next to the max_rpm cap (pred_thr, evaluated per UAV on the
    device), so it HOLDS for such UAVs — no sticky host flag, no search every five ticks for the life of the swarm —
    and the results are exactly the oracle's."""
    M = mrs
    monkeypatch.setenv("MRS_SHARD_SPLIT_MIN_BLOCKS", "1")
    monkeypatch.setenv("MRS_SHARD_SPLIT_MAX_FRACTION", "0.95")
    world, n_total = 2, 3000
    rng = np.random.default_rng(77)
    pos, st, cmd = moving_swarm(rng, n_total, speed=3.0)
    over = rng.choice(n_total, 60, replace=False)
    st["motor_rpm"][over, :4] = rng.uniform(8000.0, 12000.0, (60, 4))  # x500: max_rpm 7800 (airframes.py); the low-pass brings them back
    cmd[over[:20]] = 1.0                                                # ... towards full throttle for some: they stay above max_rpm for good
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    assert st["motor_rpm"][over].max() > po.max_rpm
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world)
    vs = VirtualShards(M, world, order, helpers.to_product_params(M, po), pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_LITERAL,
                       M.EXCHANGE_EXPORT_SETS)
    done = 0
    for n in (80, 40):
        vs.tick_n(n, True, False, 100.0)
        for _ in range(n):
            o.step(DT)
            o.handle_collisions(True, False, 100.0)
        done += n
        a, so = vs.gather(), o.get_state()
        helpers.assert_close(a["f"], o.get_external_force(), 1e-11, f"forces after {done} ticks")
        for k in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(a[k], so[k], RTOL_LITERAL, f"{k} after {done} ticks")
    assert so["motor_rpm"][over[:20], :4].min() > po.max_rpm  # still over-speed at the end
    info, split = vs.info(), [g.split_stats()[0] for g, _ in vs.shards]
    vs.close()
    assert max(ci["searches"] for ci in info) <= 12, info   # (the old sticky flag: a search about every 5 ticks = 24+)
    assert min(split) > 60, split                             # the split form ran
