"""The boundary-UAV ("export set") collision exchange of the multi-GPU path (north_star's "all-gather of boundary-UAV
positions") on VIRTUAL SHARDS: `world` swarms on the one GPU of the test box, each driven by its own host thread through
mrs_swarm_tick_sharded_n, exchanging through an in-process loopback group (device-to-device copies in place of RCCL's all-gather).
Kernels, host protocol (batches of fused launches, stall word in the collective's headers, search + export-set derivation, replay)
and the order of the collectives are those of the 8-GPU run.

Checked: results identical to the full all-gather of every record on every tick and to the single-swarm oracle over hundreds of
moving ticks (elastic and crash collisions), with spatially sorted slabs (public index unchanged through the permutation) and with
plain index shards; the statistics show that most ticks gathered only the export sets."""
import threading

import numpy as np
import pytest

import helpers
from helpers import RTOL_LITERAL

pytestmark = pytest.mark.gpu
DT = 0.001


def run_ranks(fns):
    """one host thread per rank (tick_sharded_n is collective); the first exception is re-raised"""
    errs = [None] * len(fns)

    def wrap(k):
        try:
            fns[k]()
        except BaseException as e:  # noqa: BLE001
            errs[k] = e

    th = [threading.Thread(target=wrap, args=(k,)) for k in range(len(fns))]
    for t in th:
        t.start()
    for t in th:
        t.join(300)
    assert not any(t.is_alive() for t in th), "a rank hangs in a collective"
    # the rank that failed first makes its peers fail in the group's all-gather: report the cause, not the echo
    real = [e for e in errs if e is not None and "loopback all-gather" not in str(e)]
    seen = [e for e in errs if e is not None]
    if seen:
        raise (real + seen)[0] from RuntimeError(" | ".join(f"rank {k}: {e}" for k, e in enumerate(errs) if e is not None))


class VirtualShards:
    """`world` shards of one swarm given in PUBLIC index order; order[k] = public index at sorted position k"""

    def __init__(self, M, world, order, po, pos, heading, st, mode, cmd, arith, exchange, rendezvous=False):
        from mrs_multirotor_simulator_amd.sharded import shard_range
        self.M, self.world, self.order, self.n_total = M, world, order, len(order)
        self.group = M.LoopbackGroup(world)
        if rendezvous:
            self.group.set_rendezvous(True)
        self.shards = []
        for r in range(world):
            lo, hi = shard_range(self.n_total, world, r)
            idx = order[lo:hi]
            g = M.Swarm(hi - lo, arith=arith)
            if hi > lo:
                g.construct(0, hi - lo, po, None if pos is None else pos[idx], None if heading is None else heading[idx])
                g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
                g.set_input(0, hi - lo, mode, cmd[idx])
            g.comm_init_loopback(self.group, r, self.n_total)
            g.set_exchange(exchange)
            self.shards.append((g, idx))

    def tick_n(self, n, enabled, crash, rebounce):
        run_ranks([(lambda g=g: g.tick_sharded_n(DT, n, enabled, crash, rebounce)) for g, _ in self.shards])

    def gather(self):
        """state in PUBLIC index order"""
        out = {}
        for g, idx in self.shards:
            if len(idx) == 0:
                continue
            s = g.get_state()
            s["f"], s["crashed"], s["pid"], s["imu"] = g.get_external_force(), g.has_crashed(), g.get_pid(), g.get_imu()
            for k, v in s.items():
                out.setdefault(k, np.zeros((self.n_total,) + v.shape[1:], dtype=v.dtype))[idx] = v
        return out

    def info(self):
        return [g.comm_info() for g, _ in self.shards]

    def close(self):
        run_ranks([(lambda g=g: g.comm_destroy()) for g, _ in self.shards])


def moving_swarm(rng, n_total, speed=5.0):
    side = (64.0 * n_total) ** (1.0 / 3.0)
    pos = rng.uniform(0, side, (n_total, 3)) + [0, 0, 30]
    k = n_total // 15
    pos[:k] = pos[k:2 * k] + rng.normal(0, 0.3, (k, 3))
    st = helpers.random_state(rng, n_total, 4, tilted=True)
    st["x"] = pos
    st["v"] = rng.normal(0, speed, (n_total, 3))
    cmd = rng.uniform(0.4, 0.55, (n_total, 4))
    return pos, st, cmd


@pytest.mark.parametrize("world,slabs,n_total", [(2, True, 3001), (3, False, 2000), (4, True, 4097)])
def test_export_set_exchange_matches_full_gather_and_oracle(mrs, oracle, world, slabs, n_total):
    M = mrs
    rng = np.random.default_rng(1000 + world)
    pos, st, cmd = moving_swarm(rng, n_total)
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world) if slabs else np.arange(n_total)
    pp = helpers.to_product_params(M, po)
    mk = lambda x: VirtualShards(M, world, order, pp, pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_LITERAL, x)
    ex, full = mk(M.EXCHANGE_EXPORT_SETS), mk(M.EXCHANGE_FULL_GATHER)
    blocks = [(70, False), (1, True), (49, False), (80, False)]  # 200 ticks, one of them in crash mode
    done = 0
    for n, crash in blocks:
        ex.tick_n(n, True, crash, 100.0)
        full.tick_n(n, True, crash, 100.0)
        for _ in range(n):
            o.step(DT)
            o.handle_collisions(True, crash, 100.0)
        done += n
        a, b, so = ex.gather(), full.gather(), o.get_state()
        assert np.array_equal(a["crashed"], o.has_crashed()) and np.array_equal(b["crashed"], o.has_crashed()), f"crash flags after {done} ticks"
        fo = o.get_external_force()
        # forces of UAVs with several partners are summed in ascending RECORD order: slab order is not index order (<= 1 ulp per term)
        helpers.assert_close(a["f"], fo, 1e-11, f"forces after {done} ticks")
        helpers.assert_close(b["f"], fo, 1e-11, f"forces (full gather) after {done} ticks")
        for k in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(a[k], so[k], RTOL_LITERAL, f"{k} after {done} ticks")
            helpers.assert_close(a[k], b[k], 1e-12, f"{k}: export sets vs full gather after {done} ticks")
        helpers.assert_close_per_uav(a, so, RTOL_LITERAL, f"after {done} ticks")
    assert o.has_crashed().sum() > 0 and (np.abs(o.get_external_force()).sum(axis=1) > 0).sum() > 30
    for r, ci in enumerate(ex.info()):
        assert ci["exchange"] == M.EXCHANGE_EXPORT_SETS and ci["ticks"] == 200
        assert 2 <= ci["searches"] <= 80, ci
        # the ordinary tick moves a fraction of the full exchange (index shards of a few hundred UAVs: most UAVs have a foreign partner
        # within the 2.7 m the sharded lists reach, so only "less")
        assert ci["bytes_per_tick"] < ci["bytes_per_rebuild"] / (2 if slabs else 1), ci
    print("export-set exchange:", ex.info()[0])
    ex.close()
    full.close()


def test_export_sets_follow_host_writes_and_ragged_shards(mrs, oracle):
    """set_state between two sharded runs invalidates the export lists; n_total not divisible by the world size; one UAV on hold"""
    M = mrs
    world, n_total = 3, 1001
    rng = np.random.default_rng(7)
    pos, st, cmd = moving_swarm(rng, n_total, speed=2.0)
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world)
    vs = VirtualShards(M, world, order, helpers.to_product_params(M, po), pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_LITERAL,
                       M.EXCHANGE_EXPORT_SETS)
    where = {int(p): (r, k) for r, (_, idx) in enumerate(vs.shards) for k, p in enumerate(idx)}

    def both_ticks(n):
        vs.tick_n(n, True, False, 100.0)
        for _ in range(n):
            o.step(DT)
            o.handle_collisions(True, False, 100.0)

    both_ticks(30)
    # teleport public UAV 5 next to public UAV 400 (different slabs or not: they were nowhere near each other)
    so = o.get_state()
    tele = {k: v[5:6].copy() for k, v in so.items()}
    tele["x"][0] = so["x"][400] + [0.3, 0.1, -0.2]
    o.set_state(5, 1, tele["x"], tele["v"], tele["R"], tele["omega"], tele["motor_rpm"])
    r, k = where[5]
    vs.shards[r][0].set_state(k, 1, tele["x"], tele["v"], tele["R"], tele["omega"], tele["motor_rpm"])
    o.set_hold(77, 1, True)
    r, k = where[77]
    vs.shards[r][0].set_hold(k, 1, True)
    both_ticks(25)
    a, so = vs.gather(), o.get_state()
    fo = o.get_external_force()
    helpers.assert_close(a["f"], fo, 1e-11, "forces after the teleport")
    for key in ("x", "v", "R", "omega", "motor_rpm"):
        helpers.assert_close(a[key], so[key], RTOL_LITERAL, key)
    vs.close()


@pytest.mark.parametrize("world,n_total", [(3, 2), (4, 5), (2, 1)])
def test_export_sets_with_empty_and_tiny_shards(mrs, oracle, world, n_total):
    """more ranks than UAVs: a rank without UAVs still takes part in every collective and watches the headers"""
    M = mrs
    rng = np.random.default_rng(3)
    pos = np.array([[0.0, 0.0, 10.0], [0.5, 0.1, 10.0], [0.2, 0.6, 10.1], [30.0, 0.0, 10.0], [30.4, 0.2, 10.0]])[:n_total]
    st = helpers.random_state(rng, n_total, 4, tilted=True)
    st["x"] = pos
    st["v"] = rng.normal(0, 3.0, (n_total, 3))
    cmd = rng.uniform(0.4, 0.55, (n_total, 4))
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world)
    vs = VirtualShards(M, world, order, helpers.to_product_params(M, po), pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_LITERAL,
                       M.EXCHANGE_EXPORT_SETS)
    for n in (1, 40, 60):
        vs.tick_n(n, True, False, 100.0)
        for _ in range(n):
            o.step(DT)
            o.handle_collisions(True, False, 100.0)
        a, so = vs.gather(), o.get_state()
        helpers.assert_close(a["f"], o.get_external_force(), 1e-11, "forces")
        for key in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(a[key], so[key], RTOL_LITERAL, key)
    vs.close()
