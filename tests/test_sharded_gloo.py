"""world_size-2 `gloo` tests (CPU) of the multi-GPU collision exchange — LAYOUT ONLY: shard ranges, the padded all-gather layout of
the Python orchestrator (mrs_multirotor_simulator_amd/sharded.py) and that shard-local collision results equal the single-process
result.  The local engine here is the CPU oracle (test infrastructure) behind the same three calls the GPU engine implements, so
this file does NOT exercise mrs_swarm_tick_sharded_n.  The library's own protocol is covered elsewhere: its host decisions
(csrc/sharded_protocol.h) on a multi-rank CPU model under random host skew in tests/test_sharded_protocol.py, the whole protocol in
separate PROCESSES on the GPU box in tests/test_sharded_multiprocess_gpu.py / test_peer_window_gpu.py / test_sharded_chaos_gpu.py,
and bench.py's N > 1 path end to end in tests/test_bench_rehearsal_gpu.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_shard_ranges_cover_everything():
    from mrs_multirotor_simulator_amd.sharded import max_shard, shard_range
    for n in (0, 1, 7, 64, 1000, 100_003):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == (max_shard(n, w) if n else 0)


class OracleEngine:
    """Each rank keeps a full oracle replica but OWNS only [lo, hi): foreign positions arrive through the all-gather."""

    def __init__(self, O, sw, lo, hi, n_total, world, n_max):
        self.O, self.sw, self.lo, self.hi, self.n_total, self.world, self.n_max = O, sw, lo, hi, n_total, world, n_max

    def step(self, dt):
        self.sw.step(dt)

    def write_records(self, out):
        st = self.sw.get_state(self.lo, self.hi - self.lo)
        rec = np.zeros((self.hi - self.lo, 6))
        rec[:, :3] = st["x"]
        for k in range(self.hi - self.lo):
            p = self.sw.get_params(self.lo + k)
            rec[k, 3:] = (p.mass, p.arm_length, p.prop_radius)
        out.copy_(torch.from_numpy(rec))

    def collide(self, records, n_records, my_offset, enabled, crash, rebounce):
        from mrs_multirotor_simulator_amd.sharded import shard_range
        rec = records.numpy()
        assert n_records == self.world * self.n_max and my_offset == dist.get_rank() * self.n_max
        # install the gathered foreign positions into the replica, then run the reference semantics on all UAVs
        for r in range(self.world):
            lo, hi = shard_range(self.n_total, self.world, r)
            blk = rec[r * self.n_max: r * self.n_max + (hi - lo)]
            assert np.all(np.isnan(rec[r * self.n_max + (hi - lo): (r + 1) * self.n_max]))  # padding
            if r != dist.get_rank() and hi > lo:
                st = self.sw.get_state(lo, hi - lo)
                self.sw.set_state(lo, hi - lo, blk[:, :3].copy(), st["v"], st["R"], st["omega"], st["motor_rpm"])
        self.sw.handle_collisions(enabled, crash, rebounce)


def _worker(rank, world, port, n_total, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers
    from mrs_multirotor_simulator_amd.sharded import ShardedSwarm, max_shard, shard_range
    from oracle import oracle_swarm as O
    rng = np.random.default_rng(1234)  # same scenario on every rank
    pos = rng.uniform(0, 9, (n_total, 3)) + [0, 0, 20]
    goal = np.concatenate([pos + rng.uniform(-2, 2, (n_total, 3)), np.zeros((n_total, 1))], axis=1)
    po = helpers.oracle_params("x500")
    sw = O.OracleSwarm(n_total)
    sw.construct(0, n_total, po, pos, np.zeros(n_total))
    sw.set_input(0, n_total, O.POSITION_CMD, goal)
    lo, hi = shard_range(n_total, world, rank)
    eng = OracleEngine(O, sw, lo, hi, n_total, world, max_shard(n_total, world))
    sh = ShardedSwarm(n_total, eng, torch.device("cpu"))
    assert (sh.lo, sh.hi) == (lo, hi)
    assert [sh.global_index(k) for k in (0, sh.n_max - 1, sh.n_max)] == [0, (sh.n_max - 1 if sh.n_max - 1 < shard_range(n_total, world, 0)[1] else -1), shard_range(n_total, world, 1)[0]]
    sh.tick_n(0.001, 12, True, False, 100.0)
    st = sw.get_state(lo, hi - lo)
    np.savez(os.path.join(result_dir, f"rank{rank}.npz"), x=st["x"], v=st["v"], f=sw.get_external_force(lo, hi - lo), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [257, 300])
def test_sharded_ticks_match_single_process(tmp_path, oracle, n_total):
    import helpers
    world = 2
    import socket
    with socket.socket() as sk:  # a port the kernel reports free right now
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    O = oracle
    rng = np.random.default_rng(1234)
    pos = rng.uniform(0, 9, (n_total, 3)) + [0, 0, 20]
    goal = np.concatenate([pos + rng.uniform(-2, 2, (n_total, 3)), np.zeros((n_total, 1))], axis=1)
    ref = O.OracleSwarm(n_total)
    ref.construct(0, n_total, helpers.oracle_params("x500"), pos, np.zeros(n_total))
    ref.set_input(0, n_total, O.POSITION_CMD, goal)
    for _ in range(12):
        ref.step(0.001)
        ref.handle_collisions(True, False, 100.0)
    st, f = ref.get_state(), ref.get_external_force()
    assert (np.abs(f).sum(axis=1) > 0).sum() > 10  # the scenario really collides
    covered = 0
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        lo, hi = int(d["lo"]), int(d["hi"])
        covered += hi - lo
        assert np.array_equal(d["x"], st["x"][lo:hi]) and np.array_equal(d["v"], st["v"][lo:hi]) and np.array_equal(d["f"], f[lo:hi])
    assert covered == n_total


# ------------------------------------------------------------------------------------------------------------------------------
# export-set exchange (boundary UAVs only between two searches): the Python form of the protocol, gloo, world 2 and 4
# ------------------------------------------------------------------------------------------------------------------------------
class OracleExportEngine:
    """Full oracle replica per rank; OWNS order[lo:hi].  Foreign UAVs are on hold (they only move when the exchange delivers their
    position), so a foreign UAV the rank was not sent stays where the last search saw it — outside the reach of every own UAV."""

    def __init__(self, sw, own, order):
        self.sw, self.own, self.order = sw, own, order
        foreign = np.setdiff1d(np.arange(len(order)), own)
        for i in foreign:
            sw.set_hold(int(i), 1, True)

    def step(self, dt):
        self.sw.step(dt)

    def records(self):
        st = self.sw.get_state()
        rec = np.zeros((len(self.own), 6))
        rec[:, :3] = st["x"][self.own]
        for k, i in enumerate(self.own):
            p = self.sw.get_params(int(i))
            rec[k, 3:] = (p.mass, p.arm_length, p.prop_radius)
        return rec

    def collide(self, sorted_pos_index, records, enabled, crash, rebounce):
        # sorted_pos_index: position in the slab-sorted order -> public index through `order`
        st = self.sw.get_state()
        pub = self.order[np.asarray(sorted_pos_index, dtype=np.int64)]
        x = st["x"].copy()
        x[pub] = records[:, :3]
        self.sw.set_state(0, len(x), x, st["v"], st["R"], st["omega"], st["motor_rpm"])
        self.sw.handle_collisions(enabled, crash, rebounce)


def _scenario(n_total):
    rng = np.random.default_rng(4321)
    side = (40.0 * n_total) ** (1.0 / 3.0)
    pos = rng.uniform(0, side, (n_total, 3)) + [0, 0, 20]
    pos[:30] = pos[30:60] + rng.normal(0, 0.3, (30, 3))
    vel = rng.normal(0, 7.0, (n_total, 3))  # lists go stale after ~25 ticks
    cmd = rng.uniform(0.4, 0.55, (n_total, 4))
    return pos, vel, cmd


def _export_worker(rank, world, port, n_total, n_ticks, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers
    from mrs_multirotor_simulator_amd.sharded import ExportSetSwarm, shard_range, slab_order
    from oracle import oracle_swarm as O
    pos, vel, cmd = _scenario(n_total)
    sw = O.OracleSwarm(n_total)
    sw.construct(0, n_total, helpers.oracle_params("x500"), pos, np.zeros(n_total))
    st = sw.get_state()
    sw.set_state(0, n_total, st["x"], vel, st["R"], st["omega"], st["motor_rpm"])
    sw.set_input(0, n_total, O.ACTUATOR_CMD, cmd)
    order = slab_order(pos, world)
    lo, hi = shard_range(n_total, world, rank)
    own = order[lo:hi]
    ex = ExportSetSwarm(n_total, OracleExportEngine(sw, own, order), torch.device("cpu"))
    for t in range(n_ticks):
        ex.tick(0.001, True, t == n_ticks // 2, 100.0)
    st = sw.get_state()
    np.savez(os.path.join(result_dir, f"rank{rank}.npz"), own=own, x=st["x"][own], v=st["v"][own], f=sw.get_external_force()[own],
             crashed=sw.has_crashed()[own], **{k: np.array(v) for k, v in ex.stats.items()}, cap=ex.cap)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_export_set_exchange_matches_single_process(tmp_path, oracle, world):
    import helpers
    import socket
    n_total, n_ticks = 403, 70
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_export_worker, args=(world, port, n_total, n_ticks, str(tmp_path)), nprocs=world, join=True)
    O = oracle
    pos, vel, cmd = _scenario(n_total)
    ref = O.OracleSwarm(n_total)
    ref.construct(0, n_total, helpers.oracle_params("x500"), pos, np.zeros(n_total))
    st = ref.get_state()
    ref.set_state(0, n_total, st["x"], vel, st["R"], st["omega"], st["motor_rpm"])
    ref.set_input(0, n_total, O.ACTUATOR_CMD, cmd)
    for t in range(n_ticks):
        ref.step(0.001)
        ref.handle_collisions(True, t == n_ticks // 2, 100.0)
    st, f, cr = ref.get_state(), ref.get_external_force(), ref.has_crashed()
    assert (np.abs(f).sum(axis=1) > 0).sum() > 10 and cr.sum() > 0  # the scenario really collides and crashes
    covered = np.zeros(n_total, dtype=bool)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        own = d["own"]
        covered[own] = True
        assert np.array_equal(d["x"], st["x"][own]) and np.array_equal(d["v"], st["v"][own]), f"rank {r} state"
        assert np.array_equal(d["f"], f[own]) and np.array_equal(d["crashed"], cr[own]), f"rank {r} forces / crash flags"
        # most ticks exchanged only the export sets, and those are a fraction of a shard
        assert int(d["ticks"]) == n_ticks and 2 <= int(d["searches"]) <= n_ticks // 4, (int(d["searches"]), n_ticks)
        assert int(d["cap"]) < len(own)
    assert covered.all()
