"""world_size-2/3 `gloo` tests (CPU) of the multi-GPU collision exchange.
(1) LAYOUT: shard ranges, the padded all-gather layout of the Python orchestrator (mrs_multirotor_simulator_amd/sharded.py) and that
shard-local collision results equal the single-process result — the local engine there is the CPU oracle (test infrastructure)
behind the same three calls the GPU engine implements, so that part does not exercise mrs_swarm_tick_sharded_n.
(2) PROTOCOL (round 5): the library's own host decisions (csrc/sharded_protocol.h, what export_ticks calls) driven from separate
processes bound by gloo collectives that carry the stall / warning words — all ranks must issue the same launches under real
asynchrony between hosts, with a negative control.  The device side of the protocol runs on the GPU box
(tests/test_sharded_multiprocess_gpu.py, test_peer_window_gpu.py, test_sharded_chaos_gpu.py); bench.py's N > 1 path end to end in
tests/test_bench_rehearsal_gpu.py."""
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


# ---- the library's own protocol decisions (csrc/sharded_protocol.h) across PROCESSES: every rank is a process with a host thread that
# issues launches (at most `lead` ahead of its device, on mirror words it reads at arbitrary moments, under random sleeps) and a device
# thread that runs them in order, each launch behind the gloo all-gather of the headers of the launch before — the role the export
# collective plays on the GPUs.  Reports (warning W, stall T) are raised by scripted launches on scripted ranks.  Whatever the
# interleaving: all ranks issue the SAME number of launches (else the collectives would not match up — here a gloo timeout, on
# hardware a hang), no launch after T steps on any rank, all ranks agree on T, W and the ticks that ran.  The arithmetic of ONE process
# with a random scheduler is tests/cpp/sharded_protocol_test.cpp; this is the same protocol with real asynchrony between hosts.
def _protocol_shim():
    import ctypes as C
    import subprocess
    so = os.path.join(ROOT, "tests", "cpp", "sharded_protocol_shim.so")
    src = os.path.join(ROOT, "tests", "cpp", "sharded_protocol_shim.cpp")
    hdr = os.path.join(ROOT, "mrs_multirotor_simulator_amd", "csrc", "sharded_protocol.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", src, "-o", so])
    L = C.CDLL(so)
    u = C.c_uint
    L.sp_host_is_behind.argtypes, L.sp_host_is_behind.restype = [u, u, u, C.c_int], C.c_int
    L.sp_search_ahead.argtypes, L.sp_search_ahead.restype = [u, C.c_int], u
    L.sp_segment_last.argtypes, L.sp_segment_last.restype = [u, u, u, u, u], u
    L.sp_ticks_ran.argtypes, L.sp_ticks_ran.restype = [u, u, u], u
    L.sp_search_due.argtypes, L.sp_search_due.restype = [u, u, C.c_int], C.c_int
    return L


def _min_nz(a, b):
    return b if a == 0 else (a if b == 0 else min(a, b))


def _protocol_worker(rank, world, port, n_segments, out_dir):
    import threading
    import time
    from datetime import timedelta
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
    L = _protocol_shim()
    script = np.random.default_rng(4242)           # the same scenario on every rank
    skew = np.random.default_rng(1000 + rank)      # this host's own pace
    results = []
    for seg in range(n_segments):
        lead = int(script.integers(1, 5))
        split = bool(script.integers(0, 2))
        vis = int(script.integers(3, 6)) if split else 1     # launches until a report is in every rank's words
        horizon = 4 if split else 0                            # announced stall indices in the split protocol
        n_ticks = int(script.integers(8, 40))
        reports = [(int(script.integers(0, world)), int(script.integers(1, n_ticks + 1)), bool(script.integers(0, 3) == 0)) for _ in range(int(script.integers(0, 4)))]
        ahead = L.sp_search_ahead(lead, int(split)) if not os.environ.get("MRS_TEST_BREAK_PROTOCOL") else 1  # (negative control: a search queued too soon after a warning)
        st = {"issued": 0, "last": n_ticks, "done": False, "mirT": 0, "mirW": 0, "mirP": 0, "hdrT": 0, "hdrW": 0, "error": None}
        gathered, ran = {}, {}
        lock = threading.Condition()

        def device():
            k = 0
            try:
                while True:
                    with lock:
                        while st["issued"] <= k and not st["done"]:
                            lock.wait(0.05)
                        if st["issued"] <= k:
                            return
                    k += 1
                    # launch k runs behind the collective of launch k - 1 (gathered[k - 1] exists: this thread made it)
                    T, W = st["hdrT"], st["hdrW"]
                    if k > vis:
                        for q in range(world):
                            T, W = _min_nz(T, gathered[k - vis][q][0]), _min_nz(W, gathered[k - vis][q][1])
                    with lock:
                        st["hdrT"], st["hdrW"] = T, W
                        if T:
                            st["mirT"] = T
                        if W:
                            st["mirW"] = W
                        ran[k] = not (T != 0 and k > T)
                        if ran[k]:
                            st["mirP"] = k
                        lock.notify_all()
                    time.sleep(float(skew.uniform(0, 3e-4)))
                    if ran[k]:
                        for (r, launch, stall) in reports:
                            if r == rank and launch == k:
                                with lock:
                                    if stall:
                                        st["hdrT"] = _min_nz(st["hdrT"], k + horizon)
                                        if horizon == 0:
                                            st["mirT"] = _min_nz(st["mirT"], k + horizon)
                                    else:
                                        st["hdrW"] = _min_nz(st["hdrW"], k)
                                    lock.notify_all()
                    mine = torch.tensor([st["hdrT"], st["hdrW"]], dtype=torch.int64)
                    everyone = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
                    dist.all_gather(everyone, mine)   # the export collective of tick k: carries every rank's header words
                    gathered[k] = [(int(e[0]), int(e[1])) for e in everyone]
            except Exception as e:  # noqa: BLE001 - a mismatched collective (ranks disagree on the launch count) ends here
                st["error"] = f"{type(e).__name__}: {e}"

        dev = threading.Thread(target=device)
        dev.start()
        while True:  # the host of export_ticks: wait_for_progress, segment_last, launch
            nxt = st["issued"] + 1
            if nxt > st["last"]:
                break
            with lock:
                while L.sp_host_is_behind(nxt, st["mirP"], st["mirT"], lead) and st["error"] is None:
                    lock.wait(0.02)
                T, W = st["mirT"], st["mirW"]
            if st["error"]:
                break
            time.sleep(float(skew.uniform(0, 4e-4)) if skew.integers(0, 3) == 0 else 0.0)  # (decides on what it read a moment ago)
            st["last"] = L.sp_segment_last(st["last"], T, W, lead, ahead)
            if nxt > st["last"]:
                break
            with lock:
                st["issued"] = nxt
                lock.notify_all()
        with lock:
            st["done"] = True
            lock.notify_all()
        dev.join(90)
        assert not dev.is_alive() and st["error"] is None, (seg, st["error"])
        # the final fold of the segment: every rank ends with the same words
        mine = torch.tensor([st["hdrT"], st["hdrW"], st["issued"]], dtype=torch.int64)
        everyone = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(everyone, mine)
        T = W = 0
        for e in everyone:
            T, W = _min_nz(T, int(e[0])), _min_nz(W, int(e[1]))
        issued = [int(e[2]) for e in everyone]
        assert len(set(issued)) == 1, (seg, "ranks issued different numbers of launches", issued)
        launched = issued[0]
        for k in range(1, launched + 1):
            assert ran[k] == (not (T != 0 and k > T)), (seg, rank, k, ran[k], T)
        t_ran = L.sp_ticks_ran(T, 1, launched)
        assert t_ran == (T if (T != 0 and T <= launched) else launched)
        if W != 0 and T == 0:
            assert launched == min(W + ahead - 1, n_ticks), (seg, W, ahead, launched)
        if T != 0:
            assert launched <= T + lead + 1, (seg, T, lead, launched)
        results.append((launched, T, W, int(L.sp_search_due(T, W, int(launched < n_ticks)))))
    np.save(os.path.join(out_dir, f"protocol_rank{rank}.npy"), np.array(results))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_protocol_decisions_across_processes(tmp_path, world):
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    n_segments = 40
    mp.spawn(_protocol_worker, args=(world, port, n_segments, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(os.path.join(str(tmp_path), f"protocol_rank{r}.npy")) for r in range(world)]
    for r in range(1, world):
        assert np.array_equal(res[r], res[0])            # every rank: the same launches, words and decision in every segment
    assert (res[0][:, 1] != 0).sum() >= 3 and (res[0][:, 2] != 0).sum() >= 3 and (res[0][:, 3] == 1).sum() >= 5  # stalls, warnings, searches happened


def test_protocol_negative_control_is_caught(tmp_path, monkeypatch):
    """the same model with a search queued ONE launch after a warning instead of lead + 3 / lead + 6: hosts that have not seen the
    warning yet issue more launches than the others — the collectives stop matching up and the run fails (on hardware: a hang)"""
    import socket
    monkeypatch.setenv("MRS_TEST_BREAK_PROTOCOL", "1")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    with pytest.raises(Exception):
        mp.spawn(_protocol_worker, args=(2, port, 40, str(tmp_path)), nprocs=2, join=True)
