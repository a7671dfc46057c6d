#!/usr/bin/env python3
"""Two ranks of a sharded swarm in ONE process on the one GPU, bound by the peer-window exchange (or, for comparison, by the
in-process loopback group): what the exchange kernel itself costs (rocprofv3 --kernel-trace: k_peer_allgather) at the export-block
sizes of a real shard.  Both ranks' launches share the device, so the tick time printed here is two ranks' work, not one's.
(Not a supported deployment — ranks of one process belong on different devices: here a hipFree of one rank's host during a search waits
for the whole device, i.e. for the peer's exchange kernel, which waits for this rank; when that happens the exchange gives up after 10 s
and the run ends with the library's error.  Most runs get through, and the kernel trace is what this tool is for.)
usage: peer_rank_pair.py [n_per_shard] [ticks] [peer|loopback]"""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # a hardware queue per rank and stream: the ranks' kernels wait for each other on the device
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 400
transport = sys.argv[3] if len(sys.argv) > 3 else "peer"
import bench
import mrs_multirotor_simulator_amd as M
from mrs_multirotor_simulator_amd.sharded import shard_range

world, DT = 2, 0.001
n_total = n * world
st, cmd = bench.make_inputs(n_total, "position+collisions", seed=5)
order = M.slab_partition(st["x"], world)
ranks = []
group = M.LoopbackGroup(world) if transport == "loopback" else None
if group is not None:
    group.set_rendezvous(True)
for r in range(world):
    lo, hi = shard_range(n_total, world, r)
    idx = order[lo:hi]
    g = M.Swarm(hi - lo, arith=M.ARITH_FAST)
    g.construct(0, hi - lo, M.model_params("x500", ground_enabled=True))
    g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
    g.set_input(0, hi - lo, M.POSITION_CMD, cmd[idx])
    if group is not None:
        g.comm_init_loopback(group, r, n_total)
    ranks.append(g)
if group is None:
    windows = [g.peer_window_create(world, r, n_total, want_handle=False)[0] for r, g in enumerate(ranks)]
    for g in ranks:
        g.comm_init_peer(windows=windows)


def run(k):
    errs = []

    def one(g):
        try:
            g.tick_sharded_n(DT, k, True, False, 100.0)
            g.synchronize()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=one, args=(g,)) for g in ranks]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errs:
        raise errs[0]


run(80)
t0 = time.perf_counter()
run(ticks)
el = time.perf_counter() - t0
ci = ranks[0].comm_info()
print(f"{transport}: 2 ranks x {n} UAVs in one process on one GPU: {el / ticks * 1e6:.1f} us per tick (both ranks' work); export set {ci['export_count']} "
      f"(capacity {ci['export_capacity']}, {ci['bytes_per_tick']} bytes per rank and tick); {ci['searches']} searches in {ci['ticks']} ticks; "
      f"split ticks {[g.split_stats()[0] for g in ranks]}", flush=True)
for g in ranks:
    g.comm_destroy()
