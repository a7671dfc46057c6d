"""The peer-window exchange (include/mrs_swarm.h: mrs_swarm_peer_window_create / mrs_swarm_comm_init_peer; csrc/collide.hip
k_peer_allgather): the collectives of the sharded tick as direct device-to-device writes with device-side signalling, no collective
library and no host in the tick.  Ranks in SEPARATE processes on the one GPU of the test box — the form a multi-GPU node runs, one
xGMI hop shorter: separate address spaces and HIP contexts, windows mapped with hipIpcOpenMemHandle, the 64-byte handles the only
thing the hosts ever tell each other (over gloo).  Hosts skewed by mrs_swarm_debug_chaos (random sleeps, stale word views — nothing
couples the hosts any more, so they drift as far as the protocol lets them), ticks in the split form, calls of uneven length, one
tick in crash mode; every rank against the single-swarm oracle, UAV by UAV.

(Ranks of ONE process on ONE device are not a test bed for this backend: the kernels of different ranks wait for each other on the
device, and any runtime call of one rank's host that waits for the whole device — hipFree in a search that grows a buffer — then
waits for a peer's kernel that waits for this rank: 10 s later the exchange gives up.  Measured with tools/peer_rank_pair.py.)"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
DT = 0.001
BLOCKS = [(41, False), (1, True), (118, False), (80, False)]  # ticks, crash mode: 240 ticks


def _ipc_worker(rank, world, port, n_total, shards, chaos_us, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ["MRS_SHARD_SPLIT_MIN_BLOCKS"] = "1"      # (read when a swarm is created: small shards take the split form too)
    os.environ["MRS_SHARD_SPLIT_MAX_FRACTION"] = "0.95"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers
    import mrs_multirotor_simulator_amd as M
    from mrs_multirotor_simulator_amd.sharded import shard_range
    from test_sharded_multiprocess_gpu import _scenario
    M.load_library()
    pos, st, cmd = _scenario(n_total)
    order = M.slab_partition(pos, world) if shards == "slabs" else np.arange(n_total)
    lo, hi = shard_range(n_total, world, rank)
    idx = order[lo:hi]
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    g = M.Swarm(hi - lo, arith=M.ARITH_LITERAL)
    g.construct(0, hi - lo, helpers.to_product_params(M, po), pos[idx], np.zeros(hi - lo))
    g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
    g.set_input(0, hi - lo, M.ACTUATOR_CMD, cmd[idx])
    from mrs_multirotor_simulator_amd.sharded import bind_native_exchange
    bind_native_exchange(g, n_total, "peer")  # window + IPC handle, all-gather of the handles over gloo: the only thing the hosts ever tell each other
    if chaos_us > 0:
        g.debug_chaos(chaos_us, seed=31 * world + rank)
    for n, crash in BLOCKS:
        g.tick_sharded_n(DT, n, True, crash, 100.0)
    s = g.get_state()
    ci = g.comm_info()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx, x=s["x"], v=s["v"], R=s["R"], omega=s["omega"], motor_rpm=s["motor_rpm"],
             f=g.get_external_force(), crashed=g.has_crashed(), searches=ci["searches"], ticks=ci["ticks"], split=g.split_stats()[0],
             bytes_per_tick=ci["bytes_per_tick"], bytes_per_rebuild=ci["bytes_per_rebuild"])
    dist.barrier()  # nobody unmaps a window a peer may still write into
    g.comm_destroy()
    dist.barrier()
    dist.destroy_process_group()


# (12 000 UAVs on 2 ranks: the search's full gather is 288 KB per rank — several blocks per peer and the ticket path of the exchange kernel)
@pytest.mark.parametrize("world,shards,chaos_us,n_total", [(2, "slabs", 0, 3001), (3, "slabs", 300, 3001), (4, "slabs", 300, 3001), (3, "index", 100, 1800),
                                                          (2, "slabs", 100, 12000)])
def test_peer_window_ranks_in_separate_processes(tmp_path, oracle, world, shards, chaos_us, n_total):
    import torch.multiprocessing as mp
    import helpers
    from helpers import RTOL_LITERAL
    from test_sharded_multiprocess_gpu import _scenario
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_ipc_worker, args=(world, port, n_total, shards, chaos_us, str(tmp_path)), nprocs=world, join=True)
    pos, st, cmd = _scenario(n_total)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0), pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    n_ticks = 0
    for n, crash in BLOCKS:
        for _ in range(n):
            o.step(DT)
            o.handle_collisions(True, crash, 100.0)
        n_ticks += n
    so, fo, co = o.get_state(), o.get_external_force(), o.has_crashed()
    assert co.sum() > 0 and (np.abs(fo).sum(axis=1) > 0).sum() > 20
    covered, split = np.zeros(n_total, dtype=bool), []
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        idx = d["idx"]
        covered[idx] = True
        assert np.array_equal(d["crashed"], co[idx]), f"rank {r}: crash flags"
        helpers.assert_close(d["f"], fo[idx], 1e-11, f"rank {r}: forces")
        for k in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(d[k], so[k][idx], RTOL_LITERAL, f"rank {r}: {k}")
        assert int(d["ticks"]) == n_ticks and 2 <= int(d["searches"]) <= n_ticks // 2, (int(d["ticks"]), int(d["searches"]))
        split.append(int(d["split"]))
    assert covered.all()
    if shards == "slabs":
        assert sum(split) > 20 * world, split  # the split form really ran (the 4 ticks after every search and call are serial)
    print(f"peer windows, {world} processes, {shards}, chaos {chaos_us} us: split ticks {split}")


def _lost_peer_worker(rank, world, port, n_total, out_dir):
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers
    import mrs_multirotor_simulator_amd as M
    from mrs_multirotor_simulator_amd.sharded import bind_native_exchange, shard_range
    from test_sharded_multiprocess_gpu import _scenario
    M.load_library()
    pos, st, cmd = _scenario(n_total)
    order = M.slab_partition(pos, world)
    lo, hi = shard_range(n_total, world, rank)
    idx = order[lo:hi]
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    g = M.Swarm(hi - lo, arith=M.ARITH_LITERAL)
    g.construct(0, hi - lo, helpers.to_product_params(M, po), pos[idx], np.zeros(hi - lo))
    g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
    g.set_input(0, hi - lo, M.ACTUATOR_CMD, cmd[idx])
    bind_native_exchange(g, n_total, "peer")
    g.tick_sharded_n(DT, 30, True, False, 100.0)
    msg, took = "", 0.0
    if rank == 0:  # rank 1 stops ticking (a lost rank): this one must come back with an error, not hang, and not 10 s per queued launch
        t0 = time.time()
        try:
            g.tick_sharded_n(DT, 200, True, False, 100.0)
        except M.MrsError as e:
            msg = str(e)
        took = time.time() - t0
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{took:.1f}\n{msg}\n")
    dist.barrier()  # rank 1 keeps its window mapped until rank 0 is done with it
    g.comm_destroy()
    dist.barrier()
    dist.destroy_process_group()


def test_a_lost_peer_is_an_error_not_a_hang(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_lost_peer_worker, args=(2, port, 3001, str(tmp_path)), nprocs=2, join=True)
    took, msg = open(os.path.join(str(tmp_path), "rank0.txt")).read().split("\n")[:2]
    assert "waited in vain for the block of rank 1" in msg, msg
    assert 9.0 < float(took) < 45.0, took  # one wait of 10 s on the device (and the host's patience), not one per launch
