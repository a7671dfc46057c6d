"""The sharded collision tick across PROCESSES: `world` worker processes on the one GPU of the test box, one swarm each, bound through
mrs_swarm_comm_init_custom to an all-gather that crosses the process boundary (device -> host -> gloo all_gather -> device).  RCCL
refuses two ranks on one device, so this is the closest a one-GPU box gets to the 8-process run: separate address spaces, separate
HIP contexts and pinned control words, the C ABI's own protocol (export sets, stall / warning words in the headers, searches) on
every rank, and nothing shared but the collective.  Results must equal the single-swarm oracle, UAV by UAV."""
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
BLOCKS = [(60, False), (1, True), (59, False)]  # ticks, crash mode


def _scenario(n_total):
    import helpers
    rng = np.random.default_rng(77)
    side = (64.0 * n_total) ** (1.0 / 3.0)
    pos = rng.uniform(0, side, (n_total, 3)) + [0, 0, 30]
    k = n_total // 15
    pos[:k] = pos[k:2 * k] + rng.normal(0, 0.3, (k, 3))
    st = helpers.random_state(rng, n_total, 4, tilted=True)
    st["x"] = pos
    st["v"] = rng.normal(0, 6.0, (n_total, 3))  # fast enough for several searches in 120 ticks
    cmd = rng.uniform(0.4, 0.55, (n_total, 4))
    return pos, st, cmd


def _worker(rank, world, port, n_total, exchange_name, out_dir):
    import ctypes as C
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers
    import mrs_multirotor_simulator_amd as M
    from mrs_multirotor_simulator_amd.sharded import shard_range
    M.load_library()
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    calls = [0]

    def allgather(user, send, recv, nbytes, stream):  # blocking: the launches queued so far finish, then the bytes cross the processes
        calls[0] += 1
        if hip.hipStreamSynchronize(stream):
            return 1
        mine = np.empty(nbytes, dtype=np.uint8)
        if hip.hipMemcpy(mine.ctypes.data, send, nbytes, 2):  # device -> host
            return 1
        everyone = torch.empty(world * nbytes, dtype=torch.uint8)
        dist.all_gather_into_tensor(everyone, torch.from_numpy(mine))
        return hip.hipMemcpy(recv, everyone.numpy().ctypes.data, world * nbytes, 1)  # host -> device

    pos, st, cmd = _scenario(n_total)
    order = M.slab_partition(pos, world)
    lo, hi = shard_range(n_total, world, rank)
    idx = order[lo:hi]
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    g = M.Swarm(hi - lo, arith=M.ARITH_LITERAL)
    g.construct(0, hi - lo, helpers.to_product_params(M, po), pos[idx], np.zeros(hi - lo))
    g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
    g.set_input(0, hi - lo, M.ACTUATOR_CMD, cmd[idx])
    g.comm_init_custom(world, rank, n_total, allgather)
    g.set_exchange(M.EXCHANGE_EXPORT_SETS if exchange_name == "export" else M.EXCHANGE_FULL_GATHER)
    for n, crash in BLOCKS:
        g.tick_sharded_n(DT, n, True, crash, 100.0)
    s = g.get_state()
    ci = g.comm_info()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx, x=s["x"], v=s["v"], R=s["R"], omega=s["omega"], motor_rpm=s["motor_rpm"],
             f=g.get_external_force(), crashed=g.has_crashed(), searches=ci["searches"], ticks=ci["ticks"], calls=calls[0],
             bytes_per_tick=ci["bytes_per_tick"], bytes_per_rebuild=ci["bytes_per_rebuild"])
    g.comm_destroy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange", [(2, "export"), (3, "export"), (2, "full")])
def test_sharded_ticks_across_processes_match_the_oracle(tmp_path, oracle, world, exchange):
    import torch.multiprocessing as mp
    import helpers
    from helpers import RTOL_LITERAL
    n_total = 3001
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, n_total, exchange, str(tmp_path)), nprocs=world, join=True)  # fresh interpreters: each one opens the GPU itself
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
    assert co.sum() > 0 and (np.abs(fo).sum(axis=1) > 0).sum() > 20  # the scenario collides and crashes
    covered = np.zeros(n_total, dtype=bool)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        idx = d["idx"]
        covered[idx] = True
        assert np.array_equal(d["crashed"], co[idx]), f"rank {r}: crash flags"
        helpers.assert_close(d["f"], fo[idx], 1e-11, f"rank {r}: forces")
        for k in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(d[k], so[k][idx], RTOL_LITERAL, f"rank {r}: {k}")
        assert int(d["ticks"]) == n_ticks
        if exchange == "export":
            assert 2 <= int(d["searches"]) <= n_ticks // 3, int(d["searches"])          # most ticks exchanged the export sets only
            assert int(d["bytes_per_tick"]) < int(d["bytes_per_rebuild"]) // 2
        assert int(d["calls"]) >= n_ticks
    assert covered.all()
