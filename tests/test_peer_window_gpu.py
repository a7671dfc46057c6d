"""The peer-window exchange (include/mrs_swarm.h: mrs_swarm_peer_window_create / mrs_swarm_comm_init_peer; csrc/collide.hip
k_peer_allgather): the collectives of the sharded tick as direct device-to-device writes with device-side signalling.
  * ranks in ONE process (pointers), each on its own host thread, hosts skewed, split ticks — in a child process, because a hardware
    queue per rank has to be asked for before the HIP runtime starts (kernels of different ranks wait for each other on the device;
    two of them behind each other in one queue would wait for ever, i.e. for the 10 s after which the exchange gives up);
  * ranks in SEPARATE processes on the one GPU (IPC handles carried over gloo): separate address spaces, windows mapped with
    hipIpcOpenMemHandle — the form a multi-GPU node runs, one hop shorter.
Both against the single-swarm oracle, UAV by UAV."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
DT = 0.001


@pytest.mark.parametrize("world,n_total,chaos_us", [(2, 3001, 0), (4, 5000, 300)])
def test_peer_window_ranks_of_one_process(world, n_total, chaos_us):
    env = dict(os.environ, GPU_MAX_HW_QUEUES=str(4 * world))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "peer_window_worker.py"), str(world), str(n_total), str(chaos_us)], env=env,
                       capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-3000:])
    assert r.returncode == 0 and "PEER-WINDOW OK" in r.stdout


BLOCKS = [(60, False), (1, True), (59, False)]


def _ipc_worker(rank, world, port, n_total, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers
    import mrs_multirotor_simulator_amd as M
    from mrs_multirotor_simulator_amd.sharded import shard_range
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
    _, handle = g.peer_window_create(world, rank, n_total)
    handles = [None] * world
    dist.all_gather_object(handles, handle)  # the only thing the hosts ever tell each other
    g.comm_init_peer(handles=handles)
    for n, crash in BLOCKS:
        g.tick_sharded_n(DT, n, True, crash, 100.0)
    s = g.get_state()
    ci = g.comm_info()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx, x=s["x"], v=s["v"], R=s["R"], omega=s["omega"], motor_rpm=s["motor_rpm"],
             f=g.get_external_force(), crashed=g.has_crashed(), searches=ci["searches"], ticks=ci["ticks"])
    dist.barrier()  # nobody unmaps a window a peer may still write into
    g.comm_destroy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_peer_window_ranks_in_separate_processes(tmp_path, oracle, world):
    import torch.multiprocessing as mp
    import helpers
    from helpers import RTOL_LITERAL
    from test_sharded_multiprocess_gpu import _scenario
    n_total = 3001
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_ipc_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
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
    covered = np.zeros(n_total, dtype=bool)
    for r in range(world):
        d = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        idx = d["idx"]
        covered[idx] = True
        assert np.array_equal(d["crashed"], co[idx]), f"rank {r}: crash flags"
        helpers.assert_close(d["f"], fo[idx], 1e-11, f"rank {r}: forces")
        for k in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(d[k], so[k][idx], RTOL_LITERAL, f"rank {r}: {k}")
        assert int(d["ticks"]) == n_ticks and 2 <= int(d["searches"]) <= n_ticks // 3
    assert covered.all()
