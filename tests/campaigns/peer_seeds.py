#!/usr/bin/env python3
"""More seeds of tests/test_peer_window_gpu.py: the peer-window exchange with ranks in separate processes on the one GPU, `runs`
scenarios with random world size (2..5), swarm size, speeds, chaos amplitude, shard shape and call lengths, against the oracle.
usage: peer_seeds.py [runs] [first_seed]"""
import os
import socket
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DT = 0.001


def scenario(seed):
    rng = np.random.default_rng(80_000 + seed)
    world = int(rng.integers(2, 6))
    n_total = int(rng.integers(500, 2500)) * world + int(rng.integers(0, world))
    p = dict(world=world, n_total=n_total, speed=float(rng.uniform(3.0, 9.0)), chaos=int(rng.choice([0, 100, 300, 1000])),
             slabs=bool(rng.random() < 0.8), calls=[int(c) for c in rng.integers(1, 80, 4)], crash_call=int(rng.integers(0, 4)))
    import helpers
    side = (64.0 * n_total) ** (1.0 / 3.0)
    pos = rng.uniform(0, side, (n_total, 3)) + [0, 0, 30]
    k = n_total // 15
    pos[:k] = pos[k:2 * k] + rng.normal(0, 0.3, (k, 3))
    st = helpers.random_state(rng, n_total, 4, tilted=True)
    st["x"] = pos
    st["v"] = rng.normal(0, p["speed"], (n_total, 3))
    cmd = rng.uniform(0.4, 0.55, (n_total, 4))
    return p, pos, st, cmd


def worker(rank, world, port, seed, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MRS_SHARD_SPLIT_MIN_BLOCKS="1", MRS_SHARD_SPLIT_MAX_FRACTION="0.95")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import helpers
    import mrs_multirotor_simulator_amd as M
    from mrs_multirotor_simulator_amd.sharded import bind_native_exchange, shard_range
    M.load_library()
    p, pos, st, cmd = scenario(seed)
    n_total = p["n_total"]
    order = M.slab_partition(pos, world) if p["slabs"] else np.arange(n_total)
    lo, hi = shard_range(n_total, world, rank)
    idx = order[lo:hi]
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    g = M.Swarm(hi - lo, arith=M.ARITH_LITERAL)
    g.construct(0, hi - lo, helpers.to_product_params(M, po), pos[idx], np.zeros(hi - lo))
    g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
    g.set_input(0, hi - lo, M.ACTUATOR_CMD, cmd[idx])
    bind_native_exchange(g, n_total, "peer")
    if p["chaos"]:
        g.debug_chaos(p["chaos"], seed=1000 * seed + rank)
    for k, n in enumerate(p["calls"]):
        g.tick_sharded_n(DT, n, True, k == p["crash_call"], 100.0)
    s = g.get_state()
    ci = g.comm_info()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), idx=idx, x=s["x"], v=s["v"], R=s["R"], omega=s["omega"], motor_rpm=s["motor_rpm"],
             f=g.get_external_force(), crashed=g.has_crashed(), searches=ci["searches"], split=g.split_stats()[0])
    dist.barrier()
    g.comm_destroy()
    dist.barrier()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    import helpers
    from helpers import RTOL_LITERAL
    from oracle import oracle_swarm as oracle
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    for seed in range(first, first + runs):
        p, pos, st, cmd = scenario(seed)
        world, n_total = p["world"], p["n_total"]
        t0 = time.time()
        with tempfile.TemporaryDirectory() as tmp, socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
            sk.close()
            mp.spawn(worker, args=(world, port, seed, tmp), nprocs=world, join=True)
            o = oracle.OracleSwarm(n_total)
            o.construct(0, n_total, helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0), pos, np.zeros(n_total))
            o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
            o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
            for k, n in enumerate(p["calls"]):
                for _ in range(n):
                    o.step(DT)
                    o.handle_collisions(True, k == p["crash_call"], 100.0)
            so, fo, co = o.get_state(), o.get_external_force(), o.has_crashed()
            searches, split = [], []
            for r in range(world):
                d = np.load(os.path.join(tmp, f"rank{r}.npz"))
                idx = d["idx"]
                assert np.array_equal(d["crashed"], co[idx]), f"seed {seed} rank {r}: crash flags"
                helpers.assert_close(d["f"], fo[idx], 1e-11, f"seed {seed} rank {r}: forces")
                for key in ("x", "v", "R", "omega", "motor_rpm"):
                    helpers.assert_close(d[key], so[key][idx], RTOL_LITERAL, f"seed {seed} rank {r}: {key}")
                searches.append(int(d["searches"]))
                split.append(int(d["split"]))
        print(f"seed {seed}: {world} processes, {n_total} UAVs, {'slabs' if p['slabs'] else 'index shards'}, speed {p['speed']:.1f}, chaos {p['chaos']} us, "
              f"calls {p['calls']} (crash mode in call {p['crash_call']}): searches {searches}, split ticks {split}, {time.time() - t0:.1f} s", flush=True)
    print(f"{runs} scenarios OK")


if __name__ == "__main__":
    main()
