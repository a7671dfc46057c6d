#!/usr/bin/env python3
"""Cost of the multi-GPU collision path per rank, measured with virtual shards on ONE GPU: `world` swarms of n UAVs each pack their
records into a shared gathered buffer (what the RCCL all-gather would deliver) and collide against it.  Reports the time one
shard spends per tick in step + pack + collide, with and without neighbour lists.  usage: gathered_tick_rate.py [n_per_shard] [world]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from mrs_multirotor_simulator_amd import synthetic as helpers  # numpy-only state generators
import mrs_multirotor_simulator_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
DT = 0.001
for lists in ("1", "0"):
    os.environ["MRS_NEIGHBOUR_LISTS"] = lists
    rng = np.random.default_rng(5)
    n_total = n * world
    side = (64.0 * n_total) ** (1.0 / 3.0)
    recv = torch.full((n_total, 6), float("nan"), dtype=torch.float64, device="cuda")
    shards = []
    for r in range(world):
        st = helpers.random_state(rng, n, 4, tilted=True)
        st["x"] = rng.uniform(0, 1, (n, 3)) * [side * 2, side * 2, side / 4] + [0, 0, 5]
        g = M.Swarm(n, arith=M.ARITH_FAST)
        g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), st["x"], np.zeros(n))
        g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        g.set_input(0, n, M.POSITION_CMD, np.concatenate([st["x"] + rng.uniform(-5, 5, (n, 3)), rng.uniform(-3, 3, (n, 1))], axis=1))
        shards.append(g)
    def tick():
        for r, g in enumerate(shards):
            g.step(DT)
            g.pack_positions_to(recv[r * n:].data_ptr())
        for g in shards:
            g.synchronize()
        for r, g in enumerate(shards):
            g.handle_collisions_gathered(recv.data_ptr(), n_total, r * n, True, False, 100.0)
    for _ in range(30):
        tick()
    for g in shards:
        g.synchronize()
    # time shard 0's own work only: the other shards just refresh their part of the gathered buffer
    T = 200
    t_acc = 0.0
    for _ in range(T):
        for r, g in enumerate(shards):
            if r:
                g.step(DT); g.pack_positions_to(recv[r * n:].data_ptr()); g.synchronize()
        g0 = shards[0]
        t0 = time.perf_counter()
        g0.step(DT); g0.pack_positions_to(recv.data_ptr())
        g0.handle_collisions_gathered(recv.data_ptr(), n_total, 0, True, False, 100.0)
        g0.synchronize()
        t_acc += time.perf_counter() - t0
        for r, g in enumerate(shards):
            if r:
                g.handle_collisions_gathered(recv.data_ptr(), n_total, r * n, True, False, 100.0); g.synchronize()
    print(f"lists={lists}: {n} UAVs per shard, {world} shards ({n_total} records): {t_acc / T * 1e6:8.1f} us per tick of one shard "
          f"(step + pack + collide, host-synchronised), searches {shards[0].collision_stats()}", flush=True)
    del shards
