#!/usr/bin/env python3
"""Device time ONE rank of the sharded collision tick spends per tick: rank world/2 of `world` runs alone on the GPU with the library's
measurement stand-in for the collective (mrs_swarm_comm_init_standin): every collective takes `latency` us of stream time and the
rank's neighbours in the slab order are periodic images of itself — so the boundary sets, the boundary / interior launches, the
searches and the buffer sizes are those of the real run; what is missing is the other ranks' physics and the wire.
usage: sharded_rank_cost.py [n_per_shard] [world] [ticks] [latency_us] [split|serial]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 400
latency = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
if len(sys.argv) > 5 and sys.argv[5] == "serial":
    os.environ["MRS_SHARD_SPLIT"] = "0"
import bench
import mrs_multirotor_simulator_amd as M
from mrs_multirotor_simulator_amd.sharded import shard_range

rank = world // 2
DT = 0.001
n_total = n * world
st, cmd = bench.make_inputs(n_total, "position+collisions", seed=5)
order = M.slab_partition(st["x"], world)
lo, hi = shard_range(n_total, world, rank)
idx = order[lo:hi]
width = float(st["x"][idx, 0].max() - st["x"][idx, 0].min()) * (1.0 + 1.0 / len(idx))
g = M.Swarm(hi - lo, arith=M.ARITH_FAST)
g.construct(0, hi - lo, M.model_params("x500", ground_enabled=True))
g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
g.set_input(0, hi - lo, M.POSITION_CMD, cmd[idx])
g.comm_init_standin(world, rank, n_total, latency, width)
g.tick_sharded_n(DT, 80, True, False, 100.0)
g.synchronize()
s0, _ = g.split_stats()
t0 = time.perf_counter()
g.tick_sharded_n(DT, ticks, True, False, 100.0)
g.synchronize()
el = time.perf_counter() - t0
ci = g.comm_info()
s1, nbnd = g.split_stats()
print(f"rank {rank} of {world} alone, {hi - lo} UAVs of {n_total}, stand-in collective of {latency:g} us: {el / ticks * 1e6:.1f} us per tick; "
      f"{s1 - s0} of {ticks} ticks in the split form, {nbnd} boundary blocks of {(hi - lo + 63) // 64}, export set {ci['export_count']} (capacity {ci['export_capacity']}); "
      f"{ci['searches']} searches in {ci['ticks']} ticks, {ci['noop_ticks']} ticks replayed", flush=True)
g.comm_destroy()
