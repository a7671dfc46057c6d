#!/usr/bin/env python3
"""Cost of the multi-GPU collision tick per rank, measured with VIRTUAL SHARDS on one GPU: `world` swarms (x-sorted slabs of one
swarm of world * n UAVs at 64 m^3 per UAV), one host thread each, exchanging through the in-process loopback group — the code path
of the 8-GPU run with device-to-device copies in place of RCCL.  All shards share the one GPU, so (wall time per tick) / world is
the device time one rank spends per tick, collective excluded.  usage: export_tick_rate.py [n_per_shard] [world] [ticks] [exchange]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import mrs_multirotor_simulator_amd as M
from mrs_multirotor_simulator_amd.sharded import shard_range

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 300
exchange = {"export": M.EXCHANGE_EXPORT_SETS, "full": M.EXCHANGE_FULL_GATHER}[sys.argv[4] if len(sys.argv) > 4 else "export"]
slabs = (sys.argv[5] if len(sys.argv) > 5 else "slabs") == "slabs"
DT = 0.001
n_total = n * world
st, cmd = bench.make_inputs(n_total, "position+collisions", seed=5)
order = M.slab_partition(st["x"], world) if slabs else np.arange(n_total)
group = M.LoopbackGroup(world)
shards = []
for r in range(world):
    lo, hi = shard_range(n_total, world, r)
    idx = order[lo:hi]
    g = M.Swarm(hi - lo, arith=M.ARITH_FAST)
    g.construct(0, hi - lo, M.model_params("x500", ground_enabled=True))
    g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
    g.set_input(0, hi - lo, M.POSITION_CMD, cmd[idx])
    g.comm_init_loopback(group, r, n_total)
    g.set_exchange(exchange)
    shards.append(g)


def run(k):
    th = [threading.Thread(target=lambda g=g: g.tick_sharded_n(DT, k, True, False, 100.0)) for g in shards]
    for t in th:
        t.start()
    for t in th:
        t.join()


run(60)
t0 = time.perf_counter()
run(ticks)
el = time.perf_counter() - t0
ci = shards[0].comm_info()
print(f"{world} virtual shards x {n} UAVs ({'slabs' if slabs else 'index shards'}), {ci['parallelism']}: {el / ticks * 1e6:.1f} us wall per tick for all shards = "
      f"{el / ticks / world * 1e6:.1f} us of device time per rank and tick (collective excluded); rank 0: {ci}", flush=True)
th = [threading.Thread(target=lambda g=g: g.comm_destroy()) for g in shards]
[t.start() for t in th]
[t.join() for t in th]
