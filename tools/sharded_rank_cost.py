#!/usr/bin/env python3
"""Device time ONE rank of the sharded collision tick spends per tick, collective excluded: rank `rank` of `world` runs alone on the
GPU through mrs_swarm_comm_init_custom with a stand-in collective that delivers only the rank's own block (the other ranks' blocks
read as absent UAVs / empty export sets).  Kernels, launch pattern, batch protocol and buffer sizes are those of the real run; what
is missing is the other ranks' boundary UAVs (a few hundred records) and the wire time of the collective.
usage: sharded_rank_cost.py [n_per_shard] [world] [ticks] [export|full]"""
import ctypes as C
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import mrs_multirotor_simulator_amd as M
from mrs_multirotor_simulator_amd.sharded import shard_range

n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 400
exchange = {"export": M.EXCHANGE_EXPORT_SETS, "full": M.EXCHANGE_FULL_GATHER}[sys.argv[4] if len(sys.argv) > 4 else "export"]
rank = world // 2
DT = 0.001
n_total = n * world
st, cmd = bench.make_inputs(n_total, "position+collisions", seed=5)
order = M.slab_partition(st["x"], world)
lo, hi = shard_range(n_total, world, rank)
idx = order[lo:hi]
M.load_library()
import torch  # the HIP runtime the library is bound to
hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
REC_BYTES = 48 * ((n_total + world - 1) // world)


MAP_BYTES = 4 * ((n_total + world - 1) // world + 2)


def own_block_only(user, send, recv, nbytes, stream):
    # search ticks (records, slot maps): absent UAVs are NaN records, empty slot maps are zeros.  Ordinary ticks (export blocks): the
    # other ranks' blocks stay as the search left them — zeros, i.e. empty export sets — so the stand-in is ONE small copy.
    if nbytes in (REC_BYTES, MAP_BYTES) and hip.hipMemsetAsync(recv, 0xFF if nbytes == REC_BYTES else 0x00, nbytes * world, stream):
        return 1
    return hip.hipMemcpyAsync(recv + rank * nbytes, send, nbytes, 3, stream)  # hipMemcpyDeviceToDevice


g = M.Swarm(hi - lo, arith=M.ARITH_FAST)
g.construct(0, hi - lo, M.model_params("x500", ground_enabled=True))
g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
g.set_input(0, hi - lo, M.POSITION_CMD, cmd[idx])
g.comm_init_custom(world, rank, n_total, own_block_only)
g.set_exchange(exchange)
g.tick_sharded_n(DT, 60, True, False, 100.0)
g.synchronize()
t0 = time.perf_counter()
g.tick_sharded_n(DT, ticks, True, False, 100.0)
g.synchronize()
el = time.perf_counter() - t0
ci = g.comm_info()
print(f"rank {rank} of {world} alone, {hi - lo} UAVs of {n_total}: {el / ticks * 1e6:.1f} us per tick (collective stand-in: one device-to-device copy of the rank's own block, ~5 us of it); "
      f"{ci['searches']} searches in {ci['ticks']} ticks, {ci['noop_ticks']} ticks replayed; {ci['parallelism']}", flush=True)
g.comm_destroy()
