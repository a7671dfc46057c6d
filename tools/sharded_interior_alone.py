#!/usr/bin/env python3
"""What each part of a split sharded tick costs beside the others: one rank of 8 x 125 000 behind the stand-in
collective (bench.sharded_rank_cost) with parts of the split tick LEFT OUT through MRS_EXP_SPLIT_SKIP (bit 0 boundary launch, bit 1
collective, bit 2 interior launch) on a library built with -DMRS_WAIT_TICKS=0 (every in-kernel wait gives up at once): wrong results —
the call ends in MRS_ERR_HIP, which is caught —, right timing of launches that wait for nobody.
usage: MRS_SWARM_LIB=variants/libmrs_stepflag__DMRS_WAIT_TICKS_0ll.so MRS_EXP_SPLIT_SKIP=3 sharded_interior_alone.py [latency_us] [ticks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import mrs_multirotor_simulator_amd as M
from mrs_multirotor_simulator_amd.sharded import shard_range

lat = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 400
world, n_per = 8, 125_000
rank, n_total = world // 2, n_per * world
st, cmd = bench.make_inputs(n_total, "position+collisions", seed=5)
order = M.slab_partition(st["x"], world)
lo, hi = shard_range(n_total, world, rank)
idx = order[lo:hi]
width = float(st["x"][idx, 0].max() - st["x"][idx, 0].min()) * (1.0 + 1.0 / len(idx))
g = M.Swarm(hi - lo, arith=M.ARITH_FAST)
g.construct(0, hi - lo, M.model_params("x500", ground_enabled=True))
g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
g.set_input(0, hi - lo, M.POSITION_CMD, cmd[idx])
g.comm_init_standin(world, rank, n_total, lat, width)
err = ""
for phase, k in (("warm", 80), ("timed", ticks)):
    s0, _ = g.split_stats()
    t0 = time.perf_counter()
    try:
        g.tick_sharded_n(bench.DT, k, True, False, 100.0)
    except M.MrsError as e:  # the give-ups of the measurement build
        err = str(e)[:60]
    g.synchronize()
    el = time.perf_counter() - t0
s1, nb = g.split_stats()
ci = g.comm_info()
print(f"skip={os.environ.get('MRS_EXP_SPLIT_SKIP', '0')} nt={os.environ.get('MRS_INTERIOR_NT', '1')} latency {lat:g} us: {el / ticks * 1e6:.2f} us per tick; "
      f"{s1 - s0} of {ticks} ticks split, {nb} boundary blocks, searches {ci['searches']}" + (f"  [{err}]" if err else ""), flush=True)
