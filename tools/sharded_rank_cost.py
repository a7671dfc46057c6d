#!/usr/bin/env python3
"""Device time ONE rank of the sharded collision tick spends per tick (bench.sharded_rank_cost: rank world/2 of `world` alone on the
GPU, the library's stand-in collective of fixed latency, periodic-image neighbours).
usage: sharded_rank_cost.py [n_per_shard] [world] [ticks] [latency_us] [split|serial]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 125_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 400
latency = float(sys.argv[4]) if len(sys.argv) > 4 else 20.0
split = not (len(sys.argv) > 5 and sys.argv[5] == "serial")
r = bench.sharded_rank_cost(n, world, ticks, latency, split)
print(f"rank {r['rank']} of {world} alone, {r['uavs_per_rank']} UAVs of {n * world}, stand-in collective of {latency:g} us: {r['us_per_tick']:.1f} us per tick; "
      f"{r['split_ticks']} of {ticks} ticks in the split form, {r['boundary_blocks']} boundary blocks of {r['blocks']}, export set {r['export_set']} "
      f"(capacity {r['export_capacity']}); {r['searches']} searches in {ticks} ticks, {r['replayed_noop_ticks']} ticks replayed", flush=True)
