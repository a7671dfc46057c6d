"""Device time of ONE neighbour search (pack + insert, list-building query) of a single-GPU swarm, by hipEvents around 16 searches back
to back (mrs_swarm_debug_search_ms): python tools/search_rate.py [n_uavs ...] [--volume V ...]   (run through gpurun)"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
import mrs_multirotor_simulator_amd as M  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("uavs", nargs="*", type=int, default=[100000])
ap.add_argument("--volume", nargs="*", type=float, default=[64.0])
a = ap.parse_args()
for n in a.uavs:
    for vol in a.volume:
        st, cmd = bench.make_inputs(n, "position+collisions", 1234, vol)
        g = M.Swarm(n, arith=M.ARITH_FAST)
        g.construct(0, n, M.model_params("x500", ground_enabled=True))
        g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        ms = [g.debug_search_ms(reps=16) for _ in range(3)]
        print(f"search of {n} UAVs at {vol:g} m^3 per UAV: {min(ms) * 1e3:.2f} us (three runs of 16: {', '.join(f'{m * 1e3:.2f}' for m in ms)})", flush=True)
        del g
