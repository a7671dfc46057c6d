#!/usr/bin/env python3
"""Step rate of heterogeneous fleets: airframe types grouped in runs of `run` consecutive UAVs (run >= 64 and a multiple of 64 keeps
every 64-UAV block single-type; smaller runs make mixed blocks, which take the per-lane-constants kernel)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mrs_multirotor_simulator_amd import synthetic as helpers  # numpy-only state generators
import mrs_multirotor_simulator_amd as M
n = 100_000
rng = np.random.default_rng(1)
for run in (n, 6400, 64, 16, 1):
    sw = M.Swarm(n, arith=M.ARITH_FAST)
    frames = ["x500", "f550", "t650"]
    i = 0; k = 0
    while i < n:
        c = min(run, n - i)
        sw.construct(i, c, M.model_params(frames[k % 3], ground_enabled=True, ground_z=0.0))
        i += c; k += 1
        if run < 64 and i >= 6400:  # keep the set-up time sane: the rest of the swarm repeats the pattern through set_params below
            break
    if run < 64:
        # replicate the per-UAV pattern of the first 6400 UAVs over the whole swarm with few calls: same types, interleaved
        for f in range(3):
            p = M.model_params(frames[f], ground_enabled=True, ground_z=0.0)
            idx = [j for j in range(6400, n) if (j // run) % 3 == f]
            # contiguous runs
            start = None
            for j in idx + [None]:
                if start is None: start = prev = j; continue
                if j is not None and j == prev + 1: prev = j; continue
                sw.set_params(start, prev - start + 1, p); start = prev = j
    st = helpers.random_state(rng, n, 4)
    sw.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    sw.set_input(0, n, M.ACTUATOR_CMD, rng.uniform(0.35, 0.6, (n, 8)))
    sw.step_n(0.001, 50); sw.synchronize()
    t0 = time.perf_counter(); sw.step_n(0.001, 300); sw.synchronize(); el = time.perf_counter() - t0
    print(f"runs of {run:6d} UAVs per airframe: {el / 300 * 1e6:8.2f} us per step  ({n * 300 / el:.3e} UAV-steps/s)", flush=True)
    del sw
