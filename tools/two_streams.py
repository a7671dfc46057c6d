#!/usr/bin/env python3
"""Experiment: does splitting the swarm over K concurrent HIP streams overlap the memory and compute phases of successive launches?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mrs_multirotor_simulator_amd import synthetic as helpers  # numpy-only state generators
import mrs_multirotor_simulator_amd as M
DT = 0.001
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = 1000
for K in (1, 2, 3, 4, 8):
    rng = np.random.default_rng(3)
    sws = []
    for k in range(K):
        n = N // K
        sw = M.Swarm(n, arith=M.ARITH_FAST)
        sw.construct(0, n, M.model_params("x500", ground_enabled=True))
        st = helpers.random_state(rng, n, 4)
        sw.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
        sw.set_input(0, n, M.ACTUATOR_CMD, rng.uniform(0.35, 0.6, (n, 4)))
        sws.append(sw)
    for rep in range(2):
        for sw in sws: sw.step_n(DT, 100)
        for sw in sws: sw.synchronize()
        t0 = time.perf_counter()
        # interleave the enqueue so that all streams stay fed
        chunk = 50
        for c in range(steps // chunk):
            for sw in sws: sw.step_n(DT, chunk)
        for sw in sws: sw.synchronize()
        el = time.perf_counter() - t0
    print(f"N {N} streams {K}: {el / steps * 1e6:7.2f} us per step of the whole swarm, {N * steps / el:.3e} UAV-steps/s", flush=True)
    del sws
