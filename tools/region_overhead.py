"""What a SHORT timed region costs beyond its device time: K steps of the headline workload, wall clock from before step_n to after
synchronize, against the same region's device time by hipEvents — and where the difference goes (enqueue time of the call, the wait).
usage: region_overhead.py [steps ...]   (run through gpurun; environment switches of the runtime can be tried from outside)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
import mrs_multirotor_simulator_amd as M  # noqa: E402

n = 100_000
st, cmd = bench.make_inputs(n, "actuator", 3)
g = M.Swarm(n, arith=M.ARITH_FAST)
g.construct(0, n, M.model_params("x500", ground_enabled=True))
g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
g.set_input(0, n, M.ACTUATOR_CMD, cmd)
g.step_n(0.001, 200)
g.synchronize()
for steps in [int(a) for a in sys.argv[1:]] or [20, 100, 1000]:
    rows = []
    for rep in range(12):
        g.synchronize()
        t0 = time.perf_counter()
        g.step_n(0.001, steps)
        t1 = time.perf_counter()
        g.synchronize()
        t2 = time.perf_counter()
        rows.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t2 - t0) * 1e6))
    r = np.median(np.array(rows[2:]), axis=0)
    g.set_profiling(1)
    g.synchronize()
    g.step_n(0.001, steps)
    g.synchronize()
    dev_ms, nl = g.last_step_kernel_ms()
    g.set_profiling(0)
    print(f"K = {steps}: call returns after {r[0]:.1f} us, synchronize takes {r[1]:.1f} us more, region {r[2]:.1f} us = {r[2] / steps:.2f} us per step; "
          f"device time {dev_ms * nl * 1e3:.1f} us = {dev_ms * nl * 1e3 / steps:.2f} per step; overhead {r[2] - dev_ms * nl * 1e3:.1f} us", flush=True)
