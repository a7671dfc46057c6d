#!/usr/bin/env python3
"""Is a run of steps bound by the host's launch loop?  Host time of mrs_swarm_step_n (the call returns when everything is ENQUEUED)
against the time until the device has finished, two streams and one.  usage: host_launch_rate.py [n_uavs] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import mrs_multirotor_simulator_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
st, cmd = bench.make_inputs(n, "actuator", 3)
g = M.Swarm(n, arith=M.ARITH_FAST)
g.construct(0, n, M.model_params("x500", ground_enabled=True))
g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
g.set_input(0, n, M.ACTUATOR_CMD, cmd)
g.step_n(0.001, 200); g.synchronize()
for rep in range(3):
    g.synchronize()
    t0 = time.perf_counter(); g.step_n(0.001, steps); t1 = time.perf_counter(); g.synchronize(); t2 = time.perf_counter()
    print(f"{n} UAVs, {steps} steps: enqueued after {(t1 - t0) / steps * 1e6:.2f} us per step, finished after {(t2 - t0) / steps * 1e6:.2f} us per step "
          f"({os.environ.get('MRS_SPLIT_STREAMS', '1') != '0' and 'two' or 'one'} stream(s))", flush=True)
