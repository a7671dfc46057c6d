"""Is the two-stream run of steps bound by the ONE host thread that feeds both streams?  Two swarms of half the size, each stepped on
its own stream by its own host thread (ctypes releases the GIL), against one swarm of the full size stepped by mrs_swarm_step_n (which
alternates between its two streams).  usage: two_host_threads.py [n_uavs] [steps]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import bench  # noqa: E402
import mrs_multirotor_simulator_amd as M  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000


def make(n_part, seed):
    st, cmd = bench.make_inputs(n_part, "actuator", seed)
    g = M.Swarm(n_part, arith=M.ARITH_FAST)
    g.construct(0, n_part, M.model_params("x500", ground_enabled=True))
    g.set_state(0, n_part, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    g.set_input(0, n_part, M.ACTUATOR_CMD, cmd)
    g.step_n(0.001, 100)
    g.synchronize()
    return g


whole = make(n, 3)
for rep in range(3):
    whole.synchronize()
    t0 = time.perf_counter()
    whole.step_n(0.001, steps)
    whole.synchronize()
    t1 = time.perf_counter()
    print(f"one swarm of {n}, one host thread, two streams: {(t1 - t0) / steps * 1e6:.2f} us per step", flush=True)
os.environ["MRS_SPLIT_STREAMS"] = "0"
halves = [make(n // 2, 4), make(n - n // 2, 5)]
for rep in range(3):
    for g in halves:
        g.synchronize()
    go = threading.Barrier(3)

    def work(g):
        go.wait()
        g.step_n(0.001, steps)
        g.synchronize()

    th = [threading.Thread(target=work, args=(g,)) for g in halves]
    for t in th:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    t1 = time.perf_counter()
    print(f"two swarms of {n // 2}, a host thread and a stream each: {(t1 - t0) / steps * 1e6:.2f} us per step of both", flush=True)
