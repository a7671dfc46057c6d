#!/usr/bin/env python3
"""Host<->device rates of the boundary calls (PCIe-inclusive): command upload, packed publisher download, raw getState."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrs_multirotor_simulator_amd as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sw = M.Swarm(n, arith=M.ARITH_FAST)
sw.construct(0, n, M.model_params("x500", ground_enabled=True))
cmd = np.random.default_rng(0).uniform(0.4, 0.6, (n, 4))
def rate(fn, reps=20):
    fn(); sw.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    sw.synchronize(); return (time.perf_counter() - t0) / reps
t = rate(lambda: sw.set_input(0, n, M.ACTUATOR_CMD, cmd)); print(f"set_input(ACTUATOR) {n} UAVs: {t*1e3:.3f} ms  ({n*32/t/1e9:.2f} GB/s payload)")
t = rate(lambda: sw.get_outputs()); print(f"get_outputs         {n} UAVs: {t*1e3:.3f} ms  ({n*136/t/1e9:.2f} GB/s payload)")
t = rate(lambda: sw.get_state(), 5); print(f"get_state (6 arrays) {n} UAVs: {t*1e3:.3f} ms")
t = rate(lambda: sw.get_outputs_view()); print(f"get_outputs_view    {n} UAVs: {t*1e3:.3f} ms  ({n*136/t/1e9:.2f} GB/s payload)")
def staged():
    rows = sw.input_staging(n, 4); rows[:] = cmd; sw.commit_input(0, n, M.ACTUATOR_CMD, 4)
t = rate(staged); print(f"staged input        {n} UAVs: {t*1e3:.3f} ms  ({n*32/t/1e9:.2f} GB/s payload)")
t1 = rate(lambda: sw.step_n(0.001, 1)); print(f"step                {n} UAVs: {t1*1e6:.1f} us")
def tick():
    sw.set_input(0, n, M.ACTUATOR_CMD, cmd); sw.step_n(0.001, 1); sw.get_outputs()
t = rate(tick); print(f"upload+step+download {n} UAVs: {t*1e3:.3f} ms -> {n/t:.3e} UAV-steps/s PCIe-inclusive")
def tick2():
    staged(); sw.step_n(0.001, 1); sw.get_outputs_view()
t = rate(tick2); print(f"staged upload+step+view {n} UAVs: {t*1e3:.3f} ms -> {n/t:.3e} UAV-steps/s PCIe-inclusive")
