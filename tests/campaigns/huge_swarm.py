#!/usr/bin/env python3
"""One-off capacity check: a swarm far beyond the 4 GiB reach of buffer addressing (default 50 M UAVs = 34 GB of state), built as
copies of a 4096-UAV swarm; every copy must stay bit-identical to the first one and the first one must follow the oracle.
usage: tests/campaigns/huge_swarm.py [n_uavs] [steps]   (host memory needed: about 1 kB per UAV)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
import mrs_multirotor_simulator_amd as M
from oracle import oracle_swarm as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m, DT = 4096, 0.001
reps = -(-n // m)
rng = np.random.default_rng(56)
st = helpers.random_state(rng, m, 4)
cmd = rng.uniform(0.35, 0.6, (m, 4))
tile = lambda a: np.concatenate([a] * reps, axis=0)[:n]
t0 = time.time()
g = M.Swarm(n, arith=M.ARITH_FAST)
g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), tile(st["x"]), np.zeros(n))
print(f"constructed {n} UAVs in {time.time() - t0:.1f} s", flush=True)
g.set_state(0, n, tile(st["x"]), tile(st["v"]), tile(st["R"]), tile(st["omega"]), tile(st["motor_rpm"]))
g.set_input(0, n, M.ACTUATOR_CMD, tile(cmd))
print(f"state and commands uploaded at {time.time() - t0:.1f} s", flush=True)
g.set_profiling(1)
g.step_n(DT, steps, 1)
g.synchronize()
ms, nl = g.last_step_kernel_ms()
print(f"{steps} steps: {ms:.3f} ms per step = {n / (ms * 1e-3):.3e} UAV-steps/s, {492 * n / ms / 1e6:.0f} GB/s algorithmic", flush=True)
out = g.get_state()
print(f"state downloaded at {time.time() - t0:.1f} s", flush=True)
del g
for k, a in out.items():
    first = a[:m]
    body = a[: (n // m) * m].reshape(n // m, *first.shape)
    assert (body == first[None]).all(), f"{k}: some copy differs from copy 0"
    tail = a[(n // m) * m:]
    assert np.array_equal(tail, first[:len(tail)]), f"{k}: tail copy differs"
o = O.OracleSwarm(m)
o.construct(0, m, helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0), st["x"], np.zeros(m))
o.set_state(0, m, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
o.set_input(0, m, O.ACTUATOR_CMD, cmd)
o.step_n(DT, steps)
ref = o.get_state()
for k in ("x", "v", "R", "omega", "motor_rpm"):
    helpers.assert_close(out[k][:m], ref[k], helpers.RTOL_NORTH_STAR, k)
print(f"HUGE SWARM OK: {n} UAVs, {steps} steps, all {reps} copies identical, first copy follows the oracle ({time.time() - t0:.0f} s)")
