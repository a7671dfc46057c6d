#!/usr/bin/env python3
"""More seeds of tests/test_sharded_chaos_gpu.py: the lock-free sharded protocol (split ticks, announced stall indices) under host
skew, `runs` scenarios with random world size (2..8), swarm size, speeds, chaos amplitude, loopback mode and call lengths, each
against the oracle (LITERAL, 1e-11).  A protocol slip is a collective mismatch / time-out, a missed stall a wrong force.
usage: chaos_seeds.py [runs] [first_seed] [min UAVs per rank, default 600] [max, default 1200] [literal|fast]
(fast: the FAST kernels of the split tick — boundary, non-temporal interior — held to 1e-7 over the few hundred ticks of a scenario)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MRS_SHARD_SPLIT_MIN_BLOCKS"] = "1"
os.environ["MRS_SHARD_SPLIT_MAX_FRACTION"] = "0.95"
import helpers  # noqa: E402
import mrs_multirotor_simulator_amd as M  # noqa: E402
from helpers import RTOL_LITERAL  # noqa: E402
from oracle import oracle_swarm as oracle  # noqa: E402
from test_export_sets_gpu import DT, VirtualShards, moving_swarm  # noqa: E402

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
per_lo = int(sys.argv[3]) if len(sys.argv) > 3 else 600
per_hi = int(sys.argv[4]) if len(sys.argv) > 4 else 1200
fast = len(sys.argv) > 5 and sys.argv[5] == "fast"
rtol_state, rtol_force = (1e-7, 1e-7) if fast else (RTOL_LITERAL, 1e-11)
M.load_library()
for seed in range(first, first + runs):
    rng = np.random.default_rng(70_000 + seed)
    world = int(rng.integers(2, 9))
    n_total = int(rng.integers(per_lo, per_hi)) * world
    speed = float(rng.uniform(3.0, 9.0))
    chaos = int(rng.choice([0, 50, 300, 1000]))
    rendezvous = bool(rng.integers(0, 2))
    slabs = rng.random() < 0.8
    crash_call = int(rng.integers(0, 4))
    pos, st, cmd = moving_swarm(rng, n_total, speed=speed)
    hot = rng.choice(n_total, 30, replace=False)
    st["v"][hot] = rng.normal(0, 1, (30, 3)) * [14.0, 14.0, 4.0]
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world) if slabs else np.arange(n_total)
    vs = VirtualShards(M, world, order, helpers.to_product_params(M, po), pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_FAST if fast else M.ARITH_LITERAL,
                       M.EXCHANGE_EXPORT_SETS, rendezvous=rendezvous)
    if chaos:
        for r, (g, _) in enumerate(vs.shards):
            g.debug_chaos(chaos, seed=1000 * seed + r)
    calls = [int(c) for c in rng.integers(1, 90, 4)]
    t0, done = time.time(), 0
    for k, n in enumerate(calls):
        crash = k == crash_call
        vs.tick_n(n, True, crash, 100.0)
        for _ in range(n):
            o.step_n(DT, 1, 8)
            o.handle_collisions(True, crash, 100.0)
        done += n
        a, so = vs.gather(), o.get_state()
        assert np.array_equal(a["crashed"], o.has_crashed()), f"seed {seed}: crash flags after {done} ticks"
        helpers.assert_close(a["f"], o.get_external_force(), rtol_force, f"seed {seed}: forces after {done} ticks")
        for key in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(a[key], so[key], rtol_state, f"seed {seed}: {key} after {done} ticks")
    info, split = vs.info(), [g.split_stats()[0] for g, _ in vs.shards]
    vs.close()
    print(f"seed {seed}: world {world}, {n_total} UAVs, {'slabs' if slabs else 'index shards'}, speed {speed:.1f}, chaos {chaos} us, "
          f"{'rendezvous' if rendezvous else 'barrier'}, calls {calls} (crash mode in call {crash_call}): searches {[c['searches'] for c in info]}, "
          f"split ticks {split}, {time.time() - t0:.1f} s", flush=True)
print(f"{runs} scenarios OK")
