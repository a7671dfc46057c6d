#!/usr/bin/env python3
"""Long differential run: a dense, moving swarm with elastic collisions, position commands that change, occasional crashes and
holds — product (tick_n) vs oracle (step + handle_collisions per tick), compared every `chunk` ticks.
usage: tests/campaigns/soak.py [n_uavs] [n_ticks] [literal|fast] [m^3 per UAV] [local|sharded]
`sharded` drives the ticks through mrs_swarm_tick_sharded_n with a one-rank RCCL communicator (the multi-GPU code path)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import Pair
import mrs_multirotor_simulator_amd as M
from oracle import oracle_swarm as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
fast = len(sys.argv) > 3 and sys.argv[3] == "fast"
rtol = 1e-6 if fast else 1e-10
DT, chunk = 0.001, 250
rng = np.random.default_rng(2026)
vol = float(sys.argv[4]) if len(sys.argv) > 4 else 30.0  # m^3 per UAV (30: plenty of contacts)
sharded = len(sys.argv) > 5 and sys.argv[5] == "sharded"
side = (vol * n) ** (1.0 / 3.0)
p = Pair(M, n, arith=M.ARITH_FAST if fast else M.ARITH_LITERAL)
pos = rng.uniform(0, side, (n, 3)) + [0, 0, 1.0]
p.construct(0, n, "x500", pos=pos, heading=rng.uniform(-3, 3, n), ground_enabled=True, ground_z=0.0)
p.both("set_input", 0, n, O.POSITION_CMD, np.concatenate([pos + rng.uniform(-6, 6, (n, 3)), rng.uniform(-3, 3, (n, 1))], axis=1))
if sharded:
    from mrs_multirotor_simulator_amd.swarm import rccl_unique_id
    p.g.comm_init(1, 0, rccl_unique_id(), n)
t0 = time.time()
worst = 0.0
grounded = np.zeros(n, dtype=bool)
for c in range(ticks // chunk):
    if c % 2 == 1:  # new goals for a third of the swarm, a few crashes, a few UAVs on hold
        a = int(rng.integers(0, n - n // 3))
        x = p.o.get_state(a, n // 3)["x"]
        p.both("set_input", a, n // 3, O.POSITION_CMD, np.concatenate([x + rng.uniform(-8, 8, (n // 3, 3)), rng.uniform(-3, 3, (n // 3, 1))], axis=1))
        p.both("crash", int(rng.integers(0, n - 5)), 5)
        p.both("set_hold", int(rng.integers(0, n - 50)), 50, bool(c % 4 == 1))
    for _ in range(chunk):
        p.o.step_n(DT, 1, 16)
        p.o.handle_collisions(True, False, 100.0)
    (p.g.tick_sharded_n if sharded else p.g.tick_n)(DT, chunk, True, False, 100.0)
    # A UAV whose goal lies below the ground is pressed against it: clamped every tick, its attitude and rate loops fight the clamp
    # with wound-up integrators — a DIVERGING closed loop (tools/soak_diag.py: the difference of two runs that start 1e-13 apart
    # triples every 25 ticks; motor speeds 6 800 / 4 200 / 1 200 / 3 700 rpm on one airframe).  LITERAL follows the oracle through
    # that bit for bit; FAST's last-bit differences leave any fixed tolerance there, so FAST compares the UAVs that have not been
    # found on the ground at a checkpoint.
    grounded |= p.o.get_state()["x"][:, 2] <= 1e-9
    e = p.compare(rtol, f"after {(c + 1) * chunk} ticks", mask=~grounded if fast else None)
    helpers.assert_close(p.g.get_external_force(), p.o.get_external_force(), max(rtol, 1e-11), "forces")
    assert np.array_equal(p.g.has_crashed(), p.o.has_crashed())
    worst = max(worst, e)
    touched = int((np.abs(p.o.get_external_force()).sum(axis=1) > 0).sum())
    print(f"tick {(c + 1) * chunk:6d}: max rel err {e:.2e}" + (f" ({int(grounded.sum())} grounded UAVs left out)" if fast else "") + f", {touched} UAVs in contact, collision stats {p.g.collision_stats()}, {time.time() - t0:.0f} s", flush=True)
print("SOAK OK", n, "UAVs", ticks, "ticks", "fast" if fast else "literal", "sharded" if sharded else "local", "worst", worst)
