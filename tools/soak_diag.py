#!/usr/bin/env python3
"""Diagnosis of a FAST-vs-oracle excursion in tests/campaigns/soak.py: the same scenario, compared every `step` ticks from `start`,
with the worst UAV's motor speeds, flags and height on both sides.  usage: soak_diag.py [n] [start] [stop] [step]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from helpers import Pair
import mrs_multirotor_simulator_amd as M
from oracle import oracle_swarm as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
start, stop, step = (int(sys.argv[k]) if len(sys.argv) > k else d for k, d in ((2, 1250), (3, 1500), (4, 10)))
DT, chunk, vol = 0.001, 250, 64.0
rng = np.random.default_rng(2026)
side = (vol * n) ** (1.0 / 3.0)
p = Pair(M, n, arith=M.ARITH_FAST)
pos = rng.uniform(0, side, (n, 3)) + [0, 0, 1.0]
p.construct(0, n, "x500", pos=pos, heading=rng.uniform(-3, 3, n), ground_enabled=True, ground_z=0.0)
p.both("set_input", 0, n, O.POSITION_CMD, np.concatenate([pos + rng.uniform(-6, 6, (n, 3)), rng.uniform(-3, 3, (n, 1))], axis=1))


def run(k):
    for _ in range(k):
        p.o.step_n(DT, 1, 16)
        p.o.handle_collisions(True, False, 100.0)
    p.g.tick_n(DT, k, True, False, 100.0)


tick = 0
for c in range(stop // chunk + 1):
    if c % 2 == 1:
        a = int(rng.integers(0, n - n // 3))
        x = p.o.get_state(a, n // 3)["x"]
        p.both("set_input", a, n // 3, O.POSITION_CMD, np.concatenate([x + rng.uniform(-8, 8, (n // 3, 3)), rng.uniform(-3, 3, (n // 3, 1))], axis=1))
        p.both("crash", int(rng.integers(0, n - 5)), 5)
        p.both("set_hold", int(rng.integers(0, n - 50)), 50, bool(c % 4 == 1))
    left = chunk
    while left > 0 and tick < stop:
        k = left if tick + left <= start else (start - tick if tick < start else min(step, left))
        run(k)
        tick += k
        left -= k
        if tick >= start:
            sg, so = p.g.get_state(), p.o.get_state()
            err = np.abs(sg["motor_rpm"] - so["motor_rpm"]).max(axis=1)
            w = int(err.argmax())
            ex = np.abs(sg["x"] - so["x"]).max(axis=1)
            fo = p.o.get_external_force()[w]
            print(f"tick {tick}: worst rpm UAV {w} err {err[w]:.3e} (x err {ex[w]:.2e}, worst x err {ex.max():.2e} at {int(ex.argmax())}); rpm gpu {sg['motor_rpm'][w, :4]} "
                  f"oracle {so['motor_rpm'][w, :4]}; z {so['x'][w, 2]:.4f} vz {so['v'][w, 2]:.3f} crashed {bool(p.o.has_crashed()[w])} force {fo}", flush=True)
