#!/usr/bin/env python3
"""Prints the collision pass's device control words (skin flags, search counter, overflow statistics) for a sparse and a dense swarm."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers
import mrs_multirotor_simulator_amd as M
from mrs_multirotor_simulator_amd import swarm as sw_mod
lib = sw_mod.load_library()
for vol in (64, 16):
    n = 100000
    rng = np.random.default_rng(3)
    side = (vol * n) ** (1 / 3)
    pos = rng.uniform(0, 1, (n, 3)) * [side * 2, side * 2, side / 4] + [0, 0, 5]
    g = M.Swarm(n, arith=M.ARITH_FAST)
    g.construct(0, n, M.model_params("x500", ground_enabled=True, ground_z=0.0), pos, np.zeros(n))
    for k in range(4):
        g.handle_collisions(True, False, 100.0)
        out = (C.c_uint32 * 8)()
        lib.mrs_swarm_debug_collision_words(g._h, out)
        print(vol, "call", k, "ctl", list(out), "stats", g.collision_stats())
