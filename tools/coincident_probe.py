"""How the collision pass behaves when MANY UAVs share one position (every list overflows, every UAV has more partners in contact than
the query's hit lists hold: the reference path, quadratic per UAV): time of one handleCollisions.  usage: coincident_probe.py n ..."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import mrs_multirotor_simulator_amd as M  # noqa: E402

for n in [int(a) for a in sys.argv[1:]] or [200, 500, 1000]:
    g = M.Swarm(n, arith=M.ARITH_LITERAL)
    g.construct(0, n, M.model_params("x500"), np.tile([[1.0, 2.0, 3.0]], (n, 1)), np.zeros(n))
    t0 = time.perf_counter()
    g.handle_collisions(True, False, 100.0)
    f = g.get_external_force()
    t1 = time.perf_counter()
    print(f"{n} UAVs at one point: handleCollisions + read-back {1e3 * (t1 - t0):.1f} ms, largest force {np.abs(f).max():.3g}", flush=True)
