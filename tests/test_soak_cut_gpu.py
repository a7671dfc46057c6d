"""A cut of tests/campaigns/soak.py inside the suite: 20 000 x500 UAVs at 30 m^3 per UAV, 500 ticks of the timerMain
order with elastic collisions, FAST arithmetic, goals that change half way, a few crashes and UAVs on hold — the WHOLE swarm on the
oracle (step + handle_collisions per tick), compared after every 250 ticks: state, PIDs, IMU, latched forces, crash flags."""
import numpy as np
import pytest

import helpers
from helpers import Pair, RTOL_NORTH_STAR

pytestmark = pytest.mark.gpu
DT = 0.001


def test_soak_cut_20k_uavs_500_ticks(mrs, oracle):
    M, O = mrs, oracle
    n, ticks, chunk, vol = 20_000, 500, 250, 30.0
    rng = np.random.default_rng(2026)
    side = (vol * n) ** (1.0 / 3.0)
    p = Pair(M, n, arith=M.ARITH_FAST)
    pos = rng.uniform(0, side, (n, 3)) + [0, 0, 1.0]
    p.construct(0, n, "x500", pos=pos, heading=rng.uniform(-3, 3, n), ground_enabled=True, ground_z=0.0)
    p.both("set_input", 0, n, O.POSITION_CMD, np.concatenate([pos + rng.uniform(-6, 6, (n, 3)), rng.uniform(-3, 3, (n, 1))], axis=1))
    worst = 0.0
    for c in range(ticks // chunk):
        if c % 2 == 1:  # new goals for a third of the swarm, a few crashes, a few UAVs on hold
            a = int(rng.integers(0, n - n // 3))
            x = p.o.get_state(a, n // 3)["x"]
            p.both("set_input", a, n // 3, O.POSITION_CMD, np.concatenate([x + rng.uniform(-8, 8, (n // 3, 3)), rng.uniform(-3, 3, (n // 3, 1))], axis=1))
            p.both("crash", int(rng.integers(0, n - 5)), 5)
            p.both("set_hold", int(rng.integers(0, n - 50)), 50, True)
        for _ in range(chunk):
            p.o.step_n(DT, 1, 16)
            p.o.handle_collisions(True, False, 100.0)
        p.g.tick_n(DT, chunk, True, False, 100.0)
        worst = max(worst, p.compare(RTOL_NORTH_STAR, f"after {(c + 1) * chunk} ticks"))
        helpers.assert_close(p.g.get_external_force(), p.o.get_external_force(), RTOL_NORTH_STAR, "forces")
        assert np.array_equal(p.g.has_crashed(), p.o.has_crashed())
    touched = int((np.abs(p.o.get_external_force()).sum(axis=1) > 0).sum())
    fused, stalls, replayed, ahead = p.g.fused_stats()
    assert touched > 50 and fused > 0.8 * ticks, (touched, fused)
    print(f"soak cut: {n} UAVs x {ticks} ticks (FAST): worst relative error {worst:.2e}, {touched} UAVs in contact at the end; "
          f"{fused} fused launches, {ahead} searches queued ahead, {stalls} stalls / {replayed} replayed")
