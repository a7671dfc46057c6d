#!/usr/bin/env python3
"""Sharded swarms with the HOST writing between calls — what the halo exchange of a search tick has to notice by itself: UAVs carried
somewhere else by set_state (a few metres, across the slab, out of the swarm's hull, next to a UAV of another rank), put on hold and
released, commands replaced; `runs` random scenarios (world 2..6, slabs, LITERAL), every call checked against the oracle (1e-11 on forces)
and the halo statistics printed.  A halo that missed a record is a missing partner: a wrong force.
usage: halo_host_writes.py [runs] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import mrs_multirotor_simulator_amd as M  # noqa: E402
from helpers import RTOL_LITERAL  # noqa: E402
from oracle import oracle_swarm as oracle  # noqa: E402
from test_export_sets_gpu import DT, VirtualShards, moving_swarm  # noqa: E402


def scenario(seed, verbose=True):
    rng = np.random.default_rng(90_000 + seed)
    world = int(rng.integers(2, 7))
    n_total = int(rng.integers(900, 1800)) * world
    speed = float(rng.uniform(2.0, 6.0))
    pos, st, cmd = moving_swarm(rng, n_total, speed=speed)
    pos[:, 1:] *= [3.0, 1.0]  # slabs that are wide in y: halos stay a fraction of a shard
    st["x"] = pos.copy()
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world)
    vs = VirtualShards(M, world, order, helpers.to_product_params(M, po), pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_LITERAL, M.EXCHANGE_EXPORT_SETS,
                       rendezvous=bool(rng.integers(0, 2)))
    where = {int(p): (r, k) for r, (_, idx) in enumerate(vs.shards) for k, p in enumerate(idx)}
    t0, done, log = time.time(), 0, []
    for call in range(6):
        n = int(rng.integers(10, 70))
        vs.tick_n(n, True, False, 100.0)
        for _ in range(n):
            o.step_n(DT, 1, 8)
            o.handle_collisions(True, False, 100.0)
        done += n
        a, so = vs.gather(), o.get_state()
        helpers.assert_close(a["f"], o.get_external_force(), 1e-11, f"seed {seed}: forces after {done} ticks ({log})")
        for key in ("x", "v", "R", "omega", "motor_rpm"):
            helpers.assert_close(a[key], so[key], RTOL_LITERAL, f"seed {seed}: {key} after {done} ticks ({log})")
        # the host writes: 0-3 of them before the next call
        for _ in range(int(rng.integers(0, 4))):
            kind = str(rng.choice(["nudge", "across", "outside", "beside", "hold", "release", "command"]))
            i = int(rng.integers(0, n_total))
            r, k = where[i]
            g = vs.shards[r][0]
            tele = {key: v[i:i + 1].copy() for key, v in so.items()}
            if kind == "nudge":
                tele["x"][0] += rng.normal(0, 2.0, 3)
            elif kind == "across":
                tele["x"][0, 0] = rng.uniform(so["x"][:, 0].min(), so["x"][:, 0].max())
            elif kind == "outside":
                tele["x"][0] += rng.choice([-1.0, 1.0], 3) * rng.uniform(5.0, 40.0, 3) * (rng.random(3) < 0.5)
            elif kind == "beside":
                j = int(rng.integers(0, n_total))
                tele["x"][0] = so["x"][j] + rng.normal(0, 0.4, 3) + [0.5, 0, 0]
            if kind in ("nudge", "across", "outside", "beside"):
                tele["x"][0, 2] = max(tele["x"][0, 2], 1.0)
                o.set_state(i, 1, tele["x"], tele["v"], tele["R"], tele["omega"], tele["motor_rpm"])
                g.set_state(k, 1, tele["x"], tele["v"], tele["R"], tele["omega"], tele["motor_rpm"])
            elif kind in ("hold", "release"):
                o.set_hold(i, 1, kind == "hold")
                g.set_hold(k, 1, kind == "hold")
            else:
                c = rng.uniform(0.35, 0.6, (1, 4))
                o.set_input(i, 1, oracle.ACTUATOR_CMD, c)
                g.set_input(k, 1, M.ACTUATOR_CMD, c)
            log.append(f"{kind}@{done}")
            so = o.get_state()
    stats = [g.search_stats() for g, _ in vs.shards]
    assert len(set(s[:3] for s in stats)) == 1, f"seed {seed}: the ranks disagree on their searches: {stats}"
    vs.close()
    if verbose:
        print(f"seed {seed}: world {world}, {n_total} UAVs, speed {speed:.1f}, {done} ticks, writes {log}: searches / on halos / repeated {stats[0][:3]}, "
              f"{time.time() - t0:.1f} s", flush=True)
    return stats[0]


if __name__ == "__main__":
    os.environ["MRS_SHARD_SPLIT_MIN_BLOCKS"] = "1"  # (split ticks on swarms of a few thousand UAVs: read when a swarm is created)
    os.environ["MRS_SHARD_SPLIT_MAX_FRACTION"] = "0.95"
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    M.load_library()
    tot = np.zeros(3, dtype=np.int64)
    for seed in range(first, first + runs):
        tot += np.array(scenario(seed)[:3])
    print(f"{runs} scenarios OK: {tot[0]} searches, {tot[1]} on halos, {tot[2]} repeated on all records")
