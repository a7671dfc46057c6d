"""The HALO exchange of a neighbour search of the export-set exchange (collide.hip mrs_collide_halo_*): a search tick sends the
records that lie inside another rank's box of the last search, 64 B each, instead of gathering every rank's 48-B records.  Which
exchange a search used must not show in any result: the same swarm is run with the halo exchange and with MRS_SEARCH_HALO=0 and the
states are compared bit for bit, then against the oracle; the cases that make a halo search repeat itself on all records (a UAV the
host has moved out of its rank's hull, shards without spatial order whose halos would be the whole swarm) are driven on purpose."""
import numpy as np
import pytest

import helpers
from helpers import RTOL_LITERAL
from test_export_sets_gpu import DT, VirtualShards, moving_swarm

pytestmark = pytest.mark.gpu


def _swarms(M, oracle, monkeypatch, world, n_total, seed, slabs=True, speed=5.0, lost=()):
    rng = np.random.default_rng(seed)
    pos, st, cmd = moving_swarm(rng, n_total, speed=speed)
    for i in lost:  # beyond the 1e9 m inside which a position takes part in the collision pass at all: in no hull, in no halo, in no list
        pos[i, 0] = st["x"][i, 0] = 2.5e9
    po = helpers.oracle_params("x500", ground_enabled=True, ground_z=0.0)
    o = oracle.OracleSwarm(n_total)
    o.construct(0, n_total, po, pos, np.zeros(n_total))
    o.set_state(0, n_total, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    o.set_input(0, n_total, oracle.ACTUATOR_CMD, cmd)
    order = M.slab_partition(pos, world) if slabs else np.arange(n_total)
    pp = helpers.to_product_params(M, po)
    out = []
    for halo in ("1", "0"):  # (read when the communicator is bound)
        monkeypatch.setenv("MRS_SEARCH_HALO", halo)
        out.append(VirtualShards(M, world, order, pp, pos, np.zeros(n_total), st, M.ACTUATOR_CMD, cmd, M.ARITH_LITERAL, M.EXCHANGE_EXPORT_SETS))
    return o, out[0], out[1]


def _oracle_ticks(o, n, crash=False):
    for _ in range(n):
        o.step(DT)
        o.handle_collisions(True, crash, 100.0)


def _same(a, b, what):
    for k in ("x", "v", "R", "omega", "motor_rpm", "f", "crashed", "pid", "imu"):
        assert np.array_equal(a[k], b[k], equal_nan=a[k].dtype.kind == "f"), f"{k}: halo search vs full search, {what}"


@pytest.mark.parametrize("world,n_total", [(4, 6000), (3, 2501)])
def test_halo_searches_change_no_result(mrs, oracle, monkeypatch, world, n_total):
    M = mrs
    # (the smaller swarm also carries two UAVs at positions no collision pass looks at: the slab partition puts them into the last slab,
    #  whose hull must not grow by them)
    o, halo, full = _swarms(M, oracle, monkeypatch, world, n_total, 50 + world, lost=(10, 2000) if n_total < 3000 else ())
    done = 0
    for n, crash in ((90, False), (1, True), (110, False), (100, False)):
        halo.tick_n(n, True, crash, 100.0)
        full.tick_n(n, True, crash, 100.0)
        _oracle_ticks(o, n, crash)
        done += n
        a, b = halo.gather(), full.gather()
        _same(a, b, f"after {done} ticks")
    so = o.get_state()
    assert np.array_equal(a["crashed"], o.has_crashed()) and o.has_crashed().sum() > 0
    helpers.assert_close(a["f"], o.get_external_force(), 1e-11, "forces")
    for k in ("x", "v", "R", "omega", "motor_rpm"):
        helpers.assert_close(a[k], so[k], RTOL_LITERAL, k)
    for (g, _), (gf, _) in zip(halo.shards, full.shards):
        searches, on_halo, repeats, cap = g.search_stats()
        assert gf.search_stats()[1:] == (0, 0, 0), gf.search_stats()
        # most searches ran on a halo; a repeat counts as a search of its own; the searches
        # that end a spell of full-gather ticks (a UAV with more neighbours than its list holds) gather all records
        assert on_halo >= 3 and repeats <= 1 and 1 + on_halo + repeats <= searches <= gf.search_stats()[0] + repeats, (searches, on_halo, repeats, cap)
        ci, cf = g.comm_info(), gf.comm_info()
        assert cap == 0 or ci["bytes_per_rebuild"] < cf["bytes_per_rebuild"], (cap, ci, cf)  # (slabs 18 m wide, halos of 3.7 m either side: 64 B x 41 % against 48 B)
    print("halo searches:", [g.search_stats() for g, _ in halo.shards], "bytes per search tick", halo.info()[0]["bytes_per_rebuild"], "against",
          full.info()[0]["bytes_per_rebuild"])
    halo.close()
    full.close()


def test_a_uav_moved_out_of_its_ranks_hull_repeats_the_search_on_all_records(mrs, oracle, monkeypatch):
    """set_state drops a UAV of the first slab six metres below everybody: its rank's box of the last search — what the other ranks
    choose their halo entries by — no longer covers the rank.  The rank says so in its halo header, every rank repeats that search on
    all records, and the searches after it are halo searches again (the new box has travelled with the repeat).
    (A UAV carried ACROSS the swarm makes its rank's box the whole swarm: tests/test_export_sets_gpu.py teleports one, and the halo of
    such a rank is everything — the case of the test below.)"""
    M = mrs
    world, n_total = 4, 4000
    o, halo, full = _swarms(M, oracle, monkeypatch, world, n_total, 77, speed=2.0)
    for vs in (halo, full):
        vs.tick_n(60, True, False, 100.0)
    _oracle_ticks(o, 60)
    before = [g.search_stats() for g, _ in halo.shards]
    so = o.get_state()
    mover = int(halo.shards[0][1][3])  # public index of UAV 3 of slab 0
    tele = {k: v[mover:mover + 1].copy() for k, v in so.items()}
    tele["x"][0, 2] = so["x"][:, 2].min() - 6.0
    o.set_state(mover, 1, tele["x"], tele["v"], tele["R"], tele["omega"], tele["motor_rpm"])
    for vs in (halo, full):
        vs.shards[0][0].set_state(3, 1, tele["x"], tele["v"], tele["R"], tele["omega"], tele["motor_rpm"])
        vs.tick_n(120, True, False, 100.0)
    _oracle_ticks(o, 120)
    a, b, so = halo.gather(), full.gather(), o.get_state()
    _same(a, b, "after the move")
    helpers.assert_close(a["f"], o.get_external_force(), 1e-11, "forces after the move")
    for k in ("x", "v", "R", "omega", "motor_rpm"):
        helpers.assert_close(a[k], so[k], RTOL_LITERAL, k)
    after = [g.search_stats() for g, _ in halo.shards]
    assert all(x[2] + 1 <= y[2] <= x[2] + 2 for x, y in zip(before, after)), (before, after)  # the repeat, on every rank
    assert all(y[1] >= x[1] + 2 and y[3] > 0 for x, y in zip(before, after)), (before, after)  # ... and halo searches again afterwards
    halo.close()
    full.close()


def test_shards_without_spatial_order_fall_back_to_the_full_gather(mrs, oracle, monkeypatch):
    """index shards of a random swarm: every rank's box is the whole swarm, a halo would carry every record at 64 B instead of 48 —
    the first halo search reports that, is repeated on all records, and the next searches do not try again for a while"""
    M = mrs
    world, n_total = 3, 1800
    o, halo, full = _swarms(M, oracle, monkeypatch, world, n_total, 5, slabs=False)
    for vs in (halo, full):
        vs.tick_n(150, True, False, 100.0)
    _oracle_ticks(o, 150)
    a, b, so = halo.gather(), full.gather(), o.get_state()
    _same(a, b, "index shards")
    for k in ("x", "v", "R", "omega", "motor_rpm"):
        helpers.assert_close(a[k], so[k], RTOL_LITERAL, k)
    for g, _ in halo.shards:
        searches, on_halo, repeats, cap = g.search_stats()
        assert searches >= 3 and on_halo == repeats and on_halo <= 1 + searches // 16 and cap == 0, (searches, on_halo, repeats, cap)
    halo.close()
    full.close()


@pytest.mark.parametrize("seed", [3, 11])
def test_halo_searches_under_random_host_writes(mrs, oracle, monkeypatch, seed):
    """tests/campaigns/halo_host_writes.py, two seeds of it: UAVs moved by set_state between calls (a few metres, across the slabs, out of
    the hull, next to a UAV of another rank), holds, new commands — every call against the oracle, every rank with the same counters"""
    import importlib.util
    import os
    monkeypatch.setenv("MRS_SHARD_SPLIT_MIN_BLOCKS", "1")  # split ticks on swarms of a few thousand UAVs, as the campaign runs them
    monkeypatch.setenv("MRS_SHARD_SPLIT_MAX_FRACTION", "0.95")
    spec = importlib.util.spec_from_file_location("halo_host_writes", os.path.join(os.path.dirname(os.path.abspath(__file__)), "campaigns", "halo_host_writes.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    searches, on_halo, repeats = mod.scenario(seed)[:3]
    assert searches >= 3 and on_halo >= 1, (searches, on_halo, repeats)


def test_measurement_stand_in_rank_is_the_same_rank_with_either_exchange(mrs, monkeypatch):
    """bench.py's `sharded_rank_standin` record: ONE rank of eight behind mrs_swarm_comm_init_standin, whose neighbours are periodic
    images of itself (records, halo entries and search boxes moved one slab width to either side by the stand-in collective).  A
    time measurement, not a simulation — but a deterministic one: the rank must end in the same state bit for bit whether its
    searches exchange halos or all records, and whether or not the stand-in charges the collectives' bytes (MRS_STANDIN_GBPS)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from mrs_multirotor_simulator_amd.sharded import shard_range
    M = mrs
    world, n_per = 8, 12_000
    rank, n_total = world // 2, n_per * world
    st, cmd = bench.make_inputs(n_total, "position+collisions", seed=5)
    order = M.slab_partition(st["x"], world)
    lo, hi = shard_range(n_total, world, rank)
    idx = order[lo:hi]
    width = float(st["x"][idx, 0].max() - st["x"][idx, 0].min()) * (1.0 + 1.0 / len(idx))
    monkeypatch.setenv("MRS_SHARD_SPLIT_MIN_BLOCKS", "1")
    monkeypatch.setenv("MRS_SHARD_SPLIT_MAX_FRACTION", "0.95")
    out = {}
    for name, halo, gbps in (("halo", "1", "0"), ("halo, bytes charged", "1", "300"), ("all records", "0", "0")):
        monkeypatch.setenv("MRS_SEARCH_HALO", halo)
        monkeypatch.setenv("MRS_STANDIN_GBPS", gbps)
        g = M.Swarm(hi - lo, arith=M.ARITH_FAST)
        g.construct(0, hi - lo, M.model_params("x500", ground_enabled=True))
        g.set_state(0, hi - lo, st["x"][idx], st["v"][idx], st["R"][idx], st["omega"][idx], st["motor_rpm"][idx])
        g.set_input(0, hi - lo, M.POSITION_CMD, cmd[idx])
        g.comm_init_standin(world, rank, n_total, 5.0, width)
        g.tick_sharded_n(bench.DT, 40, True, False, 100.0)
        g.tick_sharded_n(bench.DT, 260, True, False, 100.0)
        g.synchronize()
        s = g.get_state()
        s["f"] = g.get_external_force()
        out[name] = (s, g.search_stats(), g.split_stats()[0], g.comm_info())
        g.comm_destroy()
        del g
    ref = out["halo"][0]
    for name in ("halo, bytes charged", "all records"):
        for k in ("x", "v", "R", "omega", "motor_rpm", "f"):
            assert np.array_equal(out[name][0][k], ref[k]), f"{k}: {name} against halo"
    searches, on_halo, repeats, cap = out["halo"][1]
    assert searches >= 3 and on_halo == searches - 1 and repeats == 0 and cap > 0, out["halo"][1]
    assert out["all records"][1][1:] == (0, 0, 0) and out["all records"][1][0] == searches
    assert out["halo"][2] > 200 and np.abs(ref["f"]).sum() > 0, (out["halo"][2], "no split ticks or no contact")
    assert out["halo"][3]["bytes_per_rebuild"] < out["all records"][3]["bytes_per_rebuild"]
    print("stand-in rank:", out["halo"][1], "split ticks", out["halo"][2], "bytes per search tick", out["halo"][3]["bytes_per_rebuild"], "against",
          out["all records"][3]["bytes_per_rebuild"])
