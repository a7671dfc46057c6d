"""The neighbour lists a search builds, read back through mrs_swarm_debug_neighbour_lists and compared with brute force.

What replaces nanoflann's radiusSearch(3.0) (src/multirotor_simulator.cpp:326, include/nanoflann.hpp:273-311) between two searches
is a list per UAV of every UAV within sqrt(3) + skin of it at search time: the lists must hold EXACTLY those (the squared distance
is evaluated in the reference's own order, ((0 + dx^2) + dy^2) + dz^2), in ascending index, whatever the cells, bucket chains and
tag collisions of the hash underneath look like."""
import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu


def brute_lists(pos, radius2):
    """neighbour sets by the literal metric, blockwise (n up to ~20 000)"""
    n = len(pos)
    out = [None] * n
    usable = np.all(np.isfinite(pos), axis=1)
    for lo in range(0, n, 1024):
        hi = min(n, lo + 1024)
        with np.errstate(invalid="ignore"):
            d0 = pos[lo:hi, None, 0] - pos[None, :, 0]
            d1 = pos[lo:hi, None, 1] - pos[None, :, 1]
            d2 = pos[lo:hi, None, 2] - pos[None, :, 2]
            near = ((0.0 + d0 * d0) + d1 * d1) + d2 * d2 < radius2
        near &= usable[None, :]
        for i in range(lo, hi):
            near[i - lo, i] = False
            out[i] = np.nonzero(near[i - lo])[0] if usable[i] else np.zeros(0, dtype=np.int64)
    return out


def check_lists(M, pos, what):
    n = len(pos)
    g = M.Swarm(n, arith=M.ARITH_LITERAL)
    g.construct(0, n, helpers.to_product_params(M, helpers.oracle_params("x500")), pos, np.zeros(n))
    count, nbr, cap, radius = g.debug_neighbour_lists()
    assert cap == 24 and nbr.shape == (24, n)
    r2 = radius * radius * (1.0 + 1e-9) + 1e-5  # collide.hip LIST_R2
    want = brute_lists(pos, r2)
    longest = max(len(w) for w in want)
    assert longest <= cap, f"{what}: the scenario overflows the lists ({longest} neighbours)"
    bad = []
    for i in range(n):
        got = nbr[:count[i], i].astype(np.int64)
        if len(got) != len(want[i]) or np.any(got != want[i]):
            bad.append((i, got.tolist(), want[i].tolist()))
    assert not bad, f"{what}: {len(bad)} of {n} lists differ, first: {bad[:3]}"
    listed = sum(len(w) for w in want)
    print(f"{what}: {n} UAVs, {listed / n:.2f} listed neighbours per UAV (longest list {longest}), all lists equal brute force")
    return g


def test_lists_of_a_random_swarm(mrs):
    rng = np.random.default_rng(11)
    n = 20000
    side = (n * 30.0) ** (1 / 3)  # 30 m^3 per UAV
    pos = rng.uniform(-side / 2, side / 2, (n, 3))  # negative coordinates: cells below zero
    check_lists(mrs, pos, "random, 30 m^3 per UAV")


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_lists_with_other_lanes_per_uav(mrs, monkeypatch, lanes):
    """the query's other instantiations (three lanes per UAV is the default up to 500 000 UAVs, two beyond; MRS_QUERY_LPU forces one)"""
    monkeypatch.setenv("MRS_QUERY_LPU", str(lanes))
    rng = np.random.default_rng(20 + lanes)
    n = 6000
    side = (n * 12.0) ** (1 / 3)  # 12 m^3 per UAV: chains and lists of a dozen entries
    check_lists(mrs, rng.uniform(-side / 2, side / 2, (n, 3)), f"{lanes} lane(s) per UAV, 12 m^3 per UAV")


def test_lists_of_a_dense_sheet(mrs):
    """pairs 0.55 m apart, 1.9 m between pairs, rows 2.5 m apart: most cells hold two or three UAVs — every bucket is a chain"""
    m = 900
    pos = np.zeros((m, 3))
    for i in range(m):
        pair = i // 2
        pos[i] = (1.9 * (pair % 30) + 0.55 * (i % 2), 2.5 * (pair // 30), 9.0 + 0.02 * (i % 5))
    check_lists(mrs, pos, "dense sheet")
    check_lists(mrs, pos + np.array([0.2681, -0.0313, 0.011]), "dense sheet, shifted")  # the same sheet with other cell alignments


def test_lists_with_cells_on_their_faces_and_far_out(mrs):
    """positions exactly on cell faces (multiples of the 2.25-m edge), pairs just inside / outside the list radius, unusable
    positions (NaN, beyond the position limit), coordinates of 1e5 m"""
    rng = np.random.default_rng(5)
    edge, R = 2.25, np.sqrt(3.0) + 0.5
    pts = []
    for k in range(300):
        c = np.array([edge * rng.integers(-40, 40), edge * rng.integers(-40, 40), edge * rng.integers(0, 20)], dtype=float)
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        pts.append(c)
        pts.append(c + d * (R - 1e-9 if k % 2 else R + 1e-9) * (1.0 if k % 3 else 0.5))
    pts = np.array(pts)
    far = rng.uniform(-1.0, 1.0, (400, 3)) * 12.0 + np.array([1.0e5, -1.0e5, 5.0e4])
    # coordinates a hair below zero: floor() puts them into cell -1, at its far face (found by the 900-UAV sheet of
    # tests/cpp/sharded_tick_test.cpp, whose y drifts to -7e-22)
    hair = rng.uniform(-4, 4, (200, 3))
    hair[:, 1] = np.where(rng.random(200) < 0.5, -6.9e-22, 5.2e-24)
    hair[::7, 0] = -1e-300
    pos = np.concatenate([pts, far, rng.uniform(-20, 20, (300, 3)), hair * [3.0, 1.0, 3.0]])
    pos[7] = np.nan
    pos[11, 2] = np.inf
    check_lists(mrs, pos, "faces, radius edge, far out")


@pytest.mark.parametrize("name", ["dense", "sparse", "grid12", "tmux400"])
def test_lists_hold_what_the_reference_kdtree_returned(mrs, name):
    """tests/golden/nanoflann_radius_sets.npz: outputs of the REFERENCE's own nanoflann radiusSearch(3.0) (made by
    tests/golden/make_golden.py from oracle/_ref).  Every neighbour it returned for UAV i (itself excluded, src/multirotor_simulator.cpp:335)
    must be on the GPU's list of i; what the list holds beyond that lies between sqrt(3) and the list radius."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nanoflann_radius_sets.npz"))
    pts, off, idx, d2 = (g[f"{name}_{k}"] for k in ("points", "offsets", "indices", "d2"))
    n = len(pts)
    s = mrs.Swarm(n)
    s.construct(0, n, mrs.model_params("x500"), pts, np.zeros(n))
    count, nbr, cap, radius = s.debug_neighbour_lists()
    within = brute_lists(pts, radius * radius * (1.0 + 1e-9) + 1e-5)
    extra = over = 0
    for i in range(n):
        ref = sorted(int(j) for j in idx[off[i]:off[i + 1]] if j != i)
        if len(within[i]) > cap:  # a list that does not fit is reported empty (and keeps the library searching every tick: test_golden covers that form)
            assert count[i] == 0
            over += 1
            continue
        got = nbr[:count[i], i].astype(np.int64).tolist()
        assert got == sorted(got), f"{name}: list of {i} not ascending"
        missing = [j for j in ref if j not in got]
        assert not missing, f"{name}: UAV {i}: the kd-tree returned {missing}, the list holds {got}"
        for j in got:
            if j not in ref:
                d = pts[i] - pts[j]
                dd = ((0.0 + d[0] * d[0]) + d[1] * d[1]) + d[2] * d[2]
                assert 3.0 <= dd < radius * radius * (1 + 1e-9) + 1e-5, f"{name}: UAV {i} lists {j} at squared distance {dd}"
                extra += 1
    print(f"{name}: {n} UAVs, every kd-tree neighbour listed; {extra} more entries between sqrt(3) and {radius:.3f} m; {over} UAVs with more than {cap} neighbours")


@pytest.mark.parametrize("crash", [False, True])
def test_search_tick_with_more_contacts_than_the_hit_list_holds(mrs, oracle, crash):
    """clusters of eight UAVs inside half a metre: seven partners in contact each — more than the four hits the query keeps per UAV,
    so the forces of the search tick come from its reference path; ascending-index sums against the oracle (src/multirotor_simulator.cpp:329-358)"""
    rng = np.random.default_rng(77)
    centres = rng.uniform(-30, 30, (40, 3)) + [0, 0, 50]
    pos = np.concatenate([c + rng.uniform(-0.25, 0.25, (8, 3)) for c in centres] + [rng.uniform(-40, 40, (500, 3)) + [0, 0, 50]])
    n = len(pos)
    po = helpers.oracle_params("x500")
    o = oracle.OracleSwarm(n)
    o.construct(0, n, po, pos, np.zeros(n))
    o.handle_collisions(True, crash, 100.0)
    g = mrs.Swarm(n, arith=mrs.ARITH_LITERAL)
    g.construct(0, n, helpers.to_product_params(mrs, po), pos, np.zeros(n))
    g.handle_collisions(True, crash, 100.0)
    if crash:
        assert np.array_equal(g.has_crashed(), o.has_crashed()) and g.has_crashed()[:320].all()
    else:
        fo = o.get_external_force()
        assert (np.abs(fo[:320]).sum(axis=1) > 0).all()
        helpers.assert_close(g.get_external_force(), fo, 1e-12, "forces of eight-UAV clusters")
