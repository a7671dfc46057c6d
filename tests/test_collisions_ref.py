"""Pins the collision semantics of the oracle to the REFERENCE's own kd-tree: oracle/_ref/libref_nanoflann.so is the
reference's include/nanoflann.hpp + KDTreeVectorOfVectorsAdaptor.h compiled unmodified (oracle/Makefile).  CPU only."""
import numpy as np
import pytest

import helpers


@pytest.fixture(scope="module")
def ref(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt library)")
    return oracle


def clouds():
    rng = np.random.default_rng(42)
    yield "random dense", rng.uniform(0, 12, (1500, 3))
    yield "random sparse", rng.uniform(-200, 200, (4000, 3))
    g = np.stack(np.meshgrid(np.arange(20) * 1.2, np.arange(20) * 1.2, [0.0], indexing="ij"), -1).reshape(-1, 3)
    yield "1.2 m grid (d2 = 1.44, 2.88 < 3.0)", g
    yield "400-UAV tmux grid, 4 m pitch", np.stack(np.meshgrid(np.arange(20) * 4.0, np.arange(20) * 4.0, [0.0], indexing="ij"), -1).reshape(-1, 3)
    p = rng.uniform(0, 5, (300, 3))
    p[100:200] = p[:100]  # coincident pairs
    yield "coincident", p
    yield "two points", np.array([[0.0, 0, 0], [1.0, 1.0, 0.99]])
    yield "single point", np.array([[3.0, 4.0, 5.0]])


def test_reference_kdtree_probe(ref):
    """SURVEY §8c probe: 1.2 m grid, centre point has 9 neighbours within squared radius 3.0: d2 in {0, 1.44, 2.88}."""
    g = np.stack(np.meshgrid(np.arange(5) * 1.2, np.arange(5) * 1.2, [0.0], indexing="ij"), -1).reshape(-1, 3)
    off, idx, d2 = ref.ref_radius_neighbours(g, 3.0, 10)
    c = 12
    dd = np.sort(d2[off[c]:off[c + 1]])
    assert len(dd) == 9 and np.allclose(dd, [0] + [1.44] * 4 + [2.88] * 4, atol=1e-12)


@pytest.mark.parametrize("name,pos", list(clouds()), ids=[c[0] for c in clouds()])
def test_neighbour_set_and_forces_match_reference_kdtree(ref, name, pos):
    O = ref
    n = len(pos)
    off, idx, d2 = O.ref_radius_neighbours(pos, 3.0, 10)
    # literal metric of nanoflann L2_Adaptor::evalMetric for 3 dims
    diff = pos[:, None, :] - pos[None, :, :]
    D2 = (0.0 + diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    for i in range(n):
        got = set(idx[off[i]:off[i + 1]].tolist())
        exp = set(np.nonzero(D2[i] < 3.0)[0].tolist())
        assert got == exp, f"{name}: neighbour set of {i}"
        assert np.array_equal(np.sort(d2[off[i]:off[i + 1]]), np.sort(D2[i][sorted(exp)]))
    # handleCollisions on top of the reference's neighbour lists vs the oracle (x500 / t650 mix)
    po, pt = helpers.oracle_params("x500"), helpers.oracle_params("t650")
    half = n // 2
    s = O.OracleSwarm(n)
    if half:
        s.construct(0, half, po, pos[:half], np.zeros(half))
    s.construct(half, n - half, pt, pos[half:], np.zeros(n - half))
    ap = np.array([po.arm_length + po.prop_radius] * half + [pt.arm_length + pt.prop_radius] * (n - half))
    arm = np.array([po.arm_length] * half + [pt.arm_length] * (n - half))
    prop = np.array([po.prop_radius] * half + [pt.prop_radius] * (n - half))
    mass = np.array([po.mass] * half + [pt.mass] * (n - half))
    s.handle_collisions(True, False, 100.0)
    f = s.get_external_force()
    expect = np.zeros((n, 3))
    crashed = np.zeros(n, dtype=np.int32)
    for i in range(n):
        for j, d in sorted(zip(idx[off[i]:off[i + 1]].tolist(), d2[off[i]:off[i + 1]].tolist())):
            if j == i:
                continue
            crit = ((arm[i] + prop[i]) + arm[j]) + prop[j]
            if d < crit:
                crashed[j] = 1
                rel = pos[i] - pos[j]
                nn = np.sqrt((rel[0] * rel[0] + rel[1] * rel[1]) + rel[2] * rel[2])
                if nn > 0:
                    rel = rel / nn
                expect[i] += ((100.0 * rel) * mass[i]) * (mass[j] / (mass[i] + mass[j]))
    assert ap.shape == (n,)
    helpers.assert_close(f, expect, 1e-14, f"{name}: forces")
    s.handle_collisions(False, True, 100.0)
    assert np.array_equal(s.has_crashed(), crashed), f"{name}: crash flags"
    assert np.all(s.get_external_force() == 0)
