"""Shared scenario builders for the parity tests: the same calls are issued to the CPU oracle
(oracle/oracle_swarm.py — test infrastructure) and to the product (mrs_multirotor_simulator_amd.Swarm)."""
import numpy as np

from mrs_multirotor_simulator_amd import airframes
from oracle import oracle_swarm as O

# tolerance of BASELINE.json's north_star: state L-inf <= 1e-6 relative, FP64
RTOL_NORTH_STAR = 1e-6
# what the LITERAL kernel actually achieves against the scalar oracle (libm sin/cos may differ by an ulp)
RTOL_LITERAL = 1e-11
# FAST kernel (FMA contraction, reciprocals) vs oracle over short horizons
RTOL_FAST = 1e-8


def oracle_params(name, **kw):
    p = O.ModelParams()
    O.lib().orc_model_params_default(p)
    airframes.fill_params(p, name, **kw)
    O.lib().orc_calculate_inertia(p)
    O.lib().orc_scale_allocation(p)
    return p


def to_product_params(M, p):
    return M.ModelParams.from_buffer_copy(bytes(p))


from mrs_multirotor_simulator_amd.synthetic import random_rotations, random_state, tilted_rotations  # noqa: E402,F401  (shared with bench.py)


class Pair:
    """An oracle swarm and a product swarm driven in lockstep."""

    def __init__(self, M, n, arith=0):
        self.M, self.n = M, n
        self.o = O.OracleSwarm(n)
        self.g = M.Swarm(n, arith=arith)

    def both(self, name, *a, **kw):
        getattr(self.o, name)(*a, **kw)
        getattr(self.g, name)(*a, **kw)

    def construct(self, first, count, airframe, pos=None, heading=None, **kw):
        po = oracle_params(airframe, **kw)
        self.o.construct(first, count, po, pos, heading)
        self.g.construct(first, count, to_product_params(self.M, po), pos, heading)
        # UavSystemRos init order: controller params set after construction (src/uav_system_ros.cpp:109-157)
        for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
            self.both(nm, first, count)
        return po

    def set_state(self, first, count, st):
        self.both("set_state", first, count, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])

    def step(self, dt, n=1):
        self.o.step_n(dt, n)
        self.g.step_n(dt, n)

    def compare(self, rtol, what="", fields=("x", "v", "v_prev", "R", "omega", "motor_rpm"), mask=None):
        """mask: boolean per UAV — only these rows are compared (campaigns that have to leave diverging closed loops out)"""
        a, b = self.g.get_state(), self.o.get_state()
        a["imu"], b["imu"] = self.g.get_imu(), self.o.get_imu()
        a["pid"], b["pid"] = self.g.get_pid(), self.o.get_pid()
        if mask is not None:
            a, b = {k: v[mask] for k, v in a.items()}, {k: v[mask] for k, v in b.items()}
        worst = 0.0
        for k in tuple(fields) + ("imu", "pid"):
            worst = max(worst, assert_close(a[k], b[k], rtol, f"{what}:{k}"))
        # north_star's measure is the state-vector L-inf of each UAV, not of the swarm: a UAV near the origin must not hide
        # behind the coordinates of one 500 m away
        self.worst_uav = assert_close_per_uav(a, b, rtol, what)
        return worst


def rel_linf(a, b):
    """max |a-b| / max(|b|) over the array (L-inf relative); NaN patterns must match exactly."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), "NaN pattern differs"
    ia, ib = np.isinf(a), np.isinf(b)
    assert np.array_equal(ia, ib) and np.array_equal(a[ia], b[ib]), "inf pattern differs"
    m = ~(na | ia)
    if not m.any():
        return 0.0
    scale = np.max(np.abs(b[m]))
    diff = np.max(np.abs(a[m] - b[m]))
    return 0.0 if diff == 0 else diff / max(scale, 1e-300)


# per-UAV measures -------------------------------------------------------------------------------------------------
STATE_FIELDS = ("x", "v", "R", "omega", "motor_rpm")  # MultirotorModel::State, multirotor_model.hpp:90-98 (v_prev is a copy of v)
# below these magnitudes a field is compared absolutely (a hovering UAV has v = omega = 0 exactly: no relative measure exists there)
FIELD_FLOOR = {"x": 1.0, "v": 1.0, "v_prev": 1.0, "R": 1.0, "omega": 1.0, "motor_rpm": 1000.0, "imu": 9.81, "pid": 1.0, "f": 1.0}


def per_uav_linf(a, b, fields=STATE_FIELDS):
    """Two per-UAV relative L-inf errors of a state dict `a` against the reference `b` (arrays [n, ...]):
      state : max_c |a_c - b_c| / max(||state_i||_inf, 1)   — the whole state vector of UAV i (north_star, literally)
      field : max over fields of  max_c |a_c - b_c| / max(||field_i||_inf, floor_field)   — every field on its own scale (stricter:
              the rpm entries ~4000 do not set the scale for R or omega)
    NaN / inf patterns must match exactly.  Returns (state_err[n], field_err[n])."""
    n = len(np.asarray(b[fields[0]]))
    num_state, den_state, field_err = np.zeros(n), np.ones(n), np.zeros(n)
    for k in fields:
        x, y = np.asarray(a[k], dtype=np.float64).reshape(n, -1), np.asarray(b[k], dtype=np.float64).reshape(n, -1)
        bad = ~np.isfinite(y)
        assert np.array_equal(np.isnan(x), np.isnan(y)), f"{k}: NaN pattern differs"
        assert np.array_equal(x[bad & ~np.isnan(y)], y[bad & ~np.isnan(y)]), f"{k}: inf pattern differs"
        d = np.where(bad, 0.0, np.abs(np.where(bad, 0.0, x) - np.where(bad, 0.0, y))).max(axis=1)
        m = np.where(bad, 0.0, np.abs(y)).max(axis=1)
        num_state, den_state = np.maximum(num_state, d), np.maximum(den_state, m)
        field_err = np.maximum(field_err, d / np.maximum(m, FIELD_FLOOR.get(k, 1.0)))
    return num_state / den_state, field_err


def assert_close_per_uav(a, b, rtol, what="", fields=STATE_FIELDS):
    """every UAV's own state vector within rtol (both measures of per_uav_linf); returns (worst UAV, its field error)"""
    se, fe = per_uav_linf(a, b, fields)
    i = int(np.argmax(fe))
    assert fe[i] <= rtol, (f"{what}: UAV {i} is off by {fe[i]:.3e} > {rtol:.1e} relative to its own fields "
                           f"(state-vector measure {se[i]:.3e}; worst state-vector UAV {int(np.argmax(se))}: {se.max():.3e})")
    assert se.max() <= rtol
    return i, float(fe[i])


def assert_close(a, b, rtol, what=""):
    e = rel_linf(a, b)
    assert e <= rtol, f"{what}: relative L-inf error {e:.3e} > {rtol:.1e}"
    return e
