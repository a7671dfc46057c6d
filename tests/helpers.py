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

    def compare(self, rtol, what="", fields=("x", "v", "v_prev", "R", "omega", "motor_rpm")):
        a, b = self.g.get_state(), self.o.get_state()
        a["imu"], b["imu"] = self.g.get_imu(), self.o.get_imu()
        a["pid"], b["pid"] = self.g.get_pid(), self.o.get_pid()
        worst = 0.0
        for k in tuple(fields) + ("imu", "pid"):
            worst = max(worst, assert_close(a[k], b[k], rtol, f"{what}:{k}"))
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


def assert_close(a, b, rtol, what=""):
    e = rel_linf(a, b)
    assert e <= rtol, f"{what}: relative L-inf error {e:.3e} > {rtol:.1e}"
    return e
