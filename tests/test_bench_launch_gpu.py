"""The launches bench.py times, put against the oracle DIRECTLY (they were only checked transitively).

bench.py's default workload is BASELINE configs[2] — 100 000 x500 UAVs, ARITH_FAST, mrs_swarm_step_n with many steps — which the
library issues as `mrs_uav_model_step_buf_nt_fast` on TWO streams (half the blocks each).  UAVs are independent inside makeStep,
so a seeded sample of lanes stepped alone by the oracle must agree with the same lanes of the big launch; tolerance = north_star's
1e-6, per UAV (helpers.per_uav_linf) and per field over the swarm."""
import numpy as np
import pytest

import helpers
from helpers import RTOL_NORTH_STAR

pytestmark = pytest.mark.gpu
DT = 0.001


def _bench_inputs(n, workload, seed):
    import bench  # the bench's own generator: same states and commands as the timed run
    return bench.make_inputs(n, workload, seed)


def _oracle_sample(oracle, st, cmd, pick, mode, n_steps):
    m = len(pick)
    o = oracle.OracleSwarm(m)
    o.construct(0, m, helpers.oracle_params("x500", ground_enabled=True))
    for nm in ("set_mixer_params", "set_rate_params", "set_attitude_params", "set_velocity_params", "set_position_params"):
        getattr(o, nm)(0, m)
    o.set_state(0, m, st["x"][pick], st["v"][pick], st["R"][pick], st["omega"][pick], st["motor_rpm"][pick])
    o.set_input(0, m, mode, cmd[pick])
    o.step_n(DT, n_steps, 8)
    out = o.get_state()
    out["imu"], out["pid"] = o.get_imu(), o.get_pid()
    return out


def _gpu_sample(g, pick):
    full = g.get_state()
    out = {k: v[pick] for k, v in full.items()}
    out["imu"], out["pid"] = g.get_imu()[pick], g.get_pid()[pick]
    return out, full


@pytest.mark.parametrize("workload,n,steps", [("actuator", 100_000, 120), ("position", 100_000, 120), ("actuator", 1_000_000, 40),
                                              ("position", 50_000, 120), ("actuator", 4_000_000, 24), ("position", 2_000_000, 24)])
def test_bench_launch_against_oracle(mrs, oracle, workload, n, steps):
    """100 k actuator: *_model_step_buf_nt_fast x 2 streams; 100 k position: mrs_uav_step_buf_fast x 2 streams; 1 M actuator:
    *_model_step_buf_w3_fast x 2 streams; 50 k position: mrs_uav_step_buf_nt_fast on one stream (< 1024 blocks); 4 M actuator and
    2 M position (>= 1.4 GB moved per step, the HBM-streaming sizes of profiles/): the non-temporal two-wave kernels again."""
    M = mrs
    rng = np.random.default_rng(11)
    st, cmd = _bench_inputs(n, workload, seed=3)
    mode = M.ACTUATOR_CMD if workload == "actuator" else M.POSITION_CMD
    g = M.Swarm(n, arith=M.ARITH_FAST)
    g.construct(0, n, M.model_params("x500", ground_enabled=True))
    g.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    g.set_input(0, n, mode, cmd)
    g.step_n(DT, steps)  # >= 4 launches and >= 1024 blocks: the split-stream form bench.py times
    pick = np.sort(rng.choice(n, 2048, replace=False))
    pick[:2], pick[-1] = [0, 63], n - 1
    nb = (n + 63) // 64
    pick[2:6] = [(nb // 2) * 64 - 1, (nb // 2) * 64, (nb // 2) * 64 + 63, n - 2]  # both sides of the two streams' block boundary
    pick = np.unique(pick)
    ref = _oracle_sample(oracle, st, cmd, pick, oracle.ACTUATOR_CMD if workload == "actuator" else oracle.POSITION_CMD, steps)
    got, full = _gpu_sample(g, pick)
    for k in ("x", "v", "R", "omega", "motor_rpm", "imu") + (("pid",) if workload == "position" else ()):
        helpers.assert_close(got[k], ref[k], RTOL_NORTH_STAR, f"{workload} {n}: {k}")
    worst, err = helpers.assert_close_per_uav(got, ref, RTOL_NORTH_STAR, f"{workload} {n}")
    print(f"{workload} {n} UAVs x {steps} steps (FAST, bench launch form): worst UAV {pick[worst]} per-UAV error {err:.2e}")
    assert np.all(np.isfinite(full["x"]))
    assert np.allclose(np.einsum("nij,nik->njk", full["R"], full["R"]), np.eye(3), atol=1e-9)
