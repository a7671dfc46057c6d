"""UavSystemRos semantics that touch device state: input-timeout fallback commands, set_mass,
set_ground_z, and the construction sequence with its two warm-up steps."""
import math

import numpy as np
import pytest

import helpers

DT = 0.001


def test_oracle_timeout_holds_position_and_heading(oracle):
    O = oracle
    rng = np.random.default_rng(1)
    n = 50
    s = O.OracleSwarm(n)
    s.construct(0, n, helpers.oracle_params("x500"))
    st = helpers.random_state(rng, n, 4, tilted=True)
    s.set_state(0, n, st["x"], st["v"], st["R"], st["omega"], st["motor_rpm"])
    s.set_input(0, n, O.POSITION_CMD, np.concatenate([st["x"] + 5, np.zeros((n, 1))], axis=1))
    s.step_n(DT, 300)
    x_at_timeout = s.get_state()["x"].copy()
    R = s.get_state()["R"]
    hdg = np.arctan2(R[:, 1, 0], R[:, 0, 0])
    s.timeout_input(0, n)
    s.step_n(DT, 6000)
    st2 = s.get_state()
    assert np.allclose(st2["x"], x_at_timeout, atol=0.15)  # the cascade flies back to and holds the timeout position
    h2 = np.arctan2(st2["R"][:, 1, 0], st2["R"][:, 0, 0])
    assert np.allclose(np.angle(np.exp(1j * (h2 - hdg))), 0, atol=0.05)


def test_oracle_set_mass_equals_manual_procedure(oracle):
    O = oracle
    po = helpers.oracle_params("x500", takeoff_patch_enabled=True)
    a, b = O.OracleSwarm(1), O.OracleSwarm(1)
    for s in (a, b):
        s.construct(0, 1, po, [[0, 0, 5.0]], [0.3])
        s.set_position_params(0, 1, 3.0, 0.2, 0.1, 5.0)
        s.set_input(0, 1, O.POSITION_CMD, [[1, 1, 6, 0.5]])
        s.step_n(DT, 50)
    a.set_mass(0, 1, 2.6)
    p2 = b.get_params(0)
    old = p2.mass
    p2.mass = 2.6
    for m in range(4):
        p2.allocation_matrix[2 * 8 + m] = p2.mass * (p2.allocation_matrix[2 * 8 + m] / old)
    O.lib().orc_calculate_inertia(p2)
    b.set_params(0, 1, p2)
    assert bytes(a.get_params(0)) == bytes(b.get_params(0))
    assert np.all(a.get_pid() == 0)
    for s in (a, b):
        s.step_n(DT, 50)
    assert np.array_equal(a.get_state()["x"], b.get_state()["x"])
    a.set_ground_z(0, 1, -3.0)
    assert a.get_params(0).ground_z == -3.0 and a.get_params(0).mass == 2.6


@pytest.mark.gpu
@pytest.mark.parametrize("arith", [0, 1])
def test_gpu_timeout_input_every_mode(mrs, oracle, arith):
    from test_parity_gpu import payload_for
    rng = np.random.default_rng(77)
    per = 64
    modes = list(range(0, 11))
    n = per * len(modes)
    p = helpers.Pair(mrs, n, arith=arith)
    p.construct(0, n, "x500")
    st = helpers.random_state(rng, n, 4, tilted=True)
    p.set_state(0, n, st)
    for k, mode in enumerate(modes):
        sub = {key: val[k * per:(k + 1) * per] for key, val in st.items()}
        p.both("set_input", k * per, per, mode, payload_for(oracle, mode, rng, per, 4, sub))
    p.step(DT, 30)
    p.both("timeout_input", 0, n)
    p.step(DT, 1)
    p.compare(helpers.RTOL_LITERAL if arith == 0 else helpers.RTOL_FAST, "first step on the fallback commands")
    p.step(DT, 60)
    p.compare(helpers.RTOL_LITERAL if arith == 0 else helpers.RTOL_NORTH_STAR, "60 steps on the fallback commands")


@pytest.mark.gpu
def test_gpu_set_mass_and_ground_z(mrs, oracle):
    rng = np.random.default_rng(78)
    n = 256
    p = helpers.Pair(mrs, n)
    p.construct(0, n, "x500", ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=True,
                pos=np.concatenate([rng.uniform(-5, 5, (n, 2)), np.zeros((n, 1))], axis=1), heading=rng.uniform(-3, 3, n))
    thr = np.where(np.arange(n)[:, None] % 2 == 0, 0.9, 0.1) * np.ones((1, 4))
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, thr)
    p.step(DT, 200)  # even UAVs lift off (take-off patch cleared), odd ones stay on their patch: two flag values per type
    p.both("set_mass", 10, 100, 2.6)
    p.both("set_ground_z", 50, 120, -1.0)
    for i in (0, 10, 11, 60, 61, 120, 200):
        assert bytes(p.g.get_params(i)) == bytes(helpers.to_product_params(mrs, p.o.get_params(i))), i
    p.both("set_input", 0, n, oracle.POSITION_CMD, np.concatenate([rng.uniform(-5, 5, (n, 2)), rng.uniform(1, 4, (n, 1)), np.zeros((n, 1))], axis=1))
    p.step(DT, 100)
    p.compare(helpers.RTOL_LITERAL, "after set_mass / set_ground_z")


@pytest.mark.gpu
def test_gpu_uav_system_ros_construction_sequence(mrs, oracle):
    """src/uav_system_ros.cpp:96-157,223-232: inertia, allocation scaling, UavSystem(params, spawn, heading), five
    set*Params, zero actuators, two makeStep(0.01)."""
    n = 100
    rng = np.random.default_rng(3)
    pos = np.concatenate([rng.uniform(-30, 30, (n, 2)), np.zeros((n, 1))], axis=1)
    p = helpers.Pair(mrs, n)
    p.construct(0, n, "t650", pos=pos, heading=rng.uniform(-3.14, 3.14, n), ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False)
    p.both("set_input", 0, n, oracle.ACTUATOR_CMD, np.zeros((n, 4)))
    p.step(0.01, 2)
    p.compare(helpers.RTOL_LITERAL, "warm-up")
    c = math.exp(-0.01 / 0.03)
    assert np.allclose(p.g.get_state()["motor_rpm"][:, :4], 875 * (1 - c) * (1 + c), rtol=1e-14)
    assert np.array_equal(p.g.get_state()["x"], pos)


def test_config_loader_layout(mrs):
    import os
    from mrs_multirotor_simulator_amd import config
    cfg = config.load_yaml_files([os.path.join(os.path.dirname(__file__), "golden", "sample_config.yaml")])
    p = config.model_params_from_config(cfg, "hexa_test")
    assert (p.n_motors, p.mass, p.ground_enabled, p.ground_z, p.takeoff_patch_enabled) == (6, 2.9, 1, 0.5, 0)
    assert p.J[0] == 2.9 * (3.0 * 0.30 * 0.30 + 0.12 * 0.12) / 12.0 and p.J[8] == (2.9 * 0.30 * 0.30) / 2.0
    assert p.allocation_matrix[1 * 8 + 2] == -0.87 * (0.30 * 0.00000014) and p.allocation_matrix[3 * 8 + 5] == 0.00000014
    ctl = config.controller_params_from_config(cfg)
    assert ctl["position_controller"] == {"kp": 2.5, "kd": 0.15, "ki": 0.2, "max_velocity": 4.0}
    assert ctl["rate_controller"]["kp"] == 4.5 and ctl["attitude_controller"]["kp"] == 6.0


@pytest.mark.gpu
def test_gpu_spawn_from_config_matches_oracle(mrs, oracle):
    import os
    from mrs_multirotor_simulator_amd import config
    cfg = config.load_yaml_files([os.path.join(os.path.dirname(__file__), "golden", "sample_config.yaml")])
    sw, names = config.spawn_swarm_from_config(cfg, arith=mrs.ARITH_LITERAL)
    assert names == ["alpha", "bravo", "charlie"]
    O = oracle
    o = O.OracleSwarm(3)
    ctl = config.controller_params_from_config(cfg)
    for i, nm in enumerate(names):
        pm = config.model_params_from_config(cfg, cfg[nm]["type"])
        sp = cfg[nm]["spawn"]
        o.construct(i, 1, O.ModelParams.from_buffer_copy(bytes(pm)), [[sp["x"], sp["y"], sp["z"]]], [sp["heading"]])
    o.set_mixer_params(0, 3, True)
    o.set_rate_params(0, 3, **ctl["rate_controller"])
    o.set_attitude_params(0, 3, **ctl["attitude_controller"])
    o.set_velocity_params(0, 3, **ctl["velocity_controller"])
    o.set_position_params(0, 3, **ctl["position_controller"])
    o.set_input(0, 3, O.ACTUATOR_CMD, np.zeros((3, 8)))
    o.step_n(0.01, 2)
    goal = np.array([[2, 2, 3, 0.1], [-2, 5, 4, -0.5], [5, 0, 2, 2.0]])
    sw.set_input(0, 3, mrs.POSITION_CMD, goal)
    o.set_input(0, 3, O.POSITION_CMD, goal)
    sw.step_n(DT, 500)
    o.step_n(DT, 500)
    a, b = sw.get_state(), o.get_state()
    for k in b:
        helpers.assert_close(a[k], b[k], helpers.RTOL_LITERAL, k)


# ------------------------------------------------------------------------------------------------------------------
# the C++ loader of the same files (include/mrs_multirotor_simulator/config_loader.hpp)
# ------------------------------------------------------------------------------------------------------------------
def _build_config_loader_test(mrs):
    import os, subprocess
    from mrs_multirotor_simulator_amd import swarm
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "config_loader_test")
    libdir = os.path.dirname(swarm.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-DMRS_NO_EIGEN", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "cpp", "config_loader_test.cpp"), "-o", exe, "-L", libdir, "-lmrs_swarm", "-lpthread",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def _cpp_loader_output(exe, files, run=False):
    import subprocess
    out = subprocess.run([exe] + list(files) + (["--run"] if run else []), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = [ln.split() for ln in out.stdout.splitlines()]
    return {"sim": [r for r in rows if r[0] == "SIM"], "uav": [r for r in rows if r[0] == "UAV"], "params": [r for r in rows if r[0] == "PARAMS"],
            "pose": [r for r in rows if r[0] == "POSE"]}


def _check_cpp_against_python_loader(mrs, files):
    from mrs_multirotor_simulator_amd import config
    exe = _build_config_loader_test(mrs)
    got = _cpp_loader_output(exe, files)
    cfg = config.load_yaml_files(files)
    names = list(cfg["uav_names"])
    assert [r[1] for r in got["uav"]] == names
    for r, prm, name in zip(got["uav"], got["params"], names):
        sp = cfg[name]["spawn"]
        assert r[2] == cfg[name]["type"] and [float(v) for v in r[3:7]] == [float(sp[k]) for k in ("x", "y", "z", "heading")]
        p = config.model_params_from_config(cfg, cfg[name]["type"])
        n = p.n_motors
        want = [n, p.g, p.mass, p.kf, p.km, p.prop_radius, p.arm_length, p.body_height, p.motor_time_constant, p.max_rpm, p.min_rpm,
                p.air_resistance_coeff, p.ground_enabled, p.ground_z, p.takeoff_patch_enabled, p.J[0], p.J[4], p.J[8]]
        want += [p.allocation_matrix[rr * 8 + m] for rr in range(4) for m in range(n)]
        assert [float(v) for v in prm[2:]] == [float(v) for v in want], name  # bit-identical: both go through the library's host arithmetic
    return got, cfg


def test_cpp_config_loader_matches_python_loader(mrs):
    import os
    sample = os.path.join(os.path.dirname(__file__), "golden", "sample_config.yaml")
    got, cfg = _check_cpp_against_python_loader(mrs, [sample])
    assert got["sim"][0][1:] == ["100", "100", "1", "1", "1", "100", "1", "1"]  # defaults of the shipped simulator file


def test_cpp_config_loader_reads_the_reference_config_directory(mrs):
    """The reference's own parameter files (only present next to a reference checkout): both loaders must agree on every airframe."""
    import glob, os
    base = "/root/reference/config"
    if not os.path.isdir(base):
        pytest.skip("no reference checkout here")
    files = [os.path.join(base, "multirotor_simulator.yaml"), os.path.join(base, "uavs.yaml")] + sorted(glob.glob(os.path.join(base, "uavs", "*.yaml"))) + \
        sorted(glob.glob(os.path.join(base, "controllers", "*.yaml")))
    got, cfg = _check_cpp_against_python_loader(mrs, files)
    assert got["sim"][0][1:4] == ["100", "100", "1"] and got["uav"][0][1:3] == ["uav1", "x500"]
    # every airframe file parses (the uavs.yaml above only instantiates one of them): feed each through a one-UAV fleet
    import tempfile
    for f in sorted(glob.glob(os.path.join(base, "uavs", "*.yaml"))):
        ty = os.path.splitext(os.path.basename(f))[0]
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as t:
            t.write(f'uav_names: ["u"]\nu:\n  type: "{ty}"\n  spawn: {{x: 0, y: 0, z: 0, heading: 0}}\n')
        try:
            _check_cpp_against_python_loader(mrs, files + [t.name])
        finally:
            os.unlink(t.name)


@pytest.mark.gpu
def test_cpp_swarm_from_config_matches_python_path(mrs):
    """constructSwarmFromConfig (C++) == spawn_swarm_from_config (Python): same launches, same bits after 300 commanded steps."""
    import os
    import numpy as np
    from mrs_multirotor_simulator_amd import config
    sample = os.path.join(os.path.dirname(__file__), "golden", "sample_config.yaml")
    got = _cpp_loader_output(_build_config_loader_test(mrs), [sample], run=True)
    cfg = config.load_yaml_files([sample])
    sw, names = config.spawn_swarm_from_config(cfg, arith=mrs.ARITH_LITERAL)
    cmd = []
    for nm in names:
        sp = cfg[nm]["spawn"]
        cmd.append([sp["x"] + 1.0, sp["y"] - 1.0, sp["z"] + 2.0, 0.3])
    sw.set_input(0, len(names), mrs.POSITION_CMD, np.array(cmd))
    sw.step_n(0.001, 300)
    x = sw.get_state()["x"]
    assert np.array_equal(np.array([[float(v) for v in r[2:5]] for r in got["pose"]]), x)


def test_builtin_airframes_equal_the_reference_parameter_files(mrs):
    """mrs_multirotor_simulator_amd/airframes.py (used by tests and bench) against config/uavs/*.yaml of a reference checkout."""
    import glob, os
    from mrs_multirotor_simulator_amd import airframes, config
    base = "/root/reference/config"
    if not os.path.isdir(base):
        pytest.skip("no reference checkout here")
    files = [os.path.join(base, "multirotor_simulator.yaml")] + sorted(glob.glob(os.path.join(base, "uavs", "*.yaml")))
    cfg = config.load_yaml_files(files)
    names = sorted(os.path.splitext(os.path.basename(f))[0] for f in glob.glob(os.path.join(base, "uavs", "*.yaml")))
    assert names == sorted(airframes.AIRFRAMES)
    for name in names:
        ref = config.model_params_from_config(cfg, name)
        got = airframes.model_params(name, ground_enabled=bool(ref.ground_enabled), ground_z=ref.ground_z,
                                     takeoff_patch_enabled=bool(ref.takeoff_patch_enabled))
        assert bytes(ref) == bytes(got), name


def test_default_controller_gains_equal_the_reference_parameter_files(mrs):
    """The defaults of the five set_*_params calls (= the reference headers' defaults) against config/controllers/*.yaml."""
    import glob, inspect, os
    from mrs_multirotor_simulator_amd import config
    base = "/root/reference/config/controllers"
    if not os.path.isdir(base):
        pytest.skip("no reference checkout here")
    ref = config.load_yaml_files(sorted(glob.glob(os.path.join(base, "*.yaml"))))
    assert ref["mixer"] == config.CONTROLLER_DEFAULTS["mixer"]
    for blk, meth in (("rate_controller", "set_rate_params"), ("attitude_controller", "set_attitude_params"),
                      ("velocity_controller", "set_velocity_params"), ("position_controller", "set_position_params")):
        assert {k: float(v) for k, v in ref[blk].items()} == config.CONTROLLER_DEFAULTS[blk], blk
        sig = inspect.signature(getattr(mrs.Swarm, meth))
        defaults = {k: p.default for k, p in sig.parameters.items() if p.default is not inspect.Parameter.empty}
        assert defaults == config.CONTROLLER_DEFAULTS[blk], meth


def test_cpp_config_loader_reads_the_400_uav_launch_layering(mrs):
    """BASELINE config 2's own parameter file (tmux/standalone_400_uavs/custom_configs/simulator.yaml) layered over the defaults the
    way the launch file does: 400 UAVs, both loaders agree, and the spawn grid is the one the config-2 parity test uses."""
    import glob, os
    import numpy as np
    base = "/root/reference/config"
    custom = "/root/reference/tmux/standalone_400_uavs/custom_configs/simulator.yaml"
    if not os.path.isfile(custom):
        pytest.skip("no reference checkout here")
    files = [os.path.join(base, "multirotor_simulator.yaml"), os.path.join(base, "uavs.yaml")] + sorted(glob.glob(os.path.join(base, "uavs", "*.yaml"))) + \
        sorted(glob.glob(os.path.join(base, "controllers", "*.yaml"))) + [custom]
    got, cfg = _check_cpp_against_python_loader(mrs, files)
    assert len(got["uav"]) == 400 and {r[2] for r in got["uav"]} == {"f550"}
    assert got["sim"][0][4:7] == ["1", "0", "100"]  # collisions enabled, crash off (the custom file), rebounce 100
    xy = np.array([[float(r[3]), float(r[4])] for r in got["uav"]])
    gx, gy = np.meshgrid(np.arange(20) * 4.0, np.arange(20) * 4.0, indexing="ij")
    want = np.stack([gx.ravel(), gy.ravel()], axis=1)
    assert sorted(map(tuple, xy - xy.min(axis=0))) == sorted(map(tuple, want))  # a 20 x 20 grid with 4 m pitch


def test_python_loader_applies_spawn_randomisation_like_the_cpp_loader():
    """`randomization/enabled` (src/uav_system_ros.cpp:89-94): four randd draws per UAV, floor(to - from) span and the float cast
    included; the draws come from the C library generator like the C++ loader's (the Python loader used to ignore it)."""
    import ctypes
    import math
    import os
    from mrs_multirotor_simulator_amd import config
    cfg = config.load_yaml_files([os.path.join(os.path.dirname(__file__), "golden", "sample_config.yaml")])
    plain = config.uav_spawns_from_config(cfg)
    cfg["randomization"] = {"enabled": True, "bounds": {"x": 2.5, "y": 1.0, "z": 0.7}}
    libc = ctypes.CDLL(None)
    libc.srand(1234)
    got = config.uav_spawns_from_config(cfg)
    libc.srand(1234)
    libc.rand.restype = ctypes.c_int
    for (n0, t0, x0, y0, z0, h0), (n1, t1, x1, y1, z1, h1) in zip(plain, got):
        assert (n0, t0) == (n1, t1)
        exp = []
        for base, b in ((x0, 2.5), (y0, 1.0), (z0, 0.7), (h0, 3.14)):
            u = float(np.float32(libc.rand())) / 2147483647.0
            exp.append(base + (math.floor(2 * b) * u - b))  # floor(to - from): bounds 2.5 -> span 5, 1.0 -> 2, 0.7 -> 1 (!), 3.14 -> 6
        assert (x1, y1, z1, h1) == tuple(exp)
    assert any(abs(a[2] - b[2]) > 1e-3 for a, b in zip(plain, got))
