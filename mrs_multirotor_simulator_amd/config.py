"""Loader for parameter files laid out like the reference's config/ directory (config/multirotor_simulator.yaml,
config/uavs.yaml, config/uavs/<type>.yaml, config/controllers/*.yaml) — the values UavSystemRos reads with
mrs_lib::ParamLoader (src/uav_system_ros.cpp:27-157).  Later files override earlier ones, like the custom-config
layering of the launch file.  Returns plain dictionaries plus ready-to-use ModelParams."""
import yaml

from . import swarm as _sw

CONTROLLER_DEFAULTS = {
    "mixer": {"desaturation": True},
    "rate_controller": {"kp": 4.0, "kd": 0.04, "ki": 0.0},
    "attitude_controller": {"kp": 6.0, "kd": 0.05, "ki": 0.01, "max_rate_roll_pitch": 10.0, "max_rate_yaw": 1.0},
    "velocity_controller": {"kp": 2.0, "kd": 0.05, "ki": 0.01, "max_acceleration": 4.0},
    "position_controller": {"kp": 2.0, "kd": 0.15, "ki": 0.2, "max_velocity": 6.0},
}


def _merge(dst, src):
    for k, v in (src or {}).items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


def load_yaml_files(paths):
    cfg = {}
    for p in paths:
        with open(p) as f:
            _merge(cfg, yaml.safe_load(f))
    return cfg


def model_params_from_config(cfg, uav_type):
    """ModelParams exactly as the UavSystemRos constructor assembles them (src/uav_system_ros.cpp:51-70,96-103)."""
    t = cfg[uav_type]
    pr = t["propulsion"]
    p = _sw.default_params()
    p.n_motors = int(t["n_motors"])
    p.g = float(cfg.get("g", 9.81))
    p.mass = float(t["mass"])
    p.arm_length = float(t["arm_length"])
    p.body_height = float(t["body_height"])
    p.air_resistance_coeff = float(t["air_resistance_coeff"])
    p.motor_time_constant = float(t["motor_time_constant"])
    p.prop_radius = float(pr["prop_radius"])
    p.kf = float(pr["force_constant"])
    p.km = float(pr["moment_constant"])
    p.min_rpm = float(pr["rpm"]["min"])
    p.max_rpm = float(pr["rpm"]["max"])
    p.ground_enabled = int(bool(cfg.get("ground", {}).get("enabled", False)))
    p.ground_z = float(cfg.get("ground", {}).get("z", 0.0))
    p.takeoff_patch_enabled = int(bool(cfg.get("individual_takeoff_platform", {}).get("enabled", False)))
    flat = [float(v) for v in pr["allocation_matrix"]]
    n = p.n_motors
    if len(flat) != 4 * n or n > _sw.MAX_MOTORS:
        raise ValueError(f"{uav_type}: allocation_matrix must have 4 x n_motors entries (n_motors <= {_sw.MAX_MOTORS})")
    for i in range(4 * _sw.MAX_MOTORS):
        p.allocation_matrix[i] = 0.0
    for r in range(4):
        for m in range(n):
            p.allocation_matrix[r * _sw.MAX_MOTORS + m] = flat[r * n + m]  # loadMatrixDynamic2: row-major 4 x n
    L = _sw.load_library()
    _sw._check(L.mrs_calculate_inertia(_sw.C.byref(p)))
    _sw._check(L.mrs_scale_allocation(_sw.C.byref(p)))
    return p


def controller_params_from_config(cfg):
    out = {k: dict(v) for k, v in CONTROLLER_DEFAULTS.items()}
    for k in out:
        _merge(out[k], cfg.get(k, {}))
    return out


_libc = None


def randd(lo, hi):
    """UavSystemRos::randd (src/uav_system_ros.cpp:653-658), quirks included: the span is floor(to - from) and the unit sample goes
    through float; the C library generator, never seeded there (the same sequence as the C++ loader's, include/.../multirotor_simulator.hpp)."""
    global _libc
    import ctypes
    import math

    import numpy as np
    if _libc is None:
        _libc = ctypes.CDLL(None)
        _libc.rand.restype = ctypes.c_int
    zero_to_one = float(np.float32(_libc.rand())) / 2147483647.0  # RAND_MAX of glibc
    return math.floor(hi - lo) * zero_to_one + lo


def uav_spawns_from_config(cfg):
    """[(name, type, x, y, z, heading)] as the UavSystemRos constructors see them: spawn randomisation (four randd draws per UAV, in
    the order x, y, z, heading — src/uav_system_ros.cpp:89-94) applied when `randomization/enabled`."""
    rnd = cfg.get("randomization", {}) or {}
    out = []
    for name in cfg["uav_names"]:
        u = cfg[name]
        sp = u["spawn"]
        x, y, z, h = float(sp["x"]), float(sp["y"]), float(sp["z"]), float(sp["heading"])
        if rnd.get("enabled", False):
            b = rnd["bounds"]
            x += randd(-float(b["x"]), float(b["x"]))
            y += randd(-float(b["y"]), float(b["y"]))
            z += randd(-float(b["z"]), float(b["z"]))
            h += randd(-3.14, 3.14)
        out.append((name, u["type"], x, y, z, h))
    return out


def spawn_swarm_from_config(cfg, device=-1, arith=_sw.ARITH_FAST):
    """The loop of MultirotorSimulator::onInit (src/multirotor_simulator.cpp:150-157) + the UavSystemRos constructor for
    every name in `uav_names`: returns (Swarm, names).  Ends with the two warm-up steps."""
    import numpy as np
    spawns = uav_spawns_from_config(cfg)
    names = [sp[0] for sp in spawns]
    sw = _sw.Swarm(len(names), device=device, arith=arith)
    ctl = controller_params_from_config(cfg)
    cache = {}
    for i, (name, ty, x, y, z, heading) in enumerate(spawns):
        if ty not in cache:
            cache[ty] = model_params_from_config(cfg, ty)
        sw.construct(i, 1, cache[ty], [[x, y, z]], [heading])
    n = len(names)
    sw.set_mixer_params(0, n, bool(ctl["mixer"]["desaturation"]))
    sw.set_rate_params(0, n, **ctl["rate_controller"])
    sw.set_attitude_params(0, n, **ctl["attitude_controller"])
    sw.set_velocity_params(0, n, **ctl["velocity_controller"])
    sw.set_position_params(0, n, **ctl["position_controller"])
    sw.set_input(0, n, _sw.ACTUATOR_CMD, np.zeros((n, _sw.MAX_MOTORS)))
    sw.step_n(0.01, 2)
    return sw, names
