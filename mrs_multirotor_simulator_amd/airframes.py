"""Airframe constants of the reference's config/uavs/*.yaml (numeric values only) and the
UavSystemRos init sequence that turns them into ModelParams (src/uav_system_ros.cpp:51-103)."""
from . import swarm as _sw

_QUAD_X = [[-0.707, 0.707, 0.707, -0.707], [-0.707, 0.707, -0.707, 0.707], [-1, -1, 1, 1], [1, 1, 1, 1]]

# name: n_motors, mass, arm_length, body_height, motor_time_constant, air_resistance_coeff,
#       force_constant (kf), moment_constant (km), prop_radius, rpm_min, rpm_max, allocation_matrix (4 x n)
AIRFRAMES = {
    "x500": dict(n_motors=4, mass=2.0, arm_length=0.25, body_height=0.1, motor_time_constant=0.03,
                 air_resistance_coeff=0.30, kf=0.00000027087, km=0.07, prop_radius=0.15, rpm_min=1170, rpm_max=7800,
                 allocation=_QUAD_X),
    "a300": dict(n_motors=4, mass=1.21, arm_length=0.15, body_height=0.05, motor_time_constant=0.05,
                 air_resistance_coeff=0.30, kf=0.000000045, km=0.012, prop_radius=0.089, rpm_min=3200, rpm_max=21400,
                 allocation=_QUAD_X),
    "f330": dict(n_motors=4, mass=1.4, arm_length=0.165, body_height=0.07, motor_time_constant=0.03,
                 air_resistance_coeff=0.30, kf=0.000000094268, km=0.07, prop_radius=0.09, rpm_min=1459, rpm_max=9722,
                 allocation=_QUAD_X),
    "f450": dict(n_motors=4, mass=1.7, arm_length=0.225, body_height=0.1, motor_time_constant=0.03,
                 air_resistance_coeff=0.30, kf=0.00000012216, km=0.07, prop_radius=0.11, rpm_min=1360, rpm_max=9068,
                 allocation=_QUAD_X),
    "f550": dict(n_motors=6, mass=2.3, arm_length=0.27, body_height=0.1, motor_time_constant=0.03,
                 air_resistance_coeff=0.30, kf=0.00000012216, km=0.07, prop_radius=0.11, rpm_min=1360, rpm_max=9068,
                 allocation=[[1, -1, -0.5, 0.5, 0.5, -0.5], [0, 0, -0.87, 0.87, -0.87, 0.87], [1, -1, 1, -1, -1, 1],
                             [1, 1, 1, 1, 1, 1]]),
    "naki": dict(n_motors=8, mass=7.5, arm_length=0.20, body_height=0.2, motor_time_constant=0.03,
                 air_resistance_coeff=0.30, kf=0.00000057658, km=0.07, prop_radius=0.13, rpm_min=956, rpm_max=6376,
                 allocation=[[-0.707, 0.707, 0.707, -0.707, 0.707, -0.707, -0.707, 0.707],
                             [-0.707, 0.707, -0.707, 0.707, 0.707, -0.707, 0.707, -0.707],
                             [-1, -1, 1, 1, 1, 1, -1, -1], [1, 1, 1, 1, 1, 1, 1, 1]]),
    "robofly": dict(n_motors=4, mass=0.8, arm_length=0.135, body_height=0.07, motor_time_constant=0.03,
                    air_resistance_coeff=0.30, kf=0.00000000843, km=0.012, prop_radius=0.09, rpm_min=2058, rpm_max=41160,
                    allocation=_QUAD_X),
    "t650": dict(n_motors=4, mass=3.5, arm_length=0.325, body_height=0.15, motor_time_constant=0.03,
                 air_resistance_coeff=0.30, kf=0.00000073385, km=0.07, prop_radius=0.19, rpm_min=875, rpm_max=5832,
                 allocation=_QUAD_X),
}


def fill_params(p, name, g=9.81, ground_enabled=True, ground_z=0.0, takeoff_patch_enabled=False):
    """Fill a ModelParams-shaped ctypes struct the way UavSystemRos does (defaults of
    config/multirotor_simulator.yaml:8,42-49).  Inertia and allocation scaling are left to the caller
    (mrs_calculate_inertia / mrs_scale_allocation or the oracle's twins)."""
    a = AIRFRAMES[name]
    n = a["n_motors"]
    p.n_motors = n
    p.g = g
    p.mass = a["mass"]
    p.kf = a["kf"]
    p.km = a["km"]
    p.prop_radius = a["prop_radius"]
    p.arm_length = a["arm_length"]
    p.body_height = a["body_height"]
    p.motor_time_constant = a["motor_time_constant"]
    p.max_rpm = a["rpm_max"]
    p.min_rpm = a["rpm_min"]
    p.air_resistance_coeff = a["air_resistance_coeff"]
    p.ground_enabled = int(ground_enabled)
    p.ground_z = ground_z
    p.takeoff_patch_enabled = int(takeoff_patch_enabled)
    for i in range(4 * _sw.MAX_MOTORS):
        p.allocation_matrix[i] = 0.0
    for r in range(4):
        for m in range(n):
            p.allocation_matrix[r * _sw.MAX_MOTORS + m] = float(a["allocation"][r][m])
    return p


def model_params(name, **kw):
    """ModelParams for a shipped airframe, ready for Swarm.construct (host-only arithmetic)."""
    p = _sw.default_params()
    fill_params(p, name, **kw)
    L = _sw.load_library()
    _sw._check(L.mrs_calculate_inertia(_sw.C.byref(p)))
    _sw._check(L.mrs_scale_allocation(_sw.C.byref(p)))
    return p
