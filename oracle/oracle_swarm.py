"""ctypes binding of the CPU oracle (oracle/liboracle.so) — TEST INFRASTRUCTURE ONLY.

Imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke(); never by the product
package.  Method names mirror mrs_multirotor_simulator_amd.swarm.Swarm so parity tests can drive
both with the same calls.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_MOTORS = 8

(INPUT_UNKNOWN, ACTUATOR_CMD, CONTROL_GROUP_CMD, ATTITUDE_RATE_CMD, ATTITUDE_CMD, TILT_HDG_RATE_CMD,
 ACCELERATION_HDG_RATE_CMD, ACCELERATION_HDG_CMD, VELOCITY_HDG_RATE_CMD, VELOCITY_HDG_CMD, POSITION_CMD) = range(11)
FF_VELOCITY_HDG_RATE, FF_VELOCITY_HDG, FF_ACCELERATION_HDG_RATE, FF_ACCELERATION_HDG = range(4)


class ModelParams(C.Structure):
    _fields_ = [("n_motors", C.c_int32), ("ground_enabled", C.c_int32), ("takeoff_patch_enabled", C.c_int32),
                ("_pad", C.c_int32)] + [(k, C.c_double) for k in (
                    "g", "mass", "kf", "km", "prop_radius", "arm_length", "body_height", "motor_time_constant",
                    "max_rpm", "min_rpm", "air_resistance_coeff", "ground_z")] + [
                        ("J", C.c_double * 9), ("allocation_matrix", C.c_double * (4 * MAX_MOTORS))]


class MixerParams(C.Structure):
    _fields_ = [("desaturation", C.c_int32), ("_pad", C.c_int32)]


class RateParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("kp", "kd", "ki")]


class AttitudeParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("kp", "kd", "ki", "max_rate_roll_pitch", "max_rate_yaw")]


class VelocityParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("kp", "kd", "ki", "max_acceleration")]


class PositionParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("kp", "kd", "ki", "max_velocity")]


class Diag(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("hdg_rate_denom_small", "projected_norm_small", "yaw_rate_not_finite",
                                          "nan_rollback")]


COMP_WIDTHS = {1: (9, 9), 2: (18, 18), 3: (4, 8), 4: (3, 3), 5: (3, 3), 6: (4, 10), 7: (4, 5), 8: (10, 4), 9: (5, 4), 10: (4, 4)}
OUTPUT_DTYPE = np.dtype([("position", "f8", 3), ("orientation", "f8", 4), ("velocity_body", "f8", 3),
                         ("angular_velocity", "f8", 3), ("linear_acceleration", "f8", 3), ("range", "f8")])


def build(force=False):
    """make -C oracle (liboracle.so always; _ref only when /root/reference exists)."""
    so = os.path.join(HERE, "liboracle.so")
    src = [os.path.join(HERE, f) for f in ("uav_oracle.c", "uav_oracle.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(HERE, "_ref", "libref_nanoflann.so")
    ref_pid = os.path.join(HERE, "_ref", "libref_pid.so")
    if os.path.exists("/root/reference/include/nanoflann.hpp") and (force or not os.path.exists(ref) or not os.path.exists(ref_pid)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


_lib = None
_lib_path = None
NATIVE_FLAGS = "-O3 -march=native -ffp-contract=off -fno-fast-math"
PORTABLE_FLAGS = "-O2 -ffp-contract=off -fno-fast-math"


def use_native():
    """bench.py's cpu_baseline leg: build oracle/_native/liboracle_native.so with -O3 -march=native ON THIS MACHINE and make it the
    library behind OracleSwarm (must be called before the first lib() call).  Returns the compiler flags of the library in use."""
    global _lib_path
    assert _lib is None, "use_native() must come before the oracle library is loaded"
    so = os.path.join(HERE, "_native", "liboracle_native.so")
    try:
        subprocess.check_call(["make", "-C", HERE, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _lib_path = so
        return NATIVE_FLAGS
    except (OSError, subprocess.CalledProcessError):
        return PORTABLE_FLAGS


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_lib_path or build())
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        L = _lib
        L.orc_swarm_create.restype = C.c_void_p
        L.orc_swarm_create.argtypes = [C.c_int32]
        L.orc_swarm_destroy.argtypes = [C.c_void_p]
        L.orc_model_params_default.argtypes = [C.POINTER(ModelParams)]
        L.orc_calculate_inertia.argtypes = [C.POINTER(ModelParams)]
        L.orc_scale_allocation.argtypes = [C.POINTER(ModelParams)]
        L.orc_swarm_construct.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(ModelParams), dp, dp]
        L.orc_swarm_set_params.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(ModelParams)]
        L.orc_swarm_get_params.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ModelParams)]
        for nm, ty in (("mixer", MixerParams), ("rate", RateParams), ("attitude", AttitudeParams),
                       ("velocity", VelocityParams), ("position", PositionParams)):
            getattr(L, f"orc_swarm_set_{nm}_params").argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(ty)]
        L.orc_swarm_set_input.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, dp, C.c_int32]
        L.orc_swarm_set_feedforward.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, dp, C.c_int32]
        L.orc_swarm_step.argtypes = [C.c_void_p, C.c_double]
        L.orc_swarm_step_n.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.c_int32]
        L.orc_swarm_handle_collisions.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double]
        L.orc_swarm_apply_force.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dp]
        L.orc_swarm_crash.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.orc_swarm_set_hold.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.orc_swarm_has_crashed.argtypes = [C.c_void_p, C.c_int32, C.c_int32, ip]
        L.orc_swarm_get_state.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [dp] * 6
        L.orc_swarm_set_state.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [dp] * 5
        L.orc_swarm_get_imu.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dp]
        L.orc_swarm_get_external_force.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dp]
        L.orc_swarm_get_pid.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dp]
        L.orc_swarm_set_pid.argtypes = [C.c_void_p, C.c_int32, C.c_int32, dp]
        L.orc_swarm_get_mixer_allocation.argtypes = [C.c_void_p, C.c_int32, dp]
        L.orc_swarm_get_diag.argtypes = [C.c_void_p, C.POINTER(Diag)]
        L.orc_swarm_get_outputs.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.orc_swarm_timeout_input.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.orc_swarm_set_mass.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double]
        L.orc_swarm_set_ground_z.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_double]
        L.orc_swarm_debug_component.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, dp, C.c_int32, dp, C.c_int32, C.c_double]
        L.orc_pid_update.restype = C.c_double
        L.orc_pid_update.argtypes = [C.c_double] * 5 + [dp, dp, C.c_double, C.c_double]
        L.orc_llt_reorth.argtypes = [dp, dp]
        L.orc_inverse3.argtypes = [dp, dp]
        L.orc_inverse_lu.argtypes = [dp, C.c_int, dp]
    return _lib


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _arr(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def default_params():
    p = ModelParams()
    lib().orc_model_params_default(C.byref(p))
    return p


class OracleSwarm:
    """AoS CPU swarm; same surface as the product's Swarm."""

    def __init__(self, n):
        self.n = int(n)
        self._h = C.c_void_p(lib().orc_swarm_create(self.n))

    def __del__(self):
        try:
            if self._h:
                lib().orc_swarm_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def construct(self, first, count, params=None, pos=None, heading=None):
        pos = _arr(pos, (count, 3))
        heading = _arr(heading, (count,))
        lib().orc_swarm_construct(self._h, first, count, C.byref(params) if params is not None else None, _dp(pos),
                                  _dp(heading))

    def set_params(self, first, count, params):
        lib().orc_swarm_set_params(self._h, first, count, C.byref(params))

    def get_params(self, uav):
        p = ModelParams()
        lib().orc_swarm_get_params(self._h, uav, C.byref(p))
        return p

    def set_mixer_params(self, first, count, desaturation=True):
        lib().orc_swarm_set_mixer_params(self._h, first, count, C.byref(MixerParams(int(desaturation), 0)))

    def set_rate_params(self, first, count, kp=4.0, kd=0.04, ki=0.0):
        lib().orc_swarm_set_rate_params(self._h, first, count, C.byref(RateParams(kp, kd, ki)))

    def set_attitude_params(self, first, count, kp=6.0, kd=0.05, ki=0.01, max_rate_roll_pitch=10.0, max_rate_yaw=1.0):
        lib().orc_swarm_set_attitude_params(self._h, first, count,
                                            C.byref(AttitudeParams(kp, kd, ki, max_rate_roll_pitch, max_rate_yaw)))

    def set_velocity_params(self, first, count, kp=2.0, kd=0.05, ki=0.01, max_acceleration=4.0):
        lib().orc_swarm_set_velocity_params(self._h, first, count, C.byref(VelocityParams(kp, kd, ki, max_acceleration)))

    def set_position_params(self, first, count, kp=2.0, kd=0.15, ki=0.2, max_velocity=6.0):
        lib().orc_swarm_set_position_params(self._h, first, count, C.byref(PositionParams(kp, kd, ki, max_velocity)))

    def set_input(self, first, count, mode, payload=None):
        if payload is None:
            lib().orc_swarm_set_input(self._h, first, count, mode, None, 0)
            return
        payload = _arr(payload)
        payload = payload.reshape(count, -1)
        lib().orc_swarm_set_input(self._h, first, count, mode, _dp(payload), payload.shape[1])

    def set_feedforward(self, first, count, kind, payload):
        payload = _arr(payload).reshape(count, 4)
        lib().orc_swarm_set_feedforward(self._h, first, count, kind, _dp(payload), 4)

    def step(self, dt):
        lib().orc_swarm_step(self._h, dt)

    def step_n(self, dt, n_steps, n_threads=1):
        lib().orc_swarm_step_n(self._h, dt, n_steps, n_threads)

    def handle_collisions(self, enabled, crash, rebounce):
        lib().orc_swarm_handle_collisions(self._h, int(enabled), int(crash), float(rebounce))

    def apply_force(self, first, count, force):
        force = _arr(force, (count, 3))
        lib().orc_swarm_apply_force(self._h, first, count, _dp(force))

    def crash(self, first, count):
        lib().orc_swarm_crash(self._h, first, count)

    def set_hold(self, first, count, hold):
        lib().orc_swarm_set_hold(self._h, first, count, int(bool(hold)))

    def has_crashed(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros(count, dtype=np.int32)
        lib().orc_swarm_has_crashed(self._h, first, count, out.ctypes.data_as(C.POINTER(C.c_int32)))
        return out

    def get_state(self, first=0, count=None):
        count = self.n - first if count is None else count
        st = dict(x=np.zeros((count, 3)), v=np.zeros((count, 3)), v_prev=np.zeros((count, 3)), R=np.zeros((count, 3, 3)),
                  omega=np.zeros((count, 3)), motor_rpm=np.zeros((count, MAX_MOTORS)))
        lib().orc_swarm_get_state(self._h, first, count, *[_dp(st[k]) for k in ("x", "v", "v_prev", "R", "omega", "motor_rpm")])
        return st

    def set_state(self, first, count, x=None, v=None, R=None, omega=None, motor_rpm=None):
        x, v, omega = _arr(x, (count, 3)), _arr(v, (count, 3)), _arr(omega, (count, 3))
        R = _arr(R, (count, 9))
        motor_rpm = _arr(motor_rpm, (count, MAX_MOTORS))
        lib().orc_swarm_set_state(self._h, first, count, _dp(x), _dp(v), _dp(R), _dp(omega), _dp(motor_rpm))

    def get_imu(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros((count, 3))
        lib().orc_swarm_get_imu(self._h, first, count, _dp(out))
        return out

    def get_external_force(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros((count, 3))
        lib().orc_swarm_get_external_force(self._h, first, count, _dp(out))
        return out

    def get_pid(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros((count, 24))
        lib().orc_swarm_get_pid(self._h, first, count, _dp(out))
        return out

    def set_pid(self, first, count, pid):
        lib().orc_swarm_set_pid(self._h, first, count, _dp(_arr(pid, (count, 24))))

    def get_mixer_allocation(self, uav):
        n = self.get_params(uav).n_motors
        out = np.zeros((n, 4))
        lib().orc_swarm_get_mixer_allocation(self._h, uav, _dp(out))
        return out

    def timeout_input(self, first, count):
        lib().orc_swarm_timeout_input(self._h, first, count)

    def set_mass(self, first, count, mass):
        lib().orc_swarm_set_mass(self._h, first, count, float(mass))

    def set_ground_z(self, first, count, ground_z):
        lib().orc_swarm_set_ground_z(self._h, first, count, float(ground_z))

    def get_outputs(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros(count, dtype=OUTPUT_DTYPE)
        lib().orc_swarm_get_outputs(self._h, first, count, out.ctypes.data_as(C.c_void_p))
        return out

    def debug_component(self, component, first, count, rows, dt=0.001):
        wi, wo = COMP_WIDTHS[component]
        rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(count, wi)
        out = np.zeros((count, wo))
        lib().orc_swarm_debug_component(self._h, int(component), int(first), int(count), _dp(rows), wi, _dp(out), wo, float(dt))
        return out

    def get_diag(self):
        d = Diag()
        lib().orc_swarm_get_diag(self._h, C.byref(d))
        return {k: int(getattr(d, k)) for k, _ in Diag._fields_}


# ---- reference nanoflann (oracle/_ref) ----
_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        build()
        path = os.path.join(HERE, "_ref", "libref_nanoflann.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
        _ref.ref_nf_radius_all.restype = C.c_int64
        _ref.ref_nf_radius_all.argtypes = [C.POINTER(C.c_double), C.c_int32, C.c_double, C.c_int32,
                                           C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.c_int64]
        _ref.ref_nf_build_and_count.restype = C.c_int64
        _ref.ref_nf_build_and_count.argtypes = [C.POINTER(C.c_double), C.c_int32, C.c_double, C.c_int32]
    return _ref


_ref_pid = None


def ref_pid_lib():
    """The REFERENCE's own PIDController (controllers/pid.hpp compiled where it lies into oracle/_ref/libref_pid.so), or None."""
    global _ref_pid
    if _ref_pid is None:
        build()
        path = os.path.join(HERE, "_ref", "libref_pid.so")
        if not os.path.exists(path):
            return None
        L = C.CDLL(path)
        L.ref_pid_create.restype = C.c_void_p
        L.ref_pid_destroy.argtypes = [C.c_void_p]
        L.ref_pid_reset.argtypes = [C.c_void_p]
        L.ref_pid_set_params.argtypes = [C.c_void_p] + [C.c_double] * 5
        L.ref_pid_set_saturation.argtypes = [C.c_void_p, C.c_double]
        L.ref_pid_update.restype = C.c_double
        L.ref_pid_update.argtypes = [C.c_void_p, C.c_double, C.c_double]
        _ref_pid = L
    return _ref_pid


def ref_radius_neighbours(pts, radius=3.0, leaf_max_size=10):
    """Reference kd-tree radius search for every point: returns (offsets[n+1], idx, d2)."""
    L = ref_lib()
    pts = _arr(pts).reshape(-1, 3)
    n = pts.shape[0]
    offsets = np.zeros(n + 1, dtype=np.int64)
    total = L.ref_nf_radius_all(_dp(pts), n, radius, leaf_max_size, offsets.ctypes.data_as(C.POINTER(C.c_int64)), None,
                                None, 0)
    idx = np.zeros(total, dtype=np.int32)
    d2 = np.zeros(total)
    L.ref_nf_radius_all(_dp(pts), n, radius, leaf_max_size, offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                        idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(d2), total)
    return offsets, idx, d2
