"""ctypes mirror of include/mrs_swarm.h: `Swarm` is a whole fleet of the reference's UavSystem objects
resident on one MI355X.  Method names follow the reference API (setInput -> set_input, makeStep -> step, ...).
There is no CPU fallback: a missing libmrs_swarm.so or a missing GPU raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MRS_SWARM_LIB: alternative build of the same library (kernel tuning sweeps, tools/tune_step.py)
LIB_PATH = os.environ.get("MRS_SWARM_LIB") or os.path.join(HERE, "libmrs_swarm.so")
MAX_MOTORS = 8

(INPUT_UNKNOWN, ACTUATOR_CMD, CONTROL_GROUP_CMD, ATTITUDE_RATE_CMD, ATTITUDE_CMD, TILT_HDG_RATE_CMD,
 ACCELERATION_HDG_RATE_CMD, ACCELERATION_HDG_CMD, VELOCITY_HDG_RATE_CMD, VELOCITY_HDG_CMD, POSITION_CMD) = range(11)
FF_VELOCITY_HDG_RATE, FF_VELOCITY_HDG, FF_ACCELERATION_HDG_RATE, FF_ACCELERATION_HDG = range(4)
ARITH_LITERAL, ARITH_FAST = 0, 1


class MrsError(RuntimeError):
    pass


class ModelParams(C.Structure):
    """mrs_model_params_t == MultirotorModel::ModelParams (multirotor_model.hpp:24-88)."""
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


class UavOutput(C.Structure):
    """mrs_uav_output_t: what UavSystemRos publishes per UAV and tick."""
    _fields_ = [("position", C.c_double * 3), ("orientation", C.c_double * 4), ("velocity_body", C.c_double * 3),
                ("angular_velocity", C.c_double * 3), ("linear_acceleration", C.c_double * 3), ("range", C.c_double)]


OUTPUT_DTYPE = np.dtype([("position", "f8", 3), ("orientation", "f8", 4), ("velocity_body", "f8", 3),
                         ("angular_velocity", "f8", 3), ("linear_acceleration", "f8", 3), ("range", "f8")])


class CommInfo(C.Structure):
    """mrs_comm_info_t"""
    _fields_ = [("world", C.c_int32), ("rank", C.c_int32), ("rccl_ranks", C.c_int32), ("exchange", C.c_int32), ("n_total", C.c_int64),
                ("bytes_per_tick", C.c_int64), ("bytes_per_rebuild", C.c_int64), ("export_count", C.c_int64), ("export_capacity", C.c_int64),
                ("ticks", C.c_int64), ("searches", C.c_int64), ("noop_ticks", C.c_int64)]


EXCHANGE_NONE, EXCHANGE_FULL_GATHER, EXCHANGE_EXPORT_SETS = 0, 1, 2
# mrs_swarm_debug_component: component id -> (input width, output width)
(COMP_REORTH, COMP_MODEL_RHS, COMP_MIXER, COMP_POSITION, COMP_VELOCITY, COMP_ACCELERATION_HDG, COMP_ACCELERATION_HDG_RATE, COMP_ATTITUDE,
 COMP_TILT_HDG_RATE, COMP_RATE) = range(1, 11)
COMP_WIDTHS = {1: (9, 9), 2: (18, 18), 3: (4, 8), 4: (3, 3), 5: (3, 3), 6: (4, 10), 7: (4, 5), 8: (10, 4), 9: (5, 4), 10: (4, 4)}


EXCHANGE_NAMES = {0: "none", 1: "full all-gather of 48-B records per tick", 2: "export-set all-gather (boundary UAVs), full gather on search ticks"}


class Diag(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("hdg_rate_denom_small", "projected_norm_small", "yaw_rate_not_finite",
                                          "nan_rollback")]


# every symbol include/mrs_swarm.h declares
ABI_SYMBOLS = [
    "mrs_model_params_default", "mrs_calculate_inertia", "mrs_scale_allocation", "mrs_swarm_create",
    "mrs_swarm_destroy", "mrs_swarm_size", "mrs_swarm_set_arith", "mrs_swarm_stream", "mrs_swarm_synchronize",
    "mrs_last_error", "mrs_swarm_construct", "mrs_swarm_set_params", "mrs_swarm_get_params",
    "mrs_swarm_set_mixer_params", "mrs_swarm_set_rate_params", "mrs_swarm_set_attitude_params",
    "mrs_swarm_set_velocity_params", "mrs_swarm_set_position_params", "mrs_swarm_get_mixer_allocation",
    "mrs_swarm_set_input", "mrs_swarm_set_feedforward", "mrs_swarm_apply_force", "mrs_swarm_crash",
    "mrs_swarm_has_crashed", "mrs_swarm_step", "mrs_swarm_step_n", "mrs_swarm_handle_collisions", "mrs_swarm_tick_n",
    "mrs_swarm_get_state", "mrs_swarm_set_state", "mrs_swarm_get_imu", "mrs_swarm_get_external_force",
    "mrs_swarm_get_pid", "mrs_swarm_get_diag", "mrs_swarm_get_outputs", "mrs_swarm_timeout_input", "mrs_swarm_set_mass", "mrs_swarm_set_ground_z", "mrs_swarm_pack_positions", "mrs_swarm_pack_positions_to", "mrs_swarm_handle_collisions_gathered",
    "mrs_debug_pid_sequences", "mrs_swarm_debug_collision_words", "mrs_rccl_unique_id", "mrs_swarm_comm_init", "mrs_swarm_tick_sharded_n", "mrs_swarm_comm_destroy", "mrs_swarm_comm_info",
    "mrs_swarm_comm_init_custom", "mrs_loopback_group_create", "mrs_loopback_group_destroy", "mrs_swarm_comm_init_loopback", "mrs_swarm_set_exchange",
    "mrs_slab_partition", "mrs_swarm_get_fused_stats", "mrs_swarm_debug_component", "mrs_debug_pid_update", "mrs_swarm_set_state_pos", "mrs_swarm_set_pid", "mrs_swarm_clone",
    "mrs_loopback_group_set_rendezvous", "mrs_swarm_debug_chaos", "mrs_swarm_get_split_stats", "mrs_swarm_get_search_stats", "mrs_debug_stream_delay", "mrs_swarm_comm_init_standin",
    "mrs_swarm_peer_window_create", "mrs_swarm_comm_init_peer",
    "mrs_swarm_set_hold", "mrs_swarm_get_collision_stats", "mrs_swarm_get_outputs_view", "mrs_swarm_input_staging", "mrs_swarm_commit_input", "mrs_swarm_last_step_kernel_ms", "mrs_swarm_set_profiling",
    "mrs_swarm_debug_search_ms", "mrs_swarm_debug_neighbour_lists", "mrs_swarm_clone_resized", "mrs_swarm_copy_uavs", "mrs_swarm_step_range", "mrs_swarm_get_states",
    "mrs_swarm_get_outputs_async", "mrs_swarm_outputs_wait", "mrs_cell_order",
]

STATE_DTYPE = np.dtype([("x", "f8", 3), ("v", "f8", 3), ("v_prev", "f8", 3), ("R", "f8", (3, 3)), ("omega", "f8", 3), ("motor_rpm", "f8", 8),
                        ("imu_acceleration", "f8", 3), ("crashed", "i4"), ("n_motors", "i4")])

_lib = None
# mrs_allgather_fn: int (*)(void* user, const void* send, void* recv, uint64_t bytes_per_rank, void* stream)
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


def _preload_hip_runtime():
    """One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (RPATH $ORIGIN).  If
    libmrs_swarm.so pulled in /opt/rocm's copy first and torch was imported later, two runtimes would coexist and
    the second one sees no device.  So when torch is installed, its runtime is loaded first and libmrs_swarm.so
    (same SONAME) binds to it; without torch the system ROCm runtime is used."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamd_comgr.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def default_librccl_path():
    """The librccl.so that belongs to the HIP runtime this process uses: torch's bundled copy when torch is installed (see
    _preload_hip_runtime), otherwise None = the system one from the loader path."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "librccl.so")
    return path if os.path.exists(path) else None


def debug_pid_sequences(arith, params, err, dt, event, new_sat, device=-1):
    """The kernels' PID device function over [n_seq][n_steps] sequences on the GPU (test hook)."""
    load_library()
    params, err, dt, event, new_sat = (np.ascontiguousarray(a, dtype=np.float64) for a in (params, err, dt, event, new_sat))
    out = np.zeros_like(err)
    _check(_lib.mrs_debug_pid_sequences(device, arith, err.shape[0], err.shape[1], _dp(params), _dp(err), _dp(dt), _dp(event),
                                        _dp(new_sat), _dp(out)))
    return out


def rccl_unique_id(librccl_path=None):
    """128-byte communicator id, created on rank 0 and handed to the other ranks over any host channel."""
    load_library()
    path = default_librccl_path() if librccl_path is None else librccl_path
    buf = (C.c_uint8 * 128)()
    _check(_lib.mrs_rccl_unique_id(path.encode() if path else None, C.cast(buf, C.c_void_p)))
    return bytes(buf)


def slab_partition(pos, world):
    """mrs_slab_partition: public indices sorted by x; rank r of `world` equal-count slabs holds order[lo_r:hi_r] (sharded.shard_range)."""
    load_library()
    pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
    order = np.zeros(len(pos), dtype=np.int64)
    _check(_lib.mrs_slab_partition(_dp(pos), len(pos), int(world), order.ctypes.data_as(C.POINTER(C.c_int64))))
    return order


def cell_order(pos, cell=0.0):
    """mrs_cell_order: a spawn order that follows space (Morton key of the neighbour-list cells); order[k] = caller's index spawned k-th"""
    load_library()
    pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
    order = np.zeros(len(pos), dtype=np.int64)
    _check(_lib.mrs_cell_order(_dp(pos), len(pos), float(cell), order.ctypes.data_as(C.POINTER(C.c_int64))))
    return order


class LoopbackGroup:
    """mrs_loopback_group_t: `world` swarms of this process exchanging through device-to-device copies; every swarm must be driven by
    its own host thread (tick_sharded_n is collective)."""

    def __init__(self, world):
        load_library()
        self._h = C.c_void_p()
        self.world = int(world)
        _check(_lib.mrs_loopback_group_create(self.world, C.byref(self._h)))

    def set_rendezvous(self, on=True):
        """the all-gather without host barriers: a rank waits only for its peers to ARRIVE at the same collective"""
        _check(_lib.mrs_loopback_group_set_rendezvous(self._h, int(bool(on))))

    def close(self):
        if getattr(self, "_h", None):
            _lib.mrs_loopback_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_library():
    """dlopen the in-tree libmrs_swarm.so (built by mrs_multirotor_simulator_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MrsError(f"{LIB_PATH} is missing: run `python -m mrs_multirotor_simulator_amd.build` "
                       "(the HIP extension is the only implementation; there is no CPU fallback)")
    _preload_hip_runtime()
    L = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_void_p
    i32, f64 = C.c_int32, C.c_double
    sig = {
        "mrs_model_params_default": [C.POINTER(ModelParams)],
        "mrs_calculate_inertia": [C.POINTER(ModelParams)],
        "mrs_scale_allocation": [C.POINTER(ModelParams)],
        "mrs_swarm_create": [i32, i32, C.POINTER(vp)],
        "mrs_swarm_destroy": [vp],
        "mrs_swarm_size": [vp, ip],
        "mrs_swarm_set_arith": [vp, i32],
        "mrs_swarm_stream": [vp, C.POINTER(vp)],
        "mrs_swarm_synchronize": [vp],
        "mrs_swarm_construct": [vp, i32, i32, C.POINTER(ModelParams), dp, dp],
        "mrs_swarm_set_params": [vp, i32, i32, C.POINTER(ModelParams)],
        "mrs_swarm_get_params": [vp, i32, C.POINTER(ModelParams)],
        "mrs_swarm_set_mixer_params": [vp, i32, i32, C.POINTER(MixerParams)],
        "mrs_swarm_set_rate_params": [vp, i32, i32, C.POINTER(RateParams)],
        "mrs_swarm_set_attitude_params": [vp, i32, i32, C.POINTER(AttitudeParams)],
        "mrs_swarm_set_velocity_params": [vp, i32, i32, C.POINTER(VelocityParams)],
        "mrs_swarm_set_position_params": [vp, i32, i32, C.POINTER(PositionParams)],
        "mrs_swarm_get_mixer_allocation": [vp, i32, dp],
        "mrs_swarm_set_input": [vp, i32, i32, i32, dp, i32],
        "mrs_swarm_set_feedforward": [vp, i32, i32, i32, dp, i32],
        "mrs_swarm_apply_force": [vp, i32, i32, dp],
        "mrs_swarm_crash": [vp, i32, i32],
        "mrs_swarm_has_crashed": [vp, i32, i32, ip],
        "mrs_swarm_step": [vp, f64],
        "mrs_swarm_step_n": [vp, f64, i32, i32],
        "mrs_swarm_handle_collisions": [vp, i32, i32, f64],
        "mrs_swarm_tick_n": [vp, f64, i32, i32, i32, f64],
        "mrs_swarm_get_state": [vp, i32, i32, dp, dp, dp, dp, dp, dp],
        "mrs_swarm_set_state": [vp, i32, i32, dp, dp, dp, dp, dp],
        "mrs_swarm_get_imu": [vp, i32, i32, dp],
        "mrs_swarm_get_external_force": [vp, i32, i32, dp],
        "mrs_swarm_get_pid": [vp, i32, i32, dp],
        "mrs_swarm_get_diag": [vp, C.POINTER(Diag)],
        "mrs_swarm_get_outputs": [vp, i32, i32, vp],
        "mrs_swarm_timeout_input": [vp, i32, i32],
        "mrs_swarm_set_mass": [vp, i32, i32, f64],
        "mrs_swarm_set_ground_z": [vp, i32, i32, f64],
        "mrs_swarm_pack_positions": [vp, C.POINTER(vp), C.POINTER(C.c_int64)],
        "mrs_swarm_pack_positions_to": [vp, vp],
        "mrs_swarm_handle_collisions_gathered": [vp, vp, C.c_int64, C.c_int64, i32, i32, f64],
        "mrs_debug_pid_sequences": [i32, i32, i32, i32, dp, dp, dp, dp, dp, dp],
        "mrs_swarm_debug_collision_words": [vp, C.POINTER(C.c_uint32)],
        "mrs_rccl_unique_id": [C.c_char_p, vp],
        "mrs_swarm_comm_init": [vp, C.c_char_p, i32, i32, vp, C.c_int64],
        "mrs_swarm_tick_sharded_n": [vp, f64, i32, i32, i32, f64],
        "mrs_swarm_comm_destroy": [vp],
        "mrs_swarm_comm_info": [vp, C.POINTER(CommInfo)],
        "mrs_swarm_comm_init_custom": [vp, i32, i32, C.c_int64, ALLGATHER_FN, vp],
        "mrs_loopback_group_create": [i32, C.POINTER(vp)],
        "mrs_loopback_group_destroy": [vp],
        "mrs_swarm_comm_init_loopback": [vp, vp, i32, C.c_int64],
        "mrs_loopback_group_set_rendezvous": [vp, i32],
        "mrs_swarm_debug_chaos": [vp, i32, C.c_uint64],
        "mrs_swarm_get_split_stats": [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
        "mrs_swarm_get_search_stats": [vp] + [C.POINTER(C.c_int64)] * 4,
        "mrs_debug_stream_delay": [vp, C.c_double],
        "mrs_swarm_comm_init_standin": [vp, i32, i32, C.c_int64, C.c_double, C.c_double],
        "mrs_swarm_peer_window_create": [vp, i32, i32, C.c_int64, C.POINTER(C.c_void_p), C.c_char_p],
        "mrs_swarm_comm_init_peer": [vp, C.POINTER(C.c_void_p), C.c_char_p],
        "mrs_swarm_set_exchange": [vp, i32],
        "mrs_slab_partition": [dp, C.c_int64, i32, C.POINTER(C.c_int64)],
        "mrs_swarm_get_fused_stats": [vp] + [C.POINTER(C.c_int64)] * 4,
        "mrs_swarm_debug_component": [vp, i32, i32, i32, dp, i32, dp, i32, f64],
        "mrs_debug_pid_update": [i32, i32, i32, dp, dp, dp, dp, dp],
        "mrs_swarm_set_state_pos": [vp, i32, i32, dp, dp],
        "mrs_swarm_set_pid": [vp, i32, i32, dp],
        "mrs_swarm_clone": [vp, C.POINTER(vp)],
        "mrs_swarm_set_hold": [vp, i32, i32, i32],
        "mrs_swarm_get_outputs_view": [vp, i32, i32, C.POINTER(vp)],
        "mrs_swarm_input_staging": [vp, i32, i32, C.POINTER(dp)],
        "mrs_swarm_commit_input": [vp, i32, i32, i32, i32],
        "mrs_swarm_get_collision_stats": [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
        "mrs_swarm_last_step_kernel_ms": [vp, dp, ip],
        "mrs_swarm_set_profiling": [vp, i32],
        "mrs_swarm_debug_search_ms": [vp, i32, i32, f64, dp],
        "mrs_swarm_debug_neighbour_lists": [vp, i32, f64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), i32, C.POINTER(C.c_int32), dp],
        "mrs_swarm_clone_resized": [vp, i32, C.POINTER(vp)],
        "mrs_swarm_copy_uavs": [vp, i32, vp, i32, i32],
        "mrs_swarm_step_range": [vp, i32, i32, f64],
        "mrs_swarm_get_states": [vp, i32, i32, vp],
        "mrs_swarm_get_outputs_async": [vp, i32, i32, ip],
        "mrs_swarm_outputs_wait": [vp, i32, C.POINTER(vp), ip],
        "mrs_cell_order": [dp, C.c_int64, f64, C.POINTER(C.c_int64)],
    }
    for name, args in sig.items():
        if os.environ.get("MRS_SWARM_LIB") and not hasattr(L, name):
            continue  # (an older library given for a same-box A/B measurement: the call fails when used, not the import)
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.mrs_last_error.restype = C.c_char_p
    L.mrs_last_error.argtypes = []
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise MrsError(f"libmrs_swarm error {rc}: {load_library().mrs_last_error().decode()}")


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _arr(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(shape) if shape is not None else a


def default_params():
    """ModelParams() of the reference: x500 defaults (multirotor_model.hpp:26-66)."""
    p = ModelParams()
    _check(load_library().mrs_model_params_default(C.byref(p)))
    return p


class Swarm:
    """n UavSystem objects on one GPU (SoA FP64 state in HBM); every method maps to one C-ABI call."""

    def __init__(self, n, device=-1, arith=ARITH_LITERAL):
        self._h = C.c_void_p()
        self.n = int(n)
        _check(load_library().mrs_swarm_create(self.n, device, C.byref(self._h)))
        self.set_arith(arith)

    def close(self):
        if getattr(self, "_h", None):
            load_library().mrs_swarm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters --
    def set_arith(self, arith):
        _check(_lib.mrs_swarm_set_arith(self._h, arith))

    def construct(self, first, count, params=None, pos=None, heading=None):
        pos, heading = _arr(pos, (count, 3)), _arr(heading, (count,))
        _check(_lib.mrs_swarm_construct(self._h, first, count, C.byref(params) if params is not None else None, _dp(pos),
                                        _dp(heading)))

    def set_params(self, first, count, params):
        _check(_lib.mrs_swarm_set_params(self._h, first, count, C.byref(params)))

    def get_params(self, uav):
        p = ModelParams()
        _check(_lib.mrs_swarm_get_params(self._h, uav, C.byref(p)))
        return p

    def set_mixer_params(self, first, count, desaturation=True):
        _check(_lib.mrs_swarm_set_mixer_params(self._h, first, count, C.byref(MixerParams(int(desaturation), 0))))

    def set_rate_params(self, first, count, kp=4.0, kd=0.04, ki=0.0):
        _check(_lib.mrs_swarm_set_rate_params(self._h, first, count, C.byref(RateParams(kp, kd, ki))))

    def set_attitude_params(self, first, count, kp=6.0, kd=0.05, ki=0.01, max_rate_roll_pitch=10.0, max_rate_yaw=1.0):
        _check(_lib.mrs_swarm_set_attitude_params(self._h, first, count,
                                                  C.byref(AttitudeParams(kp, kd, ki, max_rate_roll_pitch, max_rate_yaw))))

    def set_velocity_params(self, first, count, kp=2.0, kd=0.05, ki=0.01, max_acceleration=4.0):
        _check(_lib.mrs_swarm_set_velocity_params(self._h, first, count, C.byref(VelocityParams(kp, kd, ki, max_acceleration))))

    def set_position_params(self, first, count, kp=2.0, kd=0.15, ki=0.2, max_velocity=6.0):
        _check(_lib.mrs_swarm_set_position_params(self._h, first, count, C.byref(PositionParams(kp, kd, ki, max_velocity))))

    def get_mixer_allocation(self, uav):
        out = np.zeros((self.get_params(uav).n_motors, 4))
        _check(_lib.mrs_swarm_get_mixer_allocation(self._h, uav, _dp(out)))
        return out

    # -- commands --
    def set_input(self, first, count, mode, payload=None):
        if payload is None:
            _check(_lib.mrs_swarm_set_input(self._h, first, count, mode, None, 0))
            return
        payload = _arr(payload).reshape(count, -1)
        _check(_lib.mrs_swarm_set_input(self._h, first, count, mode, _dp(payload), payload.shape[1]))

    def set_feedforward(self, first, count, kind, payload):
        payload = _arr(payload).reshape(count, 4)
        _check(_lib.mrs_swarm_set_feedforward(self._h, first, count, kind, _dp(payload), 4))

    def apply_force(self, first, count, force):
        _check(_lib.mrs_swarm_apply_force(self._h, first, count, _dp(_arr(force, (count, 3)))))

    def crash(self, first, count):
        _check(_lib.mrs_swarm_crash(self._h, first, count))

    def has_crashed(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.zeros(count, dtype=np.int32)
        _check(_lib.mrs_swarm_has_crashed(self._h, first, count, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    # -- hot path --
    def step(self, dt):
        _check(_lib.mrs_swarm_step(self._h, dt))

    def step_n(self, dt, n_steps, substeps_per_launch=1):
        _check(_lib.mrs_swarm_step_n(self._h, dt, n_steps, substeps_per_launch))

    def handle_collisions(self, enabled, crash, rebounce):
        _check(_lib.mrs_swarm_handle_collisions(self._h, int(enabled), int(crash), float(rebounce)))

    def tick_n(self, dt, n_ticks, enabled, crash, rebounce):
        _check(_lib.mrs_swarm_tick_n(self._h, dt, n_ticks, int(enabled), int(crash), float(rebounce)))

    def synchronize(self):
        _check(_lib.mrs_swarm_synchronize(self._h))

    def stream(self):
        st = C.c_void_p()
        _check(_lib.mrs_swarm_stream(self._h, C.byref(st)))
        return st.value

    def set_hold(self, first, count, hold):
        """UAVs on hold are not iterated by step/tick (UavSystemRos without input and iterate_without_input == false)."""
        _check(_lib.mrs_swarm_set_hold(self._h, int(first), int(count), int(bool(hold))))

    def collision_stats(self):
        """(collision ticks, ticks that repeated the neighbour search)"""
        a, b = C.c_int64(), C.c_int64()
        _check(_lib.mrs_swarm_get_collision_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def fused_stats(self):
        """(collision ticks evaluated by the following step launch, stale-list stalls, launches replayed after a stall, searches queued ahead of time)"""
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        _check(_lib.mrs_swarm_get_fused_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return a.value, b.value, c.value, d.value

    def debug_component(self, component, first, count, rows, dt=0.001):
        """one component of the path (COMP_*) for UAVs [first, first + count) on their own state; rows: (count, input width)"""
        wi, wo = COMP_WIDTHS[component]
        rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(count, wi)
        out = np.zeros((count, wo))
        _check(_lib.mrs_swarm_debug_component(self._h, int(component), int(first), int(count), _dp(rows), wi, _dp(out), wo, float(dt)))
        return out

    def debug_search_ms(self, reps=8, crash=False, rebounce=100.0):
        """average device time (ms) of one neighbour search (pack + insert, list-building query) — measurement hook of bench.py"""
        ms = C.c_double()
        _check(_lib.mrs_swarm_debug_search_ms(self._h, int(reps), int(bool(crash)), float(rebounce), C.byref(ms)))
        return ms.value

    def debug_neighbour_lists(self, crash=False, rebounce=100.0, rows=24):
        """one forced search on the current positions -> (count[n], nbr[rows, n], list capacity, list radius): test hook"""
        count, nbr = np.zeros(self.n, dtype=np.uint32), np.zeros((rows, self.n), dtype=np.uint32)
        cap, radius = C.c_int32(), C.c_double()
        u32p = C.POINTER(C.c_uint32)
        _check(_lib.mrs_swarm_debug_neighbour_lists(self._h, int(bool(crash)), float(rebounce), count.ctypes.data_as(u32p), nbr.ctypes.data_as(u32p), int(rows),
                                                    C.byref(cap), C.byref(radius)))
        return count, nbr, cap.value, radius.value

    def set_profiling(self, enabled):
        _check(_lib.mrs_swarm_set_profiling(self._h, int(enabled)))

    def last_step_kernel_ms(self):
        ms, nl = C.c_double(), C.c_int32()
        _check(_lib.mrs_swarm_last_step_kernel_ms(self._h, C.byref(ms), C.byref(nl)))
        return ms.value, nl.value

    # -- multi-GPU collision exchange --
    def pack_positions(self):
        ptr, nb = C.c_void_p(), C.c_int64()
        _check(_lib.mrs_swarm_pack_positions(self._h, C.byref(ptr), C.byref(nb)))
        return ptr.value, nb.value

    def pack_positions_to(self, dev_ptr):
        _check(_lib.mrs_swarm_pack_positions_to(self._h, C.c_void_p(dev_ptr)))

    def handle_collisions_gathered(self, dev_ptr, n_total, my_offset, enabled, crash, rebounce):
        _check(_lib.mrs_swarm_handle_collisions_gathered(self._h, C.c_void_p(dev_ptr), n_total, my_offset, int(enabled),
                                                         int(crash), float(rebounce)))

    # the exchange driven by the library itself (RCCL bound at run time, all-gather on the swarm's stream)
    def comm_init(self, world, rank, unique_id, n_total, librccl_path=None):
        path = default_librccl_path() if librccl_path is None else librccl_path
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        _check(_lib.mrs_swarm_comm_init(self._h, path.encode() if path else None, world, rank, C.cast(buf, C.c_void_p), n_total))

    def tick_sharded_n(self, dt, n_ticks, enabled, crash, rebounce):
        _check(_lib.mrs_swarm_tick_sharded_n(self._h, dt, n_ticks, int(enabled), int(crash), float(rebounce)))

    def comm_destroy(self):
        _check(_lib.mrs_swarm_comm_destroy(self._h))

    def comm_info(self):
        ci = CommInfo()
        _check(_lib.mrs_swarm_comm_info(self._h, C.byref(ci)))
        d = {k: int(getattr(ci, k)) for k, _ in CommInfo._fields_}
        d["parallelism"] = (f"{d['world']} equal-count shards, {EXCHANGE_NAMES.get(d['exchange'], '?')}, "
                            + ("RCCL" if d["rccl_ranks"] else "caller-supplied / in-process collective"))
        return d

    def comm_init_standin(self, world, rank, n_total, collective_latency_us, slab_width):
        """measurement stand-in: this rank alone, its neighbours are images of itself, every collective takes a fixed latency"""
        _check(_lib.mrs_swarm_comm_init_standin(self._h, int(world), int(rank), int(n_total), float(collective_latency_us), float(slab_width)))

    def peer_window_create(self, world, rank, n_total, want_handle=True):
        """this rank's window of the peer-window exchange: (device address, 64-byte IPC handle or None) — include/mrs_swarm.h"""
        ptr = C.c_void_p()
        handle = C.create_string_buffer(64) if want_handle else None
        _check(_lib.mrs_swarm_peer_window_create(self._h, int(world), int(rank), int(n_total), C.byref(ptr), handle))
        return int(ptr.value), (handle.raw if want_handle else None)

    def comm_init_peer(self, windows=None, handles=None):
        """windows: device addresses of every rank's window valid in THIS process (None / 0 entries: use the handle);
        handles: the ranks' 64-byte IPC handles in rank order"""
        arr = None
        if windows is not None:
            arr = (C.c_void_p * len(windows))(*[C.c_void_p(int(w) if w else 0) for w in windows])
        blob = None
        if handles is not None:
            blob = b"".join(h if h is not None else bytes(64) for h in handles)
        _check(_lib.mrs_swarm_comm_init_peer(self._h, arr, blob))

    def debug_chaos(self, max_sleep_us, seed=1):
        _check(_lib.mrs_swarm_debug_chaos(self._h, int(max_sleep_us), int(seed)))

    def split_stats(self):
        a, b = C.c_int64(), C.c_int64()
        _check(_lib.mrs_swarm_get_split_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def search_stats(self):
        """(searches, of them on a halo exchange, halo searches repeated on all records, entries per rank of the next halo block)"""
        v = [C.c_int64() for _ in range(4)]
        _check(_lib.mrs_swarm_get_search_stats(self._h, *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def comm_init_loopback(self, group, rank, n_total):
        _check(_lib.mrs_swarm_comm_init_loopback(self._h, group._h, int(rank), int(n_total)))
        self._group = group  # keep it alive as long as the swarm uses it

    def comm_init_custom(self, world, rank, n_total, fn):
        """fn(user, send_ptr, recv_ptr, bytes_per_rank, stream_ptr) -> 0; kept alive by the swarm"""
        self._allgather_cb = ALLGATHER_FN(fn)
        _check(_lib.mrs_swarm_comm_init_custom(self._h, int(world), int(rank), int(n_total), self._allgather_cb, None))

    def set_exchange(self, exchange):
        _check(_lib.mrs_swarm_set_exchange(self._h, int(exchange)))

    # -- state --
    def get_state(self, first=0, count=None):
        count = self.n - first if count is None else count
        st = dict(x=np.zeros((count, 3)), v=np.zeros((count, 3)), v_prev=np.zeros((count, 3)), R=np.zeros((count, 3, 3)),
                  omega=np.zeros((count, 3)), motor_rpm=np.zeros((count, MAX_MOTORS)))
        _check(_lib.mrs_swarm_get_state(self._h, first, count, *[_dp(st[k]) for k in ("x", "v", "v_prev", "R", "omega", "motor_rpm")]))
        return st

    def set_state(self, first, count, x=None, v=None, R=None, omega=None, motor_rpm=None):
        x, v, omega = _arr(x, (count, 3)), _arr(v, (count, 3)), _arr(omega, (count, 3))
        R, motor_rpm = _arr(R, (count, 9)), _arr(motor_rpm, (count, MAX_MOTORS))
        _check(_lib.mrs_swarm_set_state(self._h, first, count, _dp(x), _dp(v), _dp(R), _dp(omega), _dp(motor_rpm)))

    def _get3(self, fn, first, count, width=3):
        count = self.n - first if count is None else count
        out = np.zeros((count, width))
        _check(fn(self._h, first, count, _dp(out)))
        return out

    def get_imu(self, first=0, count=None):
        return self._get3(_lib.mrs_swarm_get_imu, first, count)

    def get_external_force(self, first=0, count=None):
        return self._get3(_lib.mrs_swarm_get_external_force, first, count)

    def get_pid(self, first=0, count=None):
        return self._get3(_lib.mrs_swarm_get_pid, first, count, 24)

    def set_pid(self, first, count, pid):
        _check(_lib.mrs_swarm_set_pid(self._h, first, count, _dp(_arr(pid, (count, 24)))))

    def set_state_pos(self, first, count, pos, heading=None):
        """MultirotorModel::setStatePos: position, R = AngleAxis(-heading, z) and the spawn height; nothing else changes"""
        _check(_lib.mrs_swarm_set_state_pos(self._h, first, count, _dp(_arr(pos, (count, 3))), _dp(_arr(heading, (count,)))))

    def clone(self):
        """an independent copy of the swarm (mrs_swarm_clone)"""
        other = object.__new__(Swarm)
        other._h = C.c_void_p()
        other.n = self.n
        _check(_lib.mrs_swarm_clone(self._h, C.byref(other._h)))
        return other

    def clone_resized(self, n_uavs):
        """a copy with room for more UAVs (mrs_swarm_clone_resized): the first self.n are copies, the rest UavSystem()"""
        other = object.__new__(Swarm)
        other._h = C.c_void_p()
        other.n = int(n_uavs)
        _check(_lib.mrs_swarm_clone_resized(self._h, int(n_uavs), C.byref(other._h)))
        return other

    def copy_uavs(self, dst_first, src, src_first, count):
        """UAVs [src_first, src_first + count) of `src` (a clone of this swarm, or this swarm) replace [dst_first, ...) here"""
        _check(_lib.mrs_swarm_copy_uavs(self._h, int(dst_first), src._h, int(src_first), int(count)))

    def step_range(self, first, count, dt):
        """makeStep for the UAVs [first, first + count) only (mrs_swarm_step_range)"""
        _check(_lib.mrs_swarm_step_range(self._h, int(first), int(count), float(dt)))

    def get_states(self, first=0, count=None):
        """packed MultirotorModel::State + IMU + crash flag records (mrs_swarm_get_states): one kernel, one copy"""
        count = self.n - first if count is None else count
        out = np.zeros(count, dtype=STATE_DTYPE)
        _check(_lib.mrs_swarm_get_states(self._h, int(first), int(count), out.ctypes.data_as(C.c_void_p)))
        return out

    def timeout_input(self, first, count):
        _check(_lib.mrs_swarm_timeout_input(self._h, first, count))

    def set_mass(self, first, count, mass):
        _check(_lib.mrs_swarm_set_mass(self._h, first, count, float(mass)))

    def set_ground_z(self, first, count, ground_z):
        _check(_lib.mrs_swarm_set_ground_z(self._h, first, count, float(ground_z)))

    def get_outputs(self, first=0, count=None):
        """odom / imu / rangefinder / pose payloads as a structured array (one pack kernel + one D2H copy)."""
        count = self.n - first if count is None else count
        out = np.zeros(count, dtype=OUTPUT_DTYPE)
        assert out.dtype.itemsize == C.sizeof(UavOutput)
        _check(_lib.mrs_swarm_get_outputs(self._h, first, count, out.ctypes.data_as(C.c_void_p)))
        return out

    def get_outputs_view(self, first=0, count=None):
        """get_outputs without the final host copy: a structured-array VIEW of the library's pinned staging buffer, valid until
        the next get_outputs* call."""
        count = self.n - first if count is None else count
        ptr = C.c_void_p()
        _check(_lib.mrs_swarm_get_outputs_view(self._h, first, count, C.byref(ptr)))
        if count == 0:
            return np.zeros(0, dtype=OUTPUT_DTYPE)
        buf = (C.c_char * (count * OUTPUT_DTYPE.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=OUTPUT_DTYPE, count=count)

    def get_outputs_async(self, first=0, count=None):
        """pipelined download: pack behind the steps queued so far, copy on a copy stream; returns a ticket for outputs_wait"""
        count = self.n - first if count is None else count
        t = C.c_int32()
        _check(_lib.mrs_swarm_get_outputs_async(self._h, first, count, C.byref(t)))
        return int(t.value)

    def outputs_wait(self, ticket):
        """blocks until the download of `ticket` has landed (steps queued since keep running); a structured-array VIEW of the pinned
        block, valid until the second get_outputs_async call after the ticket's"""
        ptr, cnt = C.c_void_p(), C.c_int32()
        _check(_lib.mrs_swarm_outputs_wait(self._h, int(ticket), C.byref(ptr), C.byref(cnt)))
        buf = (C.c_char * (cnt.value * OUTPUT_DTYPE.itemsize)).from_address(ptr.value)
        return np.frombuffer(buf, dtype=OUTPUT_DTYPE, count=cnt.value)

    def input_staging(self, count, stride):
        """pinned host rows (count x stride doubles) to be filled with setInput payloads and sent by commit_input"""
        ptr = C.POINTER(C.c_double)()
        _check(_lib.mrs_swarm_input_staging(self._h, int(count), int(stride), C.byref(ptr)))
        if count == 0:
            return np.zeros((0, stride))
        return np.ctypeslib.as_array(ptr, shape=(int(count), int(stride)))

    def commit_input(self, first, count, mode, stride):
        _check(_lib.mrs_swarm_commit_input(self._h, int(first), int(count), int(mode), int(stride)))

    def get_diag(self):
        d = Diag()
        _check(_lib.mrs_swarm_get_diag(self._h, C.byref(d)))
        return {k: int(getattr(d, k)) for k, _ in Diag._fields_}
