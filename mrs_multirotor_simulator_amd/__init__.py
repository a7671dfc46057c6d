"""MI355X-native multi-UAV stepper: drop-in for the UavSystem::makeStep() hot path of
ctu-mrs/mrs_multirotor_simulator (HIP kernels behind the C ABI of include/mrs_swarm.h)."""
from .airframes import AIRFRAMES, model_params  # noqa: F401
from .swarm import (ACCELERATION_HDG_CMD, ACCELERATION_HDG_RATE_CMD, ACTUATOR_CMD, ARITH_FAST, ARITH_LITERAL,  # noqa: F401
                    ATTITUDE_CMD, ATTITUDE_RATE_CMD, CONTROL_GROUP_CMD, FF_ACCELERATION_HDG, FF_ACCELERATION_HDG_RATE,
                    FF_VELOCITY_HDG, FF_VELOCITY_HDG_RATE, INPUT_UNKNOWN, MAX_MOTORS, POSITION_CMD, TILT_HDG_RATE_CMD,
                    VELOCITY_HDG_CMD, VELOCITY_HDG_RATE_CMD, EXCHANGE_EXPORT_SETS, EXCHANGE_FULL_GATHER, LoopbackGroup, ModelParams, MrsError, Swarm,
                    default_params, load_library, slab_partition, cell_order)
