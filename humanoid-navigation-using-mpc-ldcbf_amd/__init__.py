"""lipmpc — MI355X-native batched LIP-MPC / LDCBF step solver (drop-in for the per-timestep MPC
solve of salvatore373/Humanoid-Navigation-using-MPC-LDCBF, HumanoidNavigation/MPC)."""
from .solver import (BatchedLipMpc, LipMpcParams, pack_rings, unpack_active, FLAG_INTERIOR, FLAG_WARM_START, FLAG_NO_PRESOLVE,  # noqa: F401
                     STATUS_SOLVED, STATUS_MAX_ITER, STATUS_INFEASIBLE, STATUS_DEGENERATE, STATUS_UNCERTIFIED,
                     STATUS_SENSOR_OVERFLOW)
from .compat import HumanoidMPC, HumanoidMPCCustomLCBF, HumanoidMPCWithRRT  # noqa: F401
from .lidar import LidarSensor, HumanoidMPCUnknownEnvironment, UnknownEnvFleet, ray_table  # noqa: F401
