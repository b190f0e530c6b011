"""redmax_amd -- MI355X-native batched RedMax BDF1/BDF2 forward dynamics (host-side mirror).

Scene construction keeps the reference's +redmax class surface (redmax.py, scenes.py); the numerics
run in libredmax_hip.so (csrc/, C ABI in include/redmax_hip.h) on gfx950. No CPU fallback.
"""
from . import se3  # noqa: F401
from .redmax import (Body, BodyCuboid, ForceGroundCuboid, Joint, JointFixed, JointPrismatic, JointRevolute, Scene)  # noqa: F401
from .scenes import (IN_SCOPE_SCENES, sceneAdjointChain, sceneChain, sceneChainGround, scenesRedMax, sceneTree,  # noqa: F401
                     syntheticStates)
from .batch import BatchSim, GroupSim  # noqa: F401
from .driver import (driverRedMaxAdjointBDF1, driverRedMaxAdjointBDF2, driverRedMaxBDF1, driverRedMaxBDF2, simLoop, taskObjective,  # noqa: F401
                     testRedMax)
from ._abi import RedMaxHipError  # noqa: F401
