"""sparc_amd — MI355X-native batched Wire-EDM environment.

Drop-in for ONE hot path of geduardo/SPARC (``wedm`` 0.2.0): the per-microsecond
physics step of ``WireEDMEnv.step()``, run for N independent environments by one
fused HIP kernel (gfx950) behind the reference's own Gymnasium surface.

    from sparc_amd import WireEDMEnv, EnvironmentConfig
    env = WireEDMEnv(num_envs=65536, device="cuda")
    obs, info = env.reset(seed=0)
    obs, reward, terminated, truncated, info = env.step(action)

The names exported here mirror the reference's ``wedm/__init__.py:22-42``.
"""
from .core.env_config import EnvironmentConfig
from .core.material_db import MaterialDatabase, WireMaterial, get_material_db
from .core.state import BatchedEDMState
from .envs.wire_edm import DeviceAction, WireEDMEnv
from .modules.parameters import (
    DielectricModuleParameters,
    IgnitionModuleParameters,
    MaterialModuleParameters,
    MechanicsModuleParameters,
    WireModuleParameters,
)

from .core.state_utils import get_gap, is_short_circuited
from .modules.views import DielectricModule, IgnitionModule, MaterialRemovalModule, MechanicsModule, WireModule
from .controllers import GapController, VoltageController, run_controlled
from .trace import DeviceTrace
from .utils.logger import LoggerConfig, SimulationLogger
from .vector import WireEDMVectorEnv

EDMState = BatchedEDMState  # the reference's name for the state object

__version__ = "0.1.0"

__all__ = [
    "EDMState", "BatchedEDMState", "EnvironmentConfig", "MaterialDatabase", "WireMaterial", "get_material_db",
    "WireEDMEnv", "DeviceAction", "WireEDMVectorEnv", "GapController", "VoltageController", "run_controlled", "DeviceTrace",
    "SimulationLogger", "LoggerConfig",
    "IgnitionModule", "WireModule", "MaterialRemovalModule", "DielectricModule", "MechanicsModule",
    "get_gap", "is_short_circuited",
    "IgnitionModuleParameters", "WireModuleParameters", "MaterialModuleParameters",
    "DielectricModuleParameters", "MechanicsModuleParameters",
]
