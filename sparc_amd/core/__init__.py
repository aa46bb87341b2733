"""Core data layer of the batched environment.

* `BatchedEDMState` (alias `EDMState`) — struct-of-arrays state blocks with per-field tensor views
* `EnvironmentConfig` — fixed physical / timing configuration, JSON round-trip, validation
* `MaterialDatabase`, `WireMaterial`, `get_material_db` — wire material constants
* `derive` — host-side derivation of every constant the kernels consume
* `tables` — generator current modes and crater statistics
"""
from . import derive, tables
from .env_config import EnvironmentConfig
from .material_db import MaterialDatabase, WireMaterial, get_material_db
from .state import BatchedEDMState

EDMState = BatchedEDMState

__all__ = (
    "BatchedEDMState",
    "EDMState",
    "EnvironmentConfig",
    "MaterialDatabase",
    "WireMaterial",
    "derive",
    "get_material_db",
    "tables",
)
