from .env_config import EnvironmentConfig
from .material_db import MaterialDatabase, WireMaterial, get_material_db
from .state import BatchedEDMState

EDMState = BatchedEDMState

__all__ = ["EDMState", "BatchedEDMState", "EnvironmentConfig", "MaterialDatabase", "WireMaterial", "get_material_db"]
