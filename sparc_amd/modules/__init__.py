from .parameters import (
    DielectricModuleParameters,
    IgnitionModuleParameters,
    MaterialModuleParameters,
    MechanicsModuleParameters,
    WireModuleParameters,
)

__all__ = ["DielectricModuleParameters", "IgnitionModuleParameters", "MaterialModuleParameters",
           "MechanicsModuleParameters", "WireModuleParameters"]
