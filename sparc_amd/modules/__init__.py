"""Parameter sets of the five physics stages (ignition, wire heat, material removal, dielectric,
servo mechanics).  On the GPU the stages themselves are fused into one kernel
(`sparc_amd/csrc/wedm_device.h`); these dataclasses remain the way to configure them and keep the
reference's field names and defaults."""
from . import parameters as _p

IgnitionModuleParameters = _p.IgnitionModuleParameters
WireModuleParameters = _p.WireModuleParameters
MaterialModuleParameters = _p.MaterialModuleParameters
DielectricModuleParameters = _p.DielectricModuleParameters
MechanicsModuleParameters = _p.MechanicsModuleParameters

__all__ = tuple(sorted(name for name in dir(_p) if name.endswith("ModuleParameters")))
