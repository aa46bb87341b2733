"""Environment layer: the batched Wire-EDM environment and its device-resident action type.

`WireEDMEnv` keeps the Gymnasium-style surface of the single-environment reference but owns N
environments whose physics runs in one fused HIP kernel; `DeviceAction` is an action that has
already been laid out in GPU memory (see `WireEDMEnv.make_action`).
"""
from . import wire_edm as _wire_edm

WireEDMEnv = _wire_edm.WireEDMEnv
DeviceAction = _wire_edm.DeviceAction

__all__ = ("DeviceAction", "WireEDMEnv")
