from .wire_edm import DeviceAction, WireEDMEnv

__all__ = ["WireEDMEnv", "DeviceAction"]
