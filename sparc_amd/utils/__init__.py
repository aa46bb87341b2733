from .logger import LoggerConfig, SimulationLogger

__all__ = ["SimulationLogger", "LoggerConfig"]
