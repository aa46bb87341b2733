"""``EnvironmentConfig`` — same fields, defaults and helpers as the reference's
``wedm.core.env_config.EnvironmentConfig`` (core/env_config.py:10-90), so existing
configuration code and JSON files keep working with the batched environment."""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, fields
from pathlib import Path
from typing import Any, Dict


@dataclass
class EnvironmentConfig:
    # workpiece / wire
    workpiece_height: float = 20.0  # [mm]
    wire_diameter: float = 0.2  # [mm]
    wire_material: str = "brass"
    # time base
    dt: int = 1  # [us] physics step
    servo_interval: int = 1000  # [us] control step
    # cut
    initial_gap: float = 50.0  # [um]
    target_cutting_distance: float = 500.0  # [um]
    # declared by the reference but never read by its physics
    max_wire_temperature: float = 1500.0  # [K]
    min_gap_for_operation: float = 2.0  # [um]
    max_cutting_force: float = 100.0  # [N]

    # ---- (de)serialisation: unknown keys are ignored, as in the reference ----------------------
    @classmethod
    def from_dict(cls, config_dict: Dict[str, Any]) -> "EnvironmentConfig":
        accepted = {f.name for f in fields(cls)}
        return cls(**{key: config_dict[key] for key in config_dict if key in accepted})

    @classmethod
    def from_json(cls, json_path: str | Path) -> "EnvironmentConfig":
        return cls.from_dict(json.loads(Path(json_path).read_text()))

    def to_dict(self) -> Dict[str, Any]:
        return asdict(self)

    def to_json(self, json_path: str | Path) -> None:
        Path(json_path).write_text(json.dumps(self.to_dict(), indent=2))

    def validate(self) -> None:
        """Raises ``ValueError`` exactly where the reference does (env_config.py:73-90)."""
        for name in ("workpiece_height", "wire_diameter", "initial_gap", "target_cutting_distance", "dt",
                     "servo_interval"):
            if getattr(self, name) <= 0:
                raise ValueError(f"{name} must be positive")
        if self.max_wire_temperature <= 293.15:
            raise ValueError("max_wire_temperature must be greater than room temperature")
