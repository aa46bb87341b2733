"""Wire-material database — API of the reference's ``core/material_db.py:10-107``.

The one material the reference ships (``data/wire_materials.json``: brass) is built
in; more can be loaded from a JSON file of the same shape."""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass
from pathlib import Path
from typing import Dict, Optional


@dataclass
class WireMaterial:
    name: str
    density: float  # [kg/m^3]
    specific_heat: float  # [J/kg/K]
    thermal_conductivity: float  # [W/m/K]
    electrical_resistivity: float  # [Ohm m]
    temperature_coefficient: float  # [1/K]
    melting_point: float  # [K]
    breaking_temperature: float  # [K]


# integer literals on purpose: the reference reads these from JSON as ints and its
# constructor arithmetic (wire.py:174-182) must see the same values
_BUILTIN = {
    "brass": dict(density=8400, specific_heat=377, thermal_conductivity=120, electrical_resistivity=6.4e-8,
                  temperature_coefficient=0.0039, melting_point=1173, breaking_temperature=1500),
}


class MaterialDatabase:
    def __init__(self, data_dir: Optional[Path] = None):
        self.data_dir = Path(data_dir) if data_dir is not None else None
        self._wire_materials: Dict[str, WireMaterial] = {
            name: WireMaterial(name=name, **props) for name, props in _BUILTIN.items()
        }
        if self.data_dir is not None:
            wire_file = self.data_dir / "wire_materials.json"
            if wire_file.exists():
                with open(wire_file, "r") as fh:
                    for name, props in json.load(fh).items():
                        self._wire_materials[name] = WireMaterial(name=name, **props)

    def get_wire_material(self, name: str) -> WireMaterial:
        if name not in self._wire_materials:
            raise ValueError(f"Unknown wire material: {name}. Available: {list(self._wire_materials.keys())}")
        return self._wire_materials[name]

    def save_materials(self) -> None:
        if self.data_dir is None:
            raise ValueError("MaterialDatabase was created without a data_dir")
        self.data_dir.mkdir(parents=True, exist_ok=True)
        data = {}
        for name, mat in self._wire_materials.items():
            row = asdict(mat)
            row.pop("name")
            data[name] = row
        with open(self.data_dir / "wire_materials.json", "w") as fh:
            json.dump(data, fh, indent=2)


_material_db: Optional[MaterialDatabase] = None


def get_material_db() -> MaterialDatabase:
    global _material_db
    if _material_db is None:
        _material_db = MaterialDatabase()
    return _material_db
