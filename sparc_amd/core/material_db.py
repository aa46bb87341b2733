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


WIRE_FILE = "wire_materials.json"


class MaterialDatabase:
    """Name -> `WireMaterial`.  Starts from the built-in table; a directory holding a
    `wire_materials.json` of the reference's shape adds to / overrides it."""

    def __init__(self, data_dir: Optional[Path] = None):
        self.data_dir = None if data_dir is None else Path(data_dir)
        self._wire_materials: Dict[str, WireMaterial] = {}
        self._register(_BUILTIN)
        if self.data_dir is not None and (self.data_dir / WIRE_FILE).exists():
            self._register(json.loads((self.data_dir / WIRE_FILE).read_text()))

    def _register(self, table: Dict[str, Dict[str, float]]) -> None:
        for key, constants in table.items():
            self._wire_materials[key] = WireMaterial(name=key, **constants)

    def names(self):
        return sorted(self._wire_materials)

    def get_wire_material(self, name: str) -> WireMaterial:
        if name not in self._wire_materials:
            raise ValueError(f"Unknown wire material: {name}. Available: {list(self._wire_materials.keys())}")
        return self._wire_materials[name]

    def save_materials(self) -> None:
        """Write the current table as `<data_dir>/wire_materials.json`."""
        if self.data_dir is None:
            raise ValueError("MaterialDatabase was created without a data_dir")
        self.data_dir.mkdir(parents=True, exist_ok=True)
        table = {key: {f: v for f, v in asdict(mat).items() if f != "name"}
                 for key, mat in self._wire_materials.items()}
        (self.data_dir / WIRE_FILE).write_text(json.dumps(table, indent=2))


_material_db: Optional[MaterialDatabase] = None


def get_material_db() -> MaterialDatabase:
    global _material_db
    if _material_db is None:
        _material_db = MaterialDatabase()
    return _material_db
