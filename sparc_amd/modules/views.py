"""What is left of the reference's module OBJECTS on the host.

On the GPU the five modules are fused into one kernel, so `env.ignition`, `env.material` ... carry
no `update()`; they keep the parameters and the public read-only helpers users call from analysis
and plotting code.  The helpers take Python floats or tensors (any device) and broadcast.
"""
from __future__ import annotations

import math
from typing import Any, Dict

import torch

from ..core.tables import CRATER, MAX_MODE, MODE_CURRENT


def _t(x, like=None) -> torch.Tensor:
    if torch.is_tensor(x):
        return x.to(torch.float64)
    return torch.as_tensor(x, dtype=torch.float64, device=None if like is None else like.device)


class IgnitionView:
    """`IgnitionModule`'s parameters and getters (modules/ignition.py:348-399)."""

    def __init__(self, env, params=None):
        self._env = self.env = env
        self.params = params if params is not None else env.ignition_params

    def get_critical_density_for_gap(self, gap):
        """ignition.py:366-377: 0 below the hard-short gap, else min(base + k*gap, max)."""
        p, gap = self.params, _t(gap)
        crit = torch.clamp(p.base_critical_density + p.gap_coefficient * gap, max=p.max_critical_density)
        return torch.where(gap < p.hard_short_gap, torch.zeros_like(crit), crit)

    def get_debris_short_probability(self, gap, debris_density):
        """ignition.py:115-146: hard short below `hard_short_gap`, else the sigmoid in the debris
        density with the reference's +-500 exponent guards."""
        p, gap = self.params, _t(gap)
        rho = _t(debris_density, gap)
        crit = torch.clamp(p.base_critical_density + p.gap_coefficient * gap, max=p.max_critical_density)
        ex = -p.sigmoid_steepness * (rho - crit)
        prob = 1.0 / (1.0 + torch.exp(torch.clamp(ex, -500.0, 500.0)))
        prob = torch.where(ex > 500.0, torch.zeros_like(prob), torch.where(ex < -500.0, torch.ones_like(prob), prob))
        return torch.where(gap < p.hard_short_gap, torch.ones_like(prob), prob)

    def get_lambda(self, state=None):
        """ignition.py:348-364: ln2 / (a gap^2 + b gap + c) from the state's unclamped gap, one value
        per environment (the reference raises during a short; here shorted environments get NaN)."""
        p = self.params
        st = self._env.state if state is None else state
        gap = st.workpiece_position - st.wire_position
        lam = math.log(2.0) / (p.ignition_a_coeff * gap * gap + p.ignition_b_coeff * gap + p.ignition_c_coeff)
        return torch.where(st.is_short_circuit, torch.full_like(lam, float("nan")), lam)

    def get_short_circuit_status(self) -> Dict[str, torch.Tensor]:
        return self._env.get_short_circuit_status()


class MaterialView:
    """`MaterialRemovalModule`'s tables and getters (modules/material.py:176-227)."""

    def __init__(self, env, params=None):
        self._env = self.env = env
        self.params = params if params is not None else env.material_params
        self.currents_data = {f"I{m}": {"Current": a} for m, a in MODE_CURRENT.items()}
        self.crater_data = {f"I{m}": {"ellipsoid_volume_half": v, "ellipsoid_volume_std": s, "depth": d}
                            for m, (v, s, d) in CRATER.items()}

    def get_crater_data_for_current_mode(self, current_mode: str) -> Dict[str, Any]:
        key = current_mode if current_mode in self.currents_data else "I1"          # material.py:178-180
        out = {"current_mode": key, "machine_current": self.currents_data[key]["Current"]}
        if key not in self.crater_data:
            out["crater_data"] = None
            out["error"] = f"No crater data available for {key}. Available: {list(self.crater_data)}"
        else:
            out["crater_data"] = self.crater_data[key]
        return out

    def get_current_mapping_table(self) -> Dict[str, Dict[str, Any]]:
        return {f"I{i}": self.get_crater_data_for_current_mode(f"I{i}") for i in range(1, MAX_MODE + 1)}

    def get_crater_statistics(self) -> Dict[str, torch.Tensor]:
        return self._env.get_crater_statistics()


class DielectricView:
    """`DielectricModule`'s parameters and statistics (modules/dielectric.py:164-182)."""

    def __init__(self, env, params=None):
        self._env = self.env = env
        self.params = params if params is not None else env.dielectric_params

    def get_debris_statistics(self) -> Dict[str, torch.Tensor]:
        return self._env.get_debris_statistics()


class WireView:
    """`WireModule`'s parameters, derived geometry and zone-mean helper (modules/wire.py:143-257,390-398)."""

    def __init__(self, env, params=None):
        self._env = self.env = env
        self.params = params if params is not None else env.wire_params
        self.n_segments, self.wire_material, self.geometry = env.n_segments, env.wire_material, env.geometry

    def compute_zone_mean_temperature(self, temperature_field=None) -> torch.Tensor:
        """wire.py:395-398 for every environment (``temperature_field``: ``[N, n_seg]``, default the state's)."""
        if temperature_field is None:
            return self._env.zone_mean_temperature()
        g = self.geometry
        T = torch.as_tensor(temperature_field)
        return T[..., g.az_start:g.az_end].mean(dim=-1) if g is not None and g.az_end > g.az_start else T.mean(dim=-1)


class MechanicsView:
    """`MechanicsModule`'s parameters and control mode (modules/mechanics.py:29-67)."""

    def __init__(self, env, params=None):
        self._env = self.env = env
        self.params = params if params is not None else env.mechanics_params
        self.control_mode = env.mechanics_control_mode


# the reference's class names (wedm/__init__.py:22-42), for imports and isinstance checks; the
# physics these classes carry in the reference runs inside the fused kernel
IgnitionModule, WireModule, MaterialRemovalModule = IgnitionView, WireView, MaterialView
DielectricModule, MechanicsModule = DielectricView, MechanicsView
