"""Host-side derivation of everything the kernels treat as constant.

The reference computes its derived constants once, in the module constructors, with
Python-float arithmetic.  The kernels never re-derive them: this module evaluates
the same expressions on the host, in Python floats and in the reference's operand
order, and hands the results to the library verbatim (``wedm_params`` and, for
per-environment geometry, the geometry rows).  That keeps quirks such as
``int(30.0 // 0.2) == 149`` (wire.py:151) bit-identical by construction.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from .. import _abi
from ..modules.parameters import (
    DielectricModuleParameters,
    IgnitionModuleParameters,
    MaterialModuleParameters,
    MechanicsModuleParameters,
    WireModuleParameters,
)
from .env_config import EnvironmentConfig
from .material_db import WireMaterial
from .tables import CRATER, MAX_MODE, MODE_CURRENT, parse_mode


@dataclass
class WireGeometry:
    """What ``WireModule.__init__`` (wire.py:143-257), ``DielectricModule.__init__``
    (dielectric.py:59-66) and ``_calculate_position_increment`` (material.py:158-160)
    derive from (workpiece_height, wire_diameter)."""

    workpiece_height: float
    wire_diameter: float
    n_seg: int
    zone_start: int
    zone_end: int
    az_start: int
    az_end: int
    contact_bottom: int
    contact_top: int
    k_cond: float
    tuf: float
    a_surf: float
    s_area: float
    joule_geom: float
    cavity_coeff: float
    kerf_base: float


def derive_geometry(workpiece_height: float, wire_diameter: float, wire: WireModuleParameters,
                    material: WireMaterial, material_params: MaterialModuleParameters) -> WireGeometry:
    h, d = float(workpiece_height), float(wire_diameter)
    seg = wire.segment_len
    total_len = wire.buffer_len_bottom + h + wire.buffer_len_top  # wire.py:144-148
    n_seg = max(1, int(total_len / seg))  # wire.py:149
    zone_start = int(wire.buffer_len_bottom // seg)  # wire.py:151
    zone_end = zone_start + int(h // seg)  # wire.py:152-154
    zone_end = min(zone_end, n_seg)
    zone_start = min(zone_start, zone_end)
    r_mm = d / 2.0
    dy = seg * 1e-3  # [m]
    s_area = math.pi * (r_mm * 1e-3) ** 2  # wire.py:170
    a_surf = 2 * math.pi * (r_mm * 1e-3) * dy  # wire.py:171
    k_cond = material.thermal_conductivity * s_area / dy  # wire.py:174-176
    denominator = material.density * material.specific_heat * s_area * dy  # wire.py:177-182
    if denominator == 0:
        raise ValueError("Denominator for dT/dt is zero. Check wire/segment properties.")
    joule_geom = dy / s_area if s_area != 0 else 0.0  # wire.py:183
    tuf = 1e-6 / denominator  # wire.py:195
    az_start = min(zone_start, n_seg - 1)  # wire.py:207
    az_end = min(zone_end, n_seg)  # wire.py:208
    cb_pos = wire.buffer_len_bottom - wire.contact_offset_bottom  # wire.py:229-231
    ct_pos = wire.buffer_len_bottom + h + wire.contact_offset_top  # wire.py:232-236
    cb = max(0, int(cb_pos / seg))
    ct = min(n_seg - 1, int(ct_pos / seg))
    cb = max(0, min(cb, zone_start - 1))  # wire.py:247-249
    ct = min(n_seg - 1, max(ct, zone_end))  # wire.py:250-252
    return WireGeometry(
        workpiece_height=h, wire_diameter=d, n_seg=n_seg, zone_start=zone_start, zone_end=zone_end,
        az_start=az_start, az_end=az_end, contact_bottom=cb, contact_top=ct, k_cond=k_cond, tuf=tuf,
        a_surf=a_surf, s_area=s_area, joule_geom=joule_geom,
        cavity_coeff=math.pi * r_mm * h,  # dielectric.py:62
        kerf_base=material_params.base_overcut + d,  # material.py:158-160 (first two terms)
    )


def build_params(config: EnvironmentConfig, control_mode: str, ignition: IgnitionModuleParameters,
                 wire: WireModuleParameters, material_params: MaterialModuleParameters,
                 dielectric: DielectricModuleParameters, mechanics: MechanicsModuleParameters,
                 material: WireMaterial, *, geometry: Optional[WireGeometry], env_id_offset: int = 0,
                 obs_dim: int = _abi.OBS_DIM, disable_ignition: bool = False, autoreset: bool = False,
                 reward_mode: int = 0, reward_break_penalty: float = 10.0, stencil_mode: int = 0,
                 reset_semantics: int = 0, keep_stepping_terminated: bool = False) -> _abi.Params:
    """Fill ``struct wedm_params``.  ``geometry=None`` means per-environment rows."""
    p = _abi.Params()
    p.servo_interval = int(config.servo_interval)
    p.dt_us = int(config.dt)
    p.control_mode = 0 if control_mode == "position" else 1
    p.per_env_geometry = 0 if geometry is not None else 1
    p.initial_gap = float(config.initial_gap)
    p.target_cutting_distance = float(config.target_cutting_distance)
    if geometry is not None:
        g = geometry
        p.n_seg, p.zone_start, p.az_start, p.az_end = g.n_seg, g.zone_start, g.az_start, g.az_end
        p.contact_bottom, p.contact_top = g.contact_bottom, g.contact_top
        p.workpiece_height, p.kerf_base, p.cavity_coeff = g.workpiece_height, g.kerf_base, g.cavity_coeff
        p.k_cond, p.tuf, p.a_surf, p.s_area, p.joule_geom = g.k_cond, g.tuf, g.a_surf, g.s_area, g.joule_geom
    p.segment_len = float(wire.segment_len)

    p.spool_T = float(wire.spool_T)
    p.temp_ref = 293.15  # wire.py:192
    p.rho_elec = float(material.electrical_resistivity)
    p.alpha_rho = float(material.temperature_coefficient)
    p.rho_c = float(material.density * material.specific_heat)  # wire.py:305-310 prefix
    p.plasma_efficiency = float(wire.plasma_efficiency)
    p.base_convection = float(wire.base_convection_coefficient)
    p.convection_velocity_factor = float(wire.convection_velocity_factor)
    p.convection_flow_enhancement = float(wire.convection_flow_enhancement)
    p.critical_temperature = float(material.melting_point * wire.critical_temp_threshold)  # wire.py:216-218
    p.breaking_temperature = float(material.breaking_temperature)  # wire.py:220
    p.dielectric_temperature = float(dielectric.dielectric_temperature)

    p.base_critical_density = float(ignition.base_critical_density)
    p.gap_coefficient = float(ignition.gap_coefficient)
    p.max_critical_density = float(ignition.max_critical_density)
    p.hard_short_gap = float(ignition.hard_short_gap)
    p.sigmoid_steepness = float(ignition.sigmoid_steepness)
    p.debris_short_duration = int(ignition.debris_short_duration)
    p.random_short_duration = int(ignition.random_short_duration)
    p.random_short_min_gap = float(ignition.random_short_min_gap)
    p.random_short_max_gap = float(ignition.random_short_max_gap)
    p.random_short_max_probability = float(ignition.random_short_max_probability)
    p.ignition_a = float(ignition.ignition_a_coeff)
    p.ignition_b = float(ignition.ignition_b_coeff)
    p.ignition_c = float(ignition.ignition_c_coeff)
    p.ln2 = float(np.log(2))  # ignition.py:362
    p.default_target_voltage = float(ignition.default_target_voltage)
    p.default_on_time = float(ignition.default_on_time)
    p.default_off_time = float(ignition.default_off_time)
    default_mode = parse_mode(ignition.default_current_mode)
    if default_mode not in MODE_CURRENT:
        raise KeyError(ignition.default_current_mode)  # the reference's dict lookup, ignition.py:110
    p.default_current = float(MODE_CURRENT[default_mode])
    p.spark_voltage_factor = float(ignition.spark_voltage_factor)

    p.reference_gap = float(dielectric.reference_gap)
    p.debris_obstruction_coeff = float(dielectric.debris_obstruction_coeff)
    p.debris_removal_per_us = dielectric.debris_removal_efficiency * dielectric.base_flow_rate * 1e-6  # :64-66

    p.dt_s = config.dt * 1e-6  # mechanics.py:48
    p.damping_coeff = -2.0 * mechanics.zeta * mechanics.omega_n  # mechanics.py:52
    p.stiffness_coeff = -(mechanics.omega_n ** 2)  # mechanics.py:53
    p.omega_n = float(mechanics.omega_n)
    p.max_acceleration = float(mechanics.max_acceleration)
    p.max_jerk_dt = mechanics.max_jerk * p.dt_s  # mechanics.py:57
    p.max_speed = float(mechanics.max_speed)

    for n in range(MAX_MODE + 1):
        p.mode_current[n] = float(MODE_CURRENT.get(n, 0.0))
        mean, std, depth = CRATER.get(n, (0.0, 0.0, 0.0))
        p.crater_mean[n], p.crater_std[n], p.crater_depth[n] = mean, std, depth
        p.crater_valid[n] = 1 if n in CRATER else 0
    p.env_id_offset = int(env_id_offset) & 0xFFFFFFFF
    p.obs_dim = int(obs_dim)
    p.disable_ignition = 1 if disable_ignition else 0
    p.autoreset = 1 if autoreset else 0
    p.reward_mode = int(reward_mode)
    p.stencil_mode = int(stencil_mode)
    p.reward_break_penalty = float(reward_break_penalty)
    p.reset_semantics = int(reset_semantics)
    p.keep_stepping_terminated = 1 if keep_stepping_terminated else 0
    return p


def geometry_rows(heights: Sequence[float], diameters: Sequence[float], wire: WireModuleParameters,
                  material: WireMaterial, material_params: MaterialModuleParameters, stride: int):
    """Per-environment geometry blocks (BASELINE config 5).  Returns
    ``(f64[GEOM_F64_COUNT, stride], i32[GEOM_I32_COUNT, stride], n_seg_max)`` as NumPy
    arrays; padding environments repeat the last real one."""
    heights = np.asarray(heights, dtype=np.float64).reshape(-1)
    diameters = np.asarray(diameters, dtype=np.float64).reshape(-1)
    if heights.shape != diameters.shape:
        raise ValueError("workpiece_height and wire_diameter must have one value per environment")
    n = heights.shape[0]
    f64 = np.zeros((_abi.GEOM_F64_COUNT, stride), dtype=np.float64)
    i32 = np.zeros((_abi.GEOM_I32_COUNT, stride), dtype=np.int32)
    cache = {}
    for e in range(n):
        key = (float(heights[e]), float(diameters[e]))
        g = cache.get(key)
        if g is None:
            if key[0] <= 0:
                raise ValueError("workpiece_height must be positive")
            if key[1] <= 0:
                raise ValueError("wire_diameter must be positive")
            g = cache[key] = derive_geometry(key[0], key[1], wire, material, material_params)
        f64[:, e] = (g.workpiece_height, g.kerf_base, g.cavity_coeff, g.k_cond, g.tuf, g.a_surf, g.s_area,
                     g.joule_geom)
        i32[:, e] = (g.n_seg, g.zone_start, g.az_start, g.az_end, g.contact_bottom, g.contact_top)
    if n < stride:
        f64[:, n:] = f64[:, n - 1: n]
        i32[:, n:] = i32[:, n - 1: n]
    return f64, i32, int(i32[_abi.GI32.N_SEG, :n].max())
