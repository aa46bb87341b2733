"""The five ``*ModuleParameters`` dataclasses of the reference, field for field
(ignition.py:17-57, wire.py:16-54, material.py:17-22, dielectric.py:15-31,
mechanics.py:12-23).  On the GPU the modules themselves are fused into one kernel;
these dataclasses remain the way to parameterise them."""
from __future__ import annotations

from dataclasses import dataclass


@dataclass
class IgnitionModuleParameters:
    base_critical_density: float = 0.3
    gap_coefficient: float = 0.02  # [1/um]
    max_critical_density: float = 0.95
    hard_short_gap: float = 2.0  # [um]
    sigmoid_steepness: float = 500.0
    debris_short_duration: int = 50  # [us]
    random_short_duration: int = 100  # [us]
    random_short_min_gap: float = 2.0  # [um]
    random_short_max_gap: float = 50.0  # [um]
    random_short_max_probability: float = 0.000  # [1/us]
    ignition_a_coeff: float = 0.48
    ignition_b_coeff: float = -3.69
    ignition_c_coeff: float = 14.05
    default_target_voltage: float = 80.0  # [V]
    default_on_time: float = 3.0  # [us]
    default_off_time: float = 80.0  # [us]
    default_current_mode: str = "I5"
    spark_voltage_factor: float = 0.3


@dataclass
class WireModuleParameters:
    buffer_len_bottom: float = 30.0  # [mm]
    buffer_len_top: float = 30.0  # [mm]
    segment_len: float = 0.2  # [mm]
    spool_T: float = 293.15  # [K]
    contact_offset_bottom: float = 10.0  # [mm]
    contact_offset_top: float = 10.0  # [mm]
    base_convection_coefficient: float = 14000  # [W/m^2/K]
    plasma_efficiency: float = 0.1
    convection_velocity_factor: float = 0.5
    convection_flow_enhancement: float = 1.0
    compute_zone_mean: bool = False
    zone_mean_interval: int = 100
    critical_temp_threshold: float = 0.9
    wire_breaking_temp_factor: float = 1.1  # never read by the reference's physics


@dataclass
class MaterialModuleParameters:
    base_overcut: float = 0.12  # [mm]


@dataclass
class DielectricModuleParameters:
    base_flow_rate: float = 100.0  # [mm^3/s]
    debris_removal_efficiency: float = 0.01
    debris_obstruction_coeff: float = 1.0
    reference_gap: float = 25.0  # [um]
    dielectric_temperature: float = 293.15  # [K]
    ion_channel_duration: int = 6  # [us] (write-only in the reference)


@dataclass
class MechanicsModuleParameters:
    omega_n: float = 235.0  # [rad/s]
    zeta: float = 0.38
    max_acceleration: float = 3.0e5  # [um/s^2]
    max_jerk: float = 1.0e8  # [um/s^3]
    max_speed: float = 3.0e4  # [um/s]
