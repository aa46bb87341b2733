"""Compare a batched environment against a reference fixture (tests/golden) environment by
environment.  Shared by the CPU (oracle backend) and GPU tests."""
from __future__ import annotations

import math

import numpy as np

INT_MAP = {
    "time": "time", "time_since_servo": "time_since_servo", "spark_state": "spark_state", "spark_dur": "spark_duration",
    "is_short_circuit": "is_short_circuit", "random_short_remaining": "random_short_remaining",
    "debris_short_remaining": "debris_short_remaining", "time_in_critical_temp": "time_in_critical_temp",
    "is_wire_broken": "is_wire_broken", "is_target_reached": "is_target_distance_reached",
    "ctrl_step": "control_step", "time_since_spark_ignition": "time_since_spark_ignition",
    "time_since_spark_end": "time_since_spark_end", "time_since_open_voltage": "time_since_open_voltage",
}
FLOAT_MAP = {
    "workpiece_position": "workpiece_position", "wire_position": "wire_position", "wire_velocity": "wire_velocity",
    "voltage": "voltage", "current": "current", "spark_y": "spark_location", "debris_volume": "debris_volume",
    "debris_density": "debris_density", "flow_rate": "flow_rate", "cavity_volume": "cavity_volume",
    "last_crater_volume": "last_crater_volume", "prev_accel": "prev_accel", "tmax": "wire_max_temperature",
    "h_base": "h_eff_base", "h_zone": "h_eff_zone", "diel_last_gap": "dielectric_last_gap",
    "diel_last_density": "dielectric_last_density", "wire_last_flow": "wire_last_flow",
}
EXACT_ALWAYS = ("workpiece_position", "wire_position", "wire_velocity", "voltage", "current", "spark_y", "prev_accel")


def check_step(env, fx, env_index, step, *, exact_floats, float_rtol=1e-12, tmax_atol=1e-4):
    """Assert that environment `env_index` equals the fixture at `step` (after that step)."""
    st = env.state
    for name, attr in INT_MAP.items():
        want = int(fx.int_row(name)[step])
        got = int(getattr(st, attr)[env_index].item())
        assert want == got, f"step {step} env {env_index} {name}: reference {want} got {got}"
    pos = np.searchsorted(fx.float_steps, step)
    if pos < len(fx.float_steps) and fx.float_steps[pos] == step:
        for name, attr in FLOAT_MAP.items():
            want = float(fx.float_row(name)[pos])
            got = float(getattr(st, attr)[env_index].item())
            same = (want == got) or (math.isnan(want) and math.isnan(got))
            if exact_floats or name in EXACT_ALWAYS:
                assert same, f"step {step} env {env_index} {name}: reference {want!r} got {got!r}"
            elif name in ("tmax",):
                assert same or abs(want - got) <= tmax_atol, f"step {step} {name}: {want!r} vs {got!r}"
            else:
                assert same or abs(want - got) <= float_rtol * max(abs(want), abs(got)), \
                    f"step {step} env {env_index} {name}: reference {want!r} got {got!r}"
