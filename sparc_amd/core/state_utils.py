"""Batched forms of the reference's state helpers (core/state_utils.py:10-43)."""
from __future__ import annotations

import torch


def is_short_circuited(state, dielectric_module=None) -> torch.Tensor:
    """bool[N]: an explicit short pulse (spark state -1) or the ignition module's short flag."""
    return (state.spark_state == -1) | state.is_short_circuit


def get_gap(state) -> torch.Tensor:
    """float64[N]: max(0, workpiece_position - wire_position) in micrometres."""
    return torch.clamp(state.workpiece_position - state.wire_position, min=0.0)
