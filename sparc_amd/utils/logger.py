"""Batched signal logger (the reference's `utils/logger.py:54-237`).

Signals are `EDMState` attribute names, exactly as in the reference (`logger.py:110-116`).
Collection happens on the device (a clone of the state view per sample, no host sync);
`finalize()` moves everything to the host once and optionally writes a compressed `.npz`,
like the reference's numpy backend.  Frequencies: ``control_step`` (sample when the step was a
control step — a host-side schedule check, no device read), ``interval`` (every N-th call) and
``every_step`` (every call; with fused `step_many` launches one call is one launch).
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

LoggerConfig = Dict[str, Any]


class SimulationLogger:
    def __init__(self, config: LoggerConfig, env_reference=None):
        self.config = dict(config)
        self.signals: List[str] = list(config.get("signals_to_log", ["time"]))
        freq = config.get("log_frequency", {"type": "every_step"})
        self.freq_type = freq.get("type", "every_step")
        if self.freq_type not in ("every_step", "control_step", "interval"):
            raise ValueError(f"Unknown log frequency type: {self.freq_type}")
        self.interval = int(freq.get("value", 1)) if self.freq_type == "interval" else 1
        backend = config.get("backend", {"type": "memory"})
        self.backend_type = backend.get("type", "memory")
        if self.backend_type not in ("memory", "numpy"):
            raise ValueError(f"Unknown backend type: {self.backend_type}")
        self.filepath: Optional[Path] = Path(backend["filepath"]) if self.backend_type == "numpy" else None
        self.compress = bool(backend.get("compress", True))
        self.env = env_reference
        self.reset()

    def reset(self) -> None:
        self._data: Dict[str, List[torch.Tensor]] = {s: [] for s in self.signals}
        self._calls = 0
        self._final: Optional[Dict[str, np.ndarray]] = None

    def collect(self, state, info: Optional[Dict[str, Any]] = None, *, control_step: Optional[bool] = None) -> None:
        """Sample the configured signals from `state` (a `BatchedEDMState`).  For the
        ``control_step`` frequency pass ``control_step=True/False`` from the host-side schedule
        (the driver knows it); reading `info["control_step"]` would force a device sync."""
        self._calls += 1
        if self.freq_type == "control_step" and not control_step:
            return
        if self.freq_type == "interval" and (self._calls - 1) % self.interval:
            return
        for name in self.signals:
            value = getattr(state, name)
            if isinstance(value, tuple):  # spark_status -> its state component
                value = value[0]
            self._data[name].append(value.detach().clone())

    def finalize(self) -> None:
        out = {}
        for name, chunks in self._data.items():
            out[name] = torch.stack(chunks).cpu().numpy() if chunks else np.empty((0,))
        self._final = out
        if self.filepath is not None:
            self.filepath.parent.mkdir(parents=True, exist_ok=True)
            (np.savez_compressed if self.compress else np.savez)(self.filepath, **out)

    def get_data(self) -> Dict[str, np.ndarray]:
        if self._final is None:
            self.finalize()
        return self._final
