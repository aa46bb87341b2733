"""Batched signal logger (the reference's `utils/logger.py:54-237`).

Signals are `EDMState` attribute names, exactly as in the reference (`logger.py:110-116`).
Collection happens on the device (no host sync); `finalize()` moves everything to the host
once and optionally writes a compressed `.npz`, like the reference's numpy backend.

Two ways to feed it:
  * `collect(state, info)` after a call, as in the reference: frequencies ``control_step``
    (sample when the call ended on a control step — a host-side schedule check, no device
    read), ``interval`` (every N-th call, `logger.py:130-132`) and ``every_step`` (every call);
  * `attach(env)` + `collect_launch(...)` after each fused launch: ``every_step`` and
    ``interval`` then mean every (N-th) MICROSECOND, sampled inside the kernels through the
    environment's device trace (`WireEDMEnv.bind_trace`), which is what the reference's driver
    gets by calling `collect` after each 1-us step (experiments/run_simulation.py:255-257).
    Logged arrays are `[sample, env, ...]`.
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

LoggerConfig = Dict[str, Any]

# signals that are functions of a traced field: name -> the field the trace must record
_DERIVED = {"wire_average_temperature": "wire_temperature", "dielectric_flow_rate": "flow_rate"}


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
        self._trace = None
        self.reset()

    def reset(self) -> None:
        self._data: Dict[str, List[torch.Tensor]] = {s: [] for s in self.signals}
        self._calls = 0
        self._final: Optional[Dict[str, np.ndarray]] = None
        self._cursor = self._trace.count if self._trace is not None else 0

    # ------------------------------------------------------------------ per-call feeding
    def collect(self, state, info: Optional[Dict[str, Any]] = None, *, control_step: Optional[bool] = None) -> None:
        """Sample the configured signals from `state` (a `BatchedEDMState`).  For the
        ``control_step`` frequency pass ``control_step=True/False`` from the host-side schedule
        (the driver knows it); reading `info["control_step"]` would force a device sync."""
        self._calls += 1
        if self.freq_type == "control_step" and not control_step:
            return
        if self.freq_type == "interval" and self._calls % self.interval:
            return
        for name in self.signals:
            value = getattr(state, name)
            if isinstance(value, tuple):  # spark_status -> its state component
                value = value[0]
            self._data[name].append(value.detach().clone().unsqueeze(0))

    # ------------------------------------------------------------------ device-trace feeding
    def attach(self, env, *, capacity: Optional[int] = None, envs=None, trace=None):
        """Sample inside the kernels.  Binds a device trace of this logger's signals on `env`
        (``every`` = 1 for ``every_step``, N for ``interval``) with room for ``capacity`` samples
        between two `collect_launch` calls (default: one control interval + 1 microsecond), or
        adopts `trace` (a `DeviceTrace` that already records the signals, e.g. the one shared with
        a `VoltageController`).  ``wire_average_temperature`` is derived from the traced wire
        temperature (float32 mean over the workpiece zone, wire.py:390-398) and
        ``dielectric_flow_rate`` from the traced flow condition (dielectric.py:160-162)."""
        if self.freq_type == "control_step":
            raise ValueError("control_step logging needs no device trace: use collect / collect_launch")
        every = self.interval if self.freq_type == "interval" else 1
        names = self._trace_names()
        if trace is None:
            if capacity is None:
                capacity = (env.servo_interval // env.dt + every) // every + 1
            trace = env.bind_trace(names, every=every, capacity=capacity, envs=envs)
        else:
            have = set(trace.signals) | ({"wire_temperature"} if trace.wire_temperature else set())
            if trace.every != every or not set(names) <= have:
                raise ValueError("the shared trace does not record this logger's signals at its frequency")
        self.env, self._trace, self._cursor = env, trace, trace.count
        return trace

    def _trace_names(self) -> List[str]:
        names: List[str] = []
        for s in self.signals:
            s = _DERIVED.get(s, s)
            if s not in names:
                names.append(s)
        return names

    def collect_launch(self, state, *, control_step: bool) -> None:
        """Call after every fused launch (`run_controlled(..., logger=...)` does)."""
        if self._trace is None:
            self.collect(state, control_step=control_step)
            return
        end = self._trace.count
        if end == self._cursor:
            return
        chunk = self._trace.read_range(self._cursor, end, self._trace_names())
        self._cursor = end
        for name in self.signals:
            if name == "wire_average_temperature":
                g = self.env.geometry
                T = chunk["wire_temperature"]
                zone = T[..., g.az_start:g.az_end] if g is not None and g.az_end > g.az_start else T
                self._data[name].append(zone.mean(dim=-1))
            elif name == "dielectric_flow_rate":
                self._data[name].append(chunk["flow_rate"] * float(self.env.dielectric_params.base_flow_rate))
            else:
                self._data[name].append(chunk[name].clone())

    # ------------------------------------------------------------------ output
    def finalize(self) -> None:
        out = {name: (torch.cat(chunks).cpu().numpy() if chunks else np.empty((0,)))
               for name, chunks in self._data.items()}
        self._final = out
        if self.filepath is not None:
            self.filepath.parent.mkdir(parents=True, exist_ok=True)
            (np.savez_compressed if self.compress else np.savez)(self.filepath, **out)

    def get_data(self) -> Dict[str, np.ndarray]:
        if self._final is None:
            self.finalize()
        return self._final
