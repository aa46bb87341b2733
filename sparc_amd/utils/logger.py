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

Two derived signals follow the reference's own definitions:
  * ``dielectric_flow_rate`` = ``flow_rate * base_flow_rate / 1e9`` (dielectric.py:160-162);
  * ``wire_average_temperature``: with ``WireModuleParameters(compute_zone_mean=True)`` the reference
    refreshes a cached float32 zone mean every ``zone_mean_interval``-th wire update and logs the
    cache in between (wire.py:339-347; initial value = spool temperature, wire.py:223).  An attached
    logger reproduces exactly that from the traced wire temperature (the mean itself is taken on
    the host with NumPy, bit for bit the reference's ``np.mean`` of the float32 zone).  Without
    ``compute_zone_mean`` the reference logs ``None``; this logger then records the zone mean of
    the sampled wire temperature.
"""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, List, Optional

import numpy as np
import torch

LoggerConfig = Dict[str, Any]


def dielectric_flow_rate(flow_rate: torch.Tensor, base_flow_rate: float) -> torch.Tensor:
    """`state.dielectric_flow_rate = (flow_condition * base_flow_rate) / 1e9` (dielectric.py:160-162).
    Tensor / tensor: torch's GPU kernel for tensor / python-scalar multiplies by the reciprocal."""
    scaled = flow_rate * float(base_flow_rate)
    return scaled / torch.full_like(scaled, 1e9)

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
        self._zm_cache = None  # cached zone mean per environment (wire.py:223: starts at the spool temperature)

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
        zone_cached = "wire_average_temperature" in self.signals and bool(env.wire_params.compute_zone_mean)
        zm_every = int(env.wire_params.zone_mean_interval)
        if self.freq_type == "control_step":
            if not zone_cached or trace is not None:
                raise ValueError("control_step logging needs no device trace: use collect / collect_launch")
            # the only thing a control-step log cannot read from the state after the launch: the zone mean
            # cached at the last multiple of zone_mean_interval -> trace the wire at exactly those steps
            if capacity is None:
                capacity = (env.servo_interval // env.dt + 1) // zm_every + 2
            if envs is not None:
                raise ValueError("a control_step log samples every environment: envs= is not supported here")
            trace = env.bind_trace(["wire_temperature"], every=zm_every, capacity=capacity)
            if trace.bind_step % zm_every:
                raise ValueError("attach the logger when the steps since reset are a multiple of zone_mean_interval")
            self.env, self._trace, self._cursor = env, trace, trace.count
            return trace
        every = self.interval if self.freq_type == "interval" else 1
        if zone_cached and (zm_every % every or env.steps_since_reset % every):
            raise ValueError(f"wire_average_temperature with compute_zone_mean needs a log interval that divides "
                             f"zone_mean_interval={zm_every} (and an attach aligned to it)")
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
        if self.freq_type == "control_step":  # attached only for the cached zone mean (see attach)
            self._refresh_zone_cache()
            self._calls += 1
            if not control_step:
                return
            for name in self.signals:
                if name == "wire_average_temperature":
                    self._data[name].append(self._zm_cache[None, :].copy())
                else:
                    value = getattr(state, name)
                    if isinstance(value, tuple):
                        value = value[0]
                    self._data[name].append(value.detach().clone().unsqueeze(0))
            return
        end = self._trace.count
        if end == self._cursor:
            return
        first = self._cursor
        chunk = self._trace.read_range(first, end, self._trace_names())
        self._cursor = end
        for name in self.signals:
            if name == "wire_average_temperature":
                if self.env.wire_params.compute_zone_mean:
                    self._data[name].append(self._cached_zone_means(chunk["wire_temperature"], first, end))
                else:
                    self._data[name].append(self._zone(chunk["wire_temperature"]).mean(dim=-1))
            elif name == "dielectric_flow_rate":
                self._data[name].append(dielectric_flow_rate(chunk["flow_rate"], self.env.dielectric_params.base_flow_rate))
            else:
                self._data[name].append(chunk[name].clone())

    # ---- wire_average_temperature as the reference caches it (wire.py:339-347,390-394)
    def _zone(self, T):
        g = self.env.geometry
        return T[..., g.az_start:g.az_end] if g is not None and g.az_end > g.az_start else T

    def _host_zone_mean(self, T) -> np.ndarray:
        """float(np.mean(T[zone])) per row: float32 pairwise sum like the reference, as float64."""
        z = np.ascontiguousarray(self._zone(T).cpu().numpy())
        return z.mean(axis=-1, dtype=np.float32).astype(np.float64)

    def _spool_cache(self, count) -> np.ndarray:
        return np.full(count, float(self.env.wire_params.spool_T), dtype=np.float64)

    def _wire_updates_at(self, first: int, last: int) -> np.ndarray:
        """Wire updates since the reset at samples first..last-1 of the trace."""
        tr = self._trace
        return tr.bind_step + (np.arange(first, last, dtype=np.int64) + 1) * tr.every

    def _cached_zone_means(self, T, first: int, last: int) -> np.ndarray:
        """[sample, env] float64: at every sample the zone mean of the last refresh (a wire update whose
        count is a multiple of zone_mean_interval), the spool temperature before the first one."""
        zm_every = int(self.env.wire_params.zone_mean_interval)
        counts = self._wire_updates_at(first, last)
        refresh = counts % zm_every == 0
        if self._zm_cache is None:
            self._zm_cache = self._spool_cache(T.shape[1])
        rows = [self._zm_cache[None, :]]
        if refresh.any():
            idx = np.nonzero(refresh)[0]
            rows.append(self._host_zone_mean(T[idx.tolist()]))
        table = np.concatenate(rows)                      # row 0: the cache carried in, then one row per refresh
        out = table[np.cumsum(refresh)]                   # sample j sees the refreshes up to and including j
        self._zm_cache = table[-1].copy()
        return out

    def _refresh_zone_cache(self) -> None:
        tr = self._trace
        if self._zm_cache is None:
            self._zm_cache = self._spool_cache(tr.env_count)
        end = tr.count
        if end > self._cursor:  # the trace samples exactly the refresh steps; the newest one is the cache
            T = tr.read_range(end - 1, end, ["wire_temperature"])["wire_temperature"]
            self._zm_cache = self._host_zone_mean(T)[0]
            self._cursor = end

    # ------------------------------------------------------------------ output
    def finalize(self) -> None:
        def host(c):
            return c if isinstance(c, np.ndarray) else c.cpu().numpy()

        out = {name: (np.concatenate([host(c) for c in chunks]) if chunks else np.empty((0,)))
               for name, chunks in self._data.items()}
        self._final = out
        if self.filepath is not None:
            self.filepath.parent.mkdir(parents=True, exist_ok=True)
            (np.savez_compressed if self.compress else np.savez)(self.filepath, **out)

    def get_data(self) -> Dict[str, np.ndarray]:
        if self._final is None:
            self.finalize()
        return self._final
