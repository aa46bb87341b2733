"""Device-side signal trace: per-microsecond samples taken INSIDE the fused launch.

The reference logs by calling `SimulationLogger.collect(env.state, info)` after every 1-us
`env.step()` (experiments/run_simulation.py:255-257, utils/logger.py:110-160) and keeps a 1 ms
voltage history the same way (run_simulation.py:262-270).  With 1000 microseconds per kernel
launch there is no host code between two steps, so the kernels themselves copy the selected
`EDMState` fields of a range of environments into ring buffers (`wedm_bind_trace`,
include/wedm_hip.h).  `DeviceTrace` owns those buffers (torch tensors; the library only borrows
the pointers) and turns the ring back into chronological `[sample, env]` tensors.

    trace = env.bind_trace(["voltage", "current", "wire_position"], every=1, capacity=2000, envs=(0, 64))
    env.step_many(action, 1000)
    v = trace.read()["voltage"]            # float64[1000, 64], microsecond by microsecond
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import _abi
from .core.state import _FIELDS

_BLOCK_COUNT = {"f64": _abi.F64_COUNT, "i32": _abi.I32_COUNT, "i8": _abi.I8_COUNT}
# the reference logs the [state, y, dur] list, we log its state; the 64-bit clocks are traced by their low 32 bits (the
# rows the kernels carry through the microsecond loop; the high word, row TIME_HI, is kept per launch)
_ALIASES = {"spark_status": "spark_state", "time": "time_low32", "time_since_open_voltage": "time_since_open_voltage_low32"}  # (a trace records the 32-bit rows the kernels carry: traced "time" is the clock modulo 2**32, signed int32 bits)


class DeviceTrace:
    """Ring buffers + descriptor of one bound trace.  Create it with `WireEDMEnv.bind_trace`."""

    def __init__(self, env, signals: Iterable[str], *, every: int = 1, capacity: int = 1000,
                 envs: Optional[Tuple[int, int]] = None, wire_temperature: bool = False):
        self.env = env
        self.bind_step = int(getattr(env, "steps_since_reset", 0))  # physics steps since the last full reset, at the bind
        self.every, self.capacity = int(every), int(capacity)
        if self.every < 1 or self.capacity < 1:
            raise ValueError("every and capacity must be >= 1")
        lo, count = (0, env.num_envs) if envs is None else (int(envs[0]), int(envs[1]))
        if lo < 0 or count < 1 or lo + count > env.num_envs:
            raise ValueError(f"envs=(first, count) must lie inside [0, {env.num_envs})")
        self.env_lo, self.env_count = lo, count
        self.signals: List[str] = []
        rows = {"f64": set(), "i32": set(), "i8": set()}
        self._where: Dict[str, Tuple[str, int]] = {}
        self.wire_temperature = bool(wire_temperature)
        for name in signals:
            if name == "wire_temperature":
                self.wire_temperature = True
                continue
            field = _ALIASES.get(name, name)
            if field not in _FIELDS:
                raise ValueError(f"unknown signal {name!r}; traceable signals: {sorted(_FIELDS)} + 'wire_temperature'")
            block, row = _FIELDS[field]
            block = "i8" if block == "b" else block
            rows[block].add(int(row))
            self._where[name] = (block, int(row))
            self.signals.append(name)
        if not self.signals and not self.wire_temperature:
            raise ValueError("nothing to trace")
        dev = env.device
        dtypes = {"f64": torch.float64, "i32": torch.int32, "i8": torch.int8}
        self._rows = {b: sorted(r) for b, r in rows.items()}
        self._buf = {b: (torch.zeros((self.capacity, len(r), count), dtype=dtypes[b], device=dev) if r else None)
                     for b, r in self._rows.items()}
        self._T = (torch.zeros((self.capacity, env.n_segments, count), dtype=torch.float32, device=dev)
                   if self.wire_temperature else None)

        def mask(b):
            return sum(1 << r for r in self._rows[b])

        def ptr(t):
            return t.data_ptr() if t is not None else None

        self.desc = _abi.TraceDesc(ptr(self._buf["f64"]), ptr(self._buf["i32"]), ptr(self._buf["i8"]), ptr(self._T),
                                   mask("f64"), mask("i32"), mask("i8"), lo, count, self.every, self.capacity, 0)

    # ------------------------------------------------------------------ reading
    @property
    def count(self) -> int:
        """Samples written since the trace was bound (host-side counter, no device sync)."""
        return self.env._backend.trace_samples() if self.env._trace is self else 0

    def _slots(self, first: int, last: int) -> torch.Tensor:
        """Ring slots of samples number first..last-1 (0-based since bind), oldest first."""
        total = self.count
        if first < max(0, total - self.capacity):
            raise RuntimeError(
                f"trace overrun: sample {first} was overwritten (ring capacity {self.capacity}, {total} written)")
        idx = torch.arange(first, last, device=self.env.device)
        return idx % self.capacity

    def read_range(self, first: int, last: int, names: Optional[Iterable[str]] = None) -> Dict[str, torch.Tensor]:
        """Samples number ``first`` (inclusive) .. ``last`` (exclusive), counted from the bind."""
        last = min(int(last), self.count)
        first = max(0, min(int(first), last))
        slots = self._slots(first, last)
        out: Dict[str, torch.Tensor] = {}
        for name in (self.signals if names is None else names):
            if name == "wire_temperature":
                continue
            block, row = self._where[name]
            k = self._rows[block].index(row)
            t = self._buf[block][:, k, :].index_select(0, slots)
            if _FIELDS[_ALIASES.get(name, name)][0] == "b":
                t = t != 0
            out[name] = t
        if self._T is not None and (names is None or "wire_temperature" in names):
            out["wire_temperature"] = self._T.index_select(0, slots).permute(0, 2, 1)  # [sample, env, segment]
        return out

    def read(self, last: Optional[int] = None, names: Optional[Iterable[str]] = None) -> Dict[str, torch.Tensor]:
        """The newest ``last`` samples still in the ring (all of them by default), oldest first."""
        total = self.count
        n = min(total, self.capacity) if last is None else min(int(last), total, self.capacity)
        return self.read_range(total - n, total, names)

    def sample_times(self, first: int, last: int) -> torch.Tensor:
        """Microseconds since the bind at which samples first..last-1 were taken."""
        return (torch.arange(first, last, device=self.env.device) + 1) * self.every
