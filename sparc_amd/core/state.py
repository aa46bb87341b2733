"""``BatchedEDMState`` — the reference's ``EDMState`` (core/state.py:25-91) with a
leading batch dimension, stored struct-of-arrays for the kernels.

Three field-major blocks hold every mutable scalar (float64 / int32 / int8), environment-minor,
so that a wavefront's 64 lanes read 64 consecutive elements; the wire temperature (float32) is
quad-interleaved, ``T[n_seg / 4][stride][4]`` (four consecutive segments of one environment in one
16-byte word, include/wedm_hip.h), so that a lane moves its run of segments 16 bytes at a time.
Attribute access keeps
the reference's names: ``state.workpiece_position`` is a length-``num_envs`` tensor
view into the float64 block; assigning to it writes through
(``state.workpiece_position = 70.0`` works like it does on the reference's state).
"""
from __future__ import annotations

import torch

from .. import _abi
from .._abi import F64, I8, I32

# reference attribute name -> (block, row)
_FIELDS = {
    "workpiece_position": ("f64", F64.WORKPIECE_POS), "wire_position": ("f64", F64.WIRE_POS),
    "wire_velocity": ("f64", F64.WIRE_VEL), "wire_unwinding_velocity": ("f64", F64.UNWIND_VEL),
    "voltage": ("f64", F64.VOLTAGE), "current": ("f64", F64.CURRENT),
    "target_voltage": ("f64", F64.TARGET_VOLTAGE), "ON_time": ("f64", F64.ON_TIME),
    "OFF_time": ("f64", F64.OFF_TIME), "target_delta": ("f64", F64.TARGET_DELTA),
    "target_position": ("f64", F64.TARGET_POS),
    "debris_volume": ("f64", F64.DEBRIS_VOLUME), "debris_density": ("f64", F64.DEBRIS_DENSITY),
    "debris_concentration": ("f64", F64.DEBRIS_DENSITY),  # legacy alias, dielectric.py:159
    "cavity_volume": ("f64", F64.CAVITY), "flow_rate": ("f64", F64.FLOW),
    "last_crater_volume": ("f64", F64.LAST_CRATER), "spark_location": ("f64", F64.SPARK_Y),
    "wire_max_temperature": ("f64", F64.TMAX),
    # module-private state of the reference, exposed for inspection / scenario setup
    "prev_accel": ("f64", F64.PREV_ACCEL), "dielectric_last_gap": ("f64", F64.LAST_GAP),
    "dielectric_last_density": ("f64", F64.LAST_DENSITY), "wire_last_flow": ("f64", F64.WIRE_LAST_FLOW),
    "h_eff_base": ("f64", F64.H_BASE), "h_eff_zone": ("f64", F64.H_ZONE),
    # the driver's 1 ms voltage history as running sums (run_simulation.py:258-281; include/wedm_hip.h)
    "voltage_sum_since_control_step": ("f64", F64.VOLT_ACC), "voltage_sum_at_control_step": ("f64", F64.VOLT_SUM),
    # `state.time` / `state.time_since_open_voltage` are Python ints in the reference (wire_edm.py:135-137): here 64-bit,
    # the low 32 bits (unsigned) in the rows below + the shared high word TIME_HI, composed on access (`_WIDE`)
    "time_low32": ("i32", I32.TIME), "time_high32": ("i32", I32.TIME_HI), "time_since_servo": ("i32", I32.SINCE_SERVO),
    "time_since_open_voltage_low32": ("i32", I32.SINCE_OPEN_V),
    "time_since_spark_ignition": ("i32", I32.SINCE_IGNITION),
    "time_since_spark_end": ("i32", I32.SINCE_SPARK_END), "spark_duration": ("i32", I32.SPARK_DUR),
    "random_short_remaining": ("i32", I32.RANDOM_SHORT_REM),
    "debris_short_remaining": ("i32", I32.DEBRIS_SHORT_REM),
    "time_in_critical_temp": ("i32", I32.TIME_CRITICAL), "current_mode": ("i32", I32.CURRENT_MODE),
    "episode": ("i32", I32.EPISODE), "spark_count": ("i32", I32.SPARK_COUNT),
    "spark_state": ("i8", I8.SPARK_STATE),
    "is_short_circuit": ("b", I8.IS_SHORT), "is_wire_broken": ("b", I8.WIRE_BROKEN),
    "is_target_distance_reached": ("b", I8.TARGET_REACHED), "done": ("b", I8.DONE),
    "control_step": ("b", I8.CTRL_STEP), "error": ("b", I8.ERROR),
    "ignition_mode_cached": ("b", I8.MODE_CACHED),  # IgnitionModule._cached_current_mode is not None (ignition.py:79-81)
}


_BINARY_DUNDERS = ("add", "sub", "mul", "truediv", "radd", "rsub", "rmul", "eq", "ne", "lt", "le", "gt", "ge")
_INPLACE_DUNDERS = ("iadd", "isub", "imul", "itruediv")


def _is_basic_index(idx) -> bool:
    """Integers, slices, None and Ellipsis: indexing with these yields a VIEW of the indexed tensor."""
    items = idx if isinstance(idx, tuple) else (idx,)
    return all(i is None or i is Ellipsis or isinstance(i, (int, slice)) for i in items)


class WireTemperature:
    """``state.wire_temperature``: the ``[num_envs, n_seg]`` face of the quad-interleaved ``T`` block.

    The block's layout (``T[seg >> 2][env][seg & 3]``) cannot be expressed as ONE strided 2-D view, so this small
    proxy keeps the reference's indexing on both sides.

    * Reading gathers: ``wt.tensor()``, ``wt.cpu()``, ``np.asarray(wt)``, any torch function and every out-of-place
      method work on a fresh ``[num_envs, n_seg]`` COPY of the block.
    * Writing goes through to the block the kernels step, in every form the reference's NumPy array allows:
      ``wt[: n // 2, 180:186] = 1600.0``, ``wt.copy_(x)``, ``wt.fill_(v)``, ``wt += 5``, every in-place method
      (``add_``, ``zero_``, ``clamp_``, ``mul_`` ... — gather, operate, scatter), and CHAINED indexing:
      ``wt[e]`` / ``wt[2:5, 10:20]`` (integers and slices) is again a write-through proxy, so ``wt[1][10] = 999``
      and ``state.wire_temperature[2].fill_(500)`` change the block.  Indexing with a tensor or a list returns a plain
      gathered tensor (a copy, as advanced indexing does on a NumPy array too).
    * ``wt.quads`` is the zero-copy ``[num_envs, n_seg / 4, 4]`` view of the block itself; ``wt.tensor()`` is the way to get a
      plain tensor (``torch.as_tensor(wt)`` walks the sequence protocol element by element: correct, seconds for 200 x 400)."""

    def __init__(self, T: torch.Tensor, num_envs: int, n_seg: int, _chain=()):
        self._T, self._n, self._s, self._chain = T, num_envs, n_seg, tuple(_chain)

    @property
    def quads(self) -> torch.Tensor:
        return self._T[:, : self._n].permute(1, 0, 2)

    def _full(self) -> torch.Tensor:
        return self.quads.reshape(self._n, -1)[:, : self._s]

    @staticmethod
    def _descend(full: torch.Tensor, chain) -> torch.Tensor:
        for idx in chain:  # basic indices only: every step is a view of `full`
            full = full[idx]
        return full

    def tensor(self) -> torch.Tensor:
        """The selected cells as a fresh tensor (a copy: writing to it does not change the block)."""
        return self._descend(self._full(), self._chain)

    def _scatter(self, full: torch.Tensor) -> None:
        q = self.quads
        flat = q.reshape(self._n, -1)
        flat[:, : self._s] = full
        q.copy_(flat.reshape(q.shape))

    def _update(self, fn) -> None:
        """gather -> `fn(view of this proxy's cells inside the gathered copy)` -> scatter."""
        full = self._full()
        fn(self._descend(full, self._chain))
        self._scatter(full)

    shape = property(lambda self: self.tensor().shape if self._chain else torch.Size((self._n, self._s)))
    dtype = property(lambda self: self._T.dtype)
    device = property(lambda self: self._T.device)

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def __len__(self):
        return self.shape[0]

    def __iter__(self):
        # rows of ONE gathered copy (plain tensors): `torch.as_tensor(wt)`, `list(wt)`, `for row in wt` walk the sequence
        # protocol, and row proxies that each gather the whole block again made that quadratic (minutes for 200 x 400)
        return iter(self.tensor())

    def __getitem__(self, idx):
        if _is_basic_index(idx):
            sub = WireTemperature(self._T, self._n, self._s, self._chain + (idx,))
            return sub if sub.tensor().dim() > 0 else sub.tensor()  # a single cell: a 0-d tensor, as before
        return self.tensor()[idx]

    def __setitem__(self, idx, value) -> None:
        if isinstance(value, WireTemperature):
            value = value.tensor()

        def assign(view):
            view[idx] = value

        self._update(assign)

    def copy_(self, value):
        if isinstance(value, WireTemperature):
            value = value.tensor()
        self._update(lambda view: view.copy_(torch.as_tensor(value, dtype=torch.float32, device=self._T.device)))
        return self

    def fill_(self, value):
        self._update(lambda view: view.fill_(float(value)))
        return self

    def __array__(self, dtype=None, copy=None):
        a = self.tensor().cpu().numpy()
        return a if dtype is None else a.astype(dtype)

    def __getattr__(self, name):
        if name.startswith("__") or name in ("_T", "_n", "_s", "_chain"):
            raise AttributeError(name)
        if name.endswith("_") and not name.endswith("__") and callable(getattr(torch.Tensor, name, None)):
            # an in-place method (add_, zero_, clamp_, mul_, ...): on the gathered tensor it would change a temporary

            def inplace(*args, **kwargs):
                conv = lambda a: a.tensor() if isinstance(a, WireTemperature) else a
                self._update(lambda view: getattr(view, name)(*[conv(a) for a in args], **{k: conv(v) for k, v in kwargs.items()}))
                return self

            return inplace
        # everything else (cpu, numpy, clone, max, mean, ...) on the gathered tensor
        return getattr(self.tensor(), name)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        def conv(a):  # (also inside lists / tuples: torch.stack([wt[0], wt[1]]))
            if isinstance(a, cls):
                return a.tensor()
            return type(a)(conv(x) for x in a) if type(a) in (list, tuple) else a

        return func(*[conv(a) for a in args], **{k: conv(v) for k, v in (kwargs or {}).items()})


for _op in _BINARY_DUNDERS:
    setattr(WireTemperature, f"__{_op}__", (lambda op: lambda self, other: getattr(self.tensor(), f"__{op}__")(
        other.tensor() if isinstance(other, WireTemperature) else other))(_op))
for _op in _INPLACE_DUNDERS:  # `wt += 5`, `wt[3] *= 2`: gather, operate, scatter
    setattr(WireTemperature, f"__{_op}__", (lambda op: lambda self, other: getattr(self, {"iadd": "add_", "isub": "sub_", "imul": "mul_", "itruediv": "div_"}[op])(other))(_op))


# int64 attributes composed from a low-word row and the clock's high word (both clocks advance together, every step
# that is not a wire-break early return: wire_edm.py:129-137)
_WIDE = {"time": "time_low32", "time_since_open_voltage": "time_since_open_voltage_low32"}


class BatchedEDMState:
    def __init__(self, num_envs: int, n_seg_max: int, obs_dim: int, device, crater_log_capacity: int = 0):
        stride = (num_envs + 63) // 64 * 64
        object.__setattr__(self, "num_envs", num_envs)
        object.__setattr__(self, "n_seg_max", n_seg_max)
        object.__setattr__(self, "stride", stride)
        object.__setattr__(self, "device", device)
        kw = dict(device=device)
        object.__setattr__(self, "f64", torch.zeros((_abi.F64_COUNT, stride), dtype=torch.float64, **kw))
        object.__setattr__(self, "i32", torch.zeros((_abi.I32_COUNT, stride), dtype=torch.int32, **kw))
        object.__setattr__(self, "i8", torch.zeros((_abi.I8_COUNT, stride), dtype=torch.int8, **kw))
        object.__setattr__(self, "b", self.i8.view(torch.bool))  # zero-copy 0/1 view of the flags
        object.__setattr__(self, "T", torch.zeros((_abi.t_quads(n_seg_max), stride, 4), dtype=torch.float32, **kw))
        object.__setattr__(self, "obs", torch.zeros((max(obs_dim, 1), stride), dtype=torch.float32, **kw))
        # running statistics the kernels accumulate (material.py:207-227)
        object.__setattr__(self, "stats", torch.zeros((_abi.STAT_COUNT, stride), dtype=torch.float64, **kw))
        # per-launch reward written by the kernels when wedm_params.reward_mode != 0
        object.__setattr__(self, "reward", torch.zeros((1, stride), dtype=torch.float32, **kw))
        # `MaterialRemovalModule.crater_volumes_um3` (material.py:133) as a ring per environment, optional
        object.__setattr__(self, "crater_log", torch.zeros((crater_log_capacity, stride), dtype=torch.float64, **kw)
                           if crater_log_capacity > 0 else None)
        # read-only attributes computed on access (registered by the environment):
        # dielectric_flow_rate (dielectric.py:160-162), wire_average_temperature (wire.py:339-347)
        object.__setattr__(self, "derived", {})
        object.__setattr__(self, "_views", {})

    # ------------------------------------------------------------------ views
    def _view(self, name: str) -> torch.Tensor:
        # the blocks are never reallocated, so a field's view is built once (a per-microsecond
        # caller would otherwise pay ~2 us of tensor indexing per field and step)
        cache = self.__dict__["_views"]
        v = cache.get(name)
        if v is None:
            block, row = _FIELDS[name]
            v = cache[name] = getattr(self, block)[int(row), : self.num_envs]
        return v

    def __getattr__(self, name: str):
        if name in _FIELDS:
            return self._view(name)
        if name == "time":  # exact 64-bit clock: (high word << 32) | unsigned low word
            lo = self._view("time_low32").to(torch.int64) & 0xFFFFFFFF
            return (self._view("time_high32").to(torch.int64) << 32) | lo
        if name == "time_since_open_voltage":
            # Its row holds the low 32 bits only, and TIME_HI is the high word of `time`.  Both clocks advance together
            # (every step that is not a wire-break early return, wire_edm.py:135-137), so their distance is constant and the
            # exact value is `time` minus that distance -- NOT `(TIME_HI << 32) | low word`, which is wrong as soon as the
            # two low words wrap at different moments (time = 2**32 - 100 assigned, 300 us later: 300, not 2**32 + 300).
            t = self.time
            d = (self._view("time_low32").to(torch.int64) - self._view("time_since_open_voltage_low32").to(torch.int64)) & 0xFFFFFFFF
            return t - d
        derived = self.__dict__.get("derived", {})
        if name in derived:
            return derived[name]()
        raise AttributeError(name)

    def __setattr__(self, name: str, value) -> None:
        if name in _WIDE:
            v = torch.as_tensor(value, dtype=torch.int64, device=self.device).expand(self.num_envs)
            lo = v & 0xFFFFFFFF
            if name == "time_since_open_voltage":
                # only the low word is stored (see __getattr__): representable are the values 0 <= time - v < 2**32; the
                # high word belongs to `time` and is left alone (assigning 7 here used to drop `time` from 2**32 + 200 to 200)
                d = self.time - v
                if bool(((d < 0) | (d >= 2**32)).any().item()):
                    raise ValueError("time_since_open_voltage must lie in (time - 2**32, time]: it is stored as a 32-bit "
                                     "distance to `time` (assign `time` first)")
            else:
                since = self.time_since_open_voltage  # keeps its value: its distance to `time` moves with `time`
                self._view("time_high32").copy_((v >> 32).to(torch.int32))
            self._view(_WIDE[name]).copy_(torch.where(lo >= 2**31, lo - 2**32, lo).to(torch.int32))
            if name == "time":
                back = since.clamp(max=v).clamp(min=v - (2**32 - 1))  # (a distance beyond 32 bits cannot be kept: nearest)
                blo = back & 0xFFFFFFFF
                self._view("time_since_open_voltage_low32").copy_(torch.where(blo >= 2**31, blo - 2**32, blo).to(torch.int32))
        elif name in _FIELDS:
            view = self._view(name)
            if torch.is_tensor(value):
                view.copy_(value.to(view.device))
            else:
                view.copy_(torch.as_tensor(value, dtype=view.dtype, device=view.device).expand_as(view))
        elif name == "wire_temperature":
            self.wire_temperature.copy_(value.tensor() if isinstance(value, WireTemperature) else value)
        else:
            raise AttributeError(f"BatchedEDMState has no writable field {name!r}")

    @property
    def wire_temperature(self) -> WireTemperature:
        """``[num_envs, n_seg]`` face of the quad-interleaved block: reads gather, writes go through (`WireTemperature`)."""
        return WireTemperature(self.T, self.num_envs, self.n_seg_max)

    @property
    def spark_status(self):
        """(state int8[N], y float64[N] with NaN for None, duration int32[N])."""
        return self._view("spark_state"), self._view("spark_location"), self._view("spark_duration")

    def pointers(self, with_obs: bool) -> _abi.StatePtrs:
        return _abi.StatePtrs(self.f64.data_ptr(), self.i32.data_ptr(), self.i8.data_ptr(), self.T.data_ptr(),
                              self.obs.data_ptr() if with_obs else None, self.stride, self.stats.data_ptr(),
                              self.reward.data_ptr(),
                              self.crater_log.data_ptr() if self.crater_log is not None else None,
                              self.crater_log.shape[0] if self.crater_log is not None else 0)

    def field_names(self):
        return tuple(_FIELDS) + tuple(_WIDE)

    def clone_blocks(self):
        """Host copies of the raw blocks (used by checkpointing and by the parity tests)."""
        out = {k: getattr(self, k).detach().cpu().clone() for k in ("f64", "i32", "i8", "T", "obs", "stats", "reward")}
        if self.crater_log is not None:  # the ring spark_count indexes (material.py:133): part of the state
            out["crater_log"] = self.crater_log.detach().cpu().clone()
        return out

    def load_blocks(self, blocks) -> None:
        for k in ("f64", "i32", "i8", "T", "obs", "stats", "reward", "crater_log"):
            if k in blocks and getattr(self, k) is not None:
                if tuple(blocks[k].shape) != tuple(getattr(self, k).shape):
                    raise ValueError(f"state block {k!r} has shape {tuple(blocks[k].shape)}, this environment's is "
                                     f"{tuple(getattr(self, k).shape)} (another ABI version or crater_log_capacity?)")
                getattr(self, k).copy_(blocks[k].to(self.device))
