"""Batched ``WireEDMEnv`` — the reference's Gymnasium surface
(envs/wire_edm.py:16-201) with a leading batch dimension, advanced by one fused HIP
kernel on an MI355X.

Same constructor keywords, same ``reset(seed=, options=)`` / ``step(action)`` tuples,
same ``EnvironmentConfig`` and ``*ModuleParameters``; additionally ``num_envs`` and
``device``.  Every per-environment scalar of the reference becomes a length-N tensor
(``env.state.workpiece_position`` ...), ``terminated``/``truncated`` are ``bool[N]``,
``info`` carries the reference's five keys as tensors.

Documented deviations from the single-environment reference:
  * ``reset`` also resets module-private state (short timers, debris, caches,
    ``prev_accel``); the reference leaks it across episodes;
  * a terminated environment is frozen until it is reset;
  * a ``current_mode`` without crater data raises ``ValueError`` when the action is
    passed to ``step`` (the reference raises at the first fresh spark after the latch);
  * ``obs`` is a fixed float32 vector (the reference returns ``{}``), refreshed at
    control steps; ``reward`` is 0 as in the reference;
  * randomness is a counter-based Philox4x32-10 stream per environment, not NumPy's
    PCG64: same distributions, different variates for the same seed.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, Optional

import numpy as np
import torch

from .. import _abi
from ..core import derive
from ..core.env_config import EnvironmentConfig
from ..core.material_db import get_material_db
from ..core.state import BatchedEDMState
from ..core.tables import CRATER, MAX_MODE, MODE_CURRENT, VALID_CRATER_MODES
from ..modules.parameters import (
    DielectricModuleParameters,
    IgnitionModuleParameters,
    MaterialModuleParameters,
    MechanicsModuleParameters,
    WireModuleParameters,
)


class Box:
    """Minimal stand-in for ``gymnasium.spaces.Box`` (gymnasium is optional)."""

    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)

    def sample(self, rng: Optional[np.random.Generator] = None):
        rng = rng or np.random.default_rng()
        if np.issubdtype(self.dtype, np.integer):
            return rng.integers(int(self.low), int(self.high) + 1, size=self.shape).astype(self.dtype)
        return rng.uniform(self.low, self.high, size=self.shape).astype(self.dtype)

    def __repr__(self):
        return f"Box({self.low}, {self.high}, {self.shape}, {self.dtype})"


class DictSpace(dict):
    """Minimal stand-in for ``gymnasium.spaces.Dict``."""

    def sample(self, rng: Optional[np.random.Generator] = None):
        return {k: v.sample(rng) for k, v in self.items()}


class StepInfo(dict):
    """``info`` of `step`: the reference's five keys (wire_edm.py:150-156) as tensors over the batch.  ``"time"`` is the
    exact 64-bit clock (`state.time`), composed from its two 32-bit rows when it is READ -- a per-microsecond loop that
    never looks at it pays nothing.  (`env.state.time_low32` is the zero-copy row itself -- the signed int32 bits of the
    clock modulo 2**32 -- for loops that want no op at all; the in-kernel trace records that row too.)"""

    __slots__ = ("_state",)

    def __init__(self, base, state):
        super().__init__(base)  # ("time" is a placeholder entry: membership and key order of the reference's dict)
        self._state = state

    def __getitem__(self, key):
        return self._state.time if key == "time" else dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in self]

    def values(self):
        return [self[k] for k in self]

    def copy(self):
        return StepInfo({k: dict.__getitem__(self, k) for k in self}, self._state)


class DeviceAction:
    """An action already laid out for the kernel: five contiguous length-N device
    tensors (float64 x4, int32).  Build it once with ``env.make_action`` and pass it
    to ``step`` to avoid per-step host->device conversion."""

    __slots__ = ("servo", "target_voltage", "on_time", "off_time", "current_mode", "ptrs")

    def __init__(self, servo, target_voltage, on_time, off_time, current_mode):
        self.servo, self.target_voltage, self.on_time, self.off_time = servo, target_voltage, on_time, off_time
        self.current_mode = current_mode
        self.ptrs = _abi.ActionPtrs(servo.data_ptr(), target_voltage.data_ptr(), on_time.data_ptr(),
                                    off_time.data_ptr(), current_mode.data_ptr())


class WireEDMEnv:
    """Main-cut Wire-EDM environment, N environments in lock-step (1 us base step,
    1 ms control step).  See the module docstring."""

    metadata = {"render_modes": ["human"], "render_fps": 300}

    def __init__(
        self,
        *,
        num_envs: int = 1,
        device: Any = None,
        render_mode: Optional[str] = None,
        mechanics_control_mode: str = "position",
        config: Optional[EnvironmentConfig] = None,
        ignition_params: Optional[IgnitionModuleParameters] = None,
        wire_params: Optional[WireModuleParameters] = None,
        material_params: Optional[MaterialModuleParameters] = None,
        dielectric_params: Optional[DielectricModuleParameters] = None,
        mechanics_params: Optional[MechanicsModuleParameters] = None,
        workpiece_height=None,
        wire_diameter=None,
        env_id_offset: int = 0,
        disable_ignition: bool = False,
        strict_actions: bool = True,
        autoreset: bool = False,
        reward: Optional[str] = None,
        reward_break_penalty: float = 10.0,
        stencil_dtype: str = "float32",
        crater_log_capacity: int = 0,
        reset_semantics: str = "full",
        freeze_terminated: bool = True,
        backend: Optional[Callable] = None,
    ):
        """Beyond the reference's keywords (wire_edm.py:22-34):

        ``autoreset``: next-step autoreset inside the launch — an environment found terminated when a
        step begins is reset by that kernel launch (as ``reset(options={"mask": done})`` would: next
        Philox episode, fresh module state) and then stepped; no host round trip.
        ``reward``: None = the reference's constant 0.0 (its `_calculate_reward` is a TODO,
        wire_edm.py:185-187); ``"progress"`` = micrometres the workpiece front advanced during the
        launch minus ``reward_break_penalty`` if the wire broke, written by the kernels (float32).
        ``stencil_dtype``: ``"float32"`` = the wire stencil exactly as the reference evaluates
        wire.py:58-123 without Numba (NumPy-2 scalar promotion: float32 op for op); ``"float64"`` =
        as Numba types the same lines (float64 expressions rounded at each float32 store).
        ``crater_log_capacity``: keep the last that many sampled crater volumes of every environment
        (`MaterialRemovalModule.crater_volumes_um3`, material.py:133) — see `get_crater_volumes`.
        ``reset_semantics``: ``"full"`` = a reset is a fresh environment, module-private state included (default);
        ``"reference"`` = exactly what the reference's ``reset()`` does (wire_edm.py:106-114): only ``EDMState`` is
        re-initialised, the module objects live on — ignition short timers and current cache, debris volume and
        the flow / density caches, ``prev_accel``, convection cache and coefficients, crater list and statistics
        carry over into the next episode (what every RL loop on the reference sees from its second episode on).
        ``freeze_terminated``: True = a terminated environment is frozen until it is reset (default); False = it
        keeps being stepped as the reference does when ``step()`` is called after ``terminated`` (wire_edm.py:116-157
        has no guard): after a wire break the wire module returns at once and the step returns before mechanics
        and clocks, after the cutting target everything goes on; ``terminated`` then repeats what the reference's
        ``step()`` returns."""
        self.render_mode = render_mode
        if mechanics_control_mode not in ["position", "velocity"]:
            raise ValueError(f"mechanics_control_mode must be 'position' or 'velocity', got {mechanics_control_mode}")
        self.mechanics_control_mode = mechanics_control_mode
        if int(num_envs) <= 0:
            raise ValueError("num_envs must be positive")
        self.num_envs = int(num_envs)

        self.config = config or EnvironmentConfig()
        self.config.validate()
        self.dt = self.config.dt
        self.servo_interval = self.config.servo_interval

        self.ignition_params = ignition_params or IgnitionModuleParameters()
        self.wire_params = wire_params or WireModuleParameters()
        self.material_params = material_params or MaterialModuleParameters()
        self.dielectric_params = dielectric_params or DielectricModuleParameters()
        self.mechanics_params = mechanics_params or MechanicsModuleParameters()
        self.wire_material = get_material_db().get_wire_material(self.config.wire_material)

        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.np_random = np.random.default_rng()
        self.strict_actions = bool(strict_actions)
        self.env_id_offset = int(env_id_offset)
        self.autoreset = bool(autoreset)
        if reward not in (None, "progress"):
            raise ValueError("reward must be None (the reference's constant 0.0) or 'progress'")
        self.reward_kind = reward
        if reset_semantics not in ("full", "reference"):
            raise ValueError("reset_semantics must be 'full' or 'reference'")
        self.reset_semantics, self.freeze_terminated = reset_semantics, bool(freeze_terminated)
        if stencil_dtype not in ("float32", "float64"):
            raise ValueError("stencil_dtype must be 'float32' or 'float64'")
        self.stencil_dtype = stencil_dtype

        # ---- geometry: uniform (reference behaviour) or one (h, d) pair per environment
        stride = (self.num_envs + 63) // 64 * 64
        self.per_env_geometry = workpiece_height is not None or wire_diameter is not None
        if self.per_env_geometry:
            h = np.broadcast_to(np.asarray(self.config.workpiece_height if workpiece_height is None
                                           else _to_numpy(workpiece_height), dtype=np.float64), (self.num_envs,))
            d = np.broadcast_to(np.asarray(self.config.wire_diameter if wire_diameter is None
                                           else _to_numpy(wire_diameter), dtype=np.float64), (self.num_envs,))
            gf, gi, n_seg_max = derive.geometry_rows(h, d, self.wire_params, self.wire_material,
                                                     self.material_params, stride)
            self.geometry = None
            self._geom_f64 = torch.from_numpy(gf).to(self.device)
            self._geom_i32 = torch.from_numpy(gi).to(self.device)
            self.n_segments = n_seg_max
        else:
            self.geometry = derive.derive_geometry(self.config.workpiece_height, self.config.wire_diameter,
                                                   self.wire_params, self.wire_material, self.material_params)
            self.n_segments = self.geometry.n_seg
        self.params = derive.build_params(
            self.config, mechanics_control_mode, self.ignition_params, self.wire_params, self.material_params,
            self.dielectric_params, self.mechanics_params, self.wire_material, geometry=self.geometry,
            env_id_offset=self.env_id_offset, obs_dim=_abi.OBS_DIM, disable_ignition=disable_ignition,
            autoreset=self.autoreset, reward_mode=1 if reward == "progress" else 0,
            reward_break_penalty=reward_break_penalty, stencil_mode=1 if stencil_dtype == "float64" else 0,
            reset_semantics=1 if reset_semantics == "reference" else 0, keep_stepping_terminated=not freeze_terminated)

        # ---- state (caller-owned memory) + backend
        self.state = BatchedEDMState(self.num_envs, self.n_segments, _abi.OBS_DIM, self.device,
                                     crater_log_capacity=int(crater_log_capacity))
        from ..utils.logger import dielectric_flow_rate

        base_flow = float(self.dielectric_params.base_flow_rate)
        self.state.derived["dielectric_flow_rate"] = lambda: dielectric_flow_rate(self.state.flow_rate, base_flow)
        if not self.per_env_geometry:
            self.state.derived["wire_average_temperature"] = self.zone_mean_temperature
        if backend is None:
            from .._lib import HipBackend

            backend = HipBackend
        self._backend = backend(self.params, self.num_envs, self.n_segments, self.device)
        self._backend.bind_state(self.state.pointers(with_obs=True))
        if self.per_env_geometry:
            self._backend.bind_geometry(_abi.GeomPtrs(self._geom_f64.data_ptr(), self._geom_i32.data_ptr()))

        # what remains of the reference's module objects: parameters + read-only helpers
        from ..modules.views import DielectricView, IgnitionView, MaterialView, MechanicsView, WireView

        self.ignition = IgnitionView(self)
        self.wire = WireView(self)
        self.material = MaterialView(self)
        self.dielectric = DielectricView(self)
        self.mechanics = MechanicsView(self)
        self.modules = {"ignition": self.ignition, "material": self.material, "dielectric": self.dielectric,
                        "wire": self.wire, "mechanics": self.mechanics}

        # ---- spaces (wire_edm.py:84-101); obs is this build's fixed vector
        self.action_space = DictSpace({
            "servo": Box(-1.0, 1.0, (1,), np.float32),
            "generator_control": DictSpace({
                "target_voltage": Box(0.0, 200.0, (1,), np.float32),
                "current_mode": Box(1, 19, (1,), np.int32),
                "ON_time": Box(0.0, 5.0, (1,), np.float32),
                "OFF_time": Box(0.0, 100.0, (1,), np.float32),
            }),
        })
        self.observation_space = Box(-np.inf, np.inf, (_abi.OBS_DIM,), np.float32)
        self.single_action_space = self.action_space
        self.single_observation_space = self.observation_space

        self._reward = self.state.reward[0, : self.num_envs]  # zeros unless reward="progress" (written by the kernels)
        self._truncated = torch.zeros(self.num_envs, dtype=torch.bool, device=self.device)
        self._mask_buf = None
        # current modes with crater data (material.py:108-113) as a device-side lookup table: index mode, clamped to [0, 20]
        self._valid_modes_dev = torch.tensor([m in VALID_CRATER_MODES for m in range(MAX_MODE + 2)], dtype=torch.bool).to(self.device)
        self._step_out = None
        self._trace = None
        self._seed = int.from_bytes(os.urandom(8), "little")
        self.steps_since_reset = 0  # host-side count of physics steps since the last reset of ALL environments
        self._backend.reset(None, self._seed, True, fresh=True)  # fresh module objects, whatever `reset_semantics`

    # ------------------------------------------------------------------ Gym API
    def reset(self, *, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None):
        """``WireEDMEnv.reset`` (wire_edm.py:106-114).  ``options={"mask": bool[N]}``
        resets only the selected environments."""
        mask_ptr = None
        if options and options.get("mask") is not None:
            mask = torch.as_tensor(options["mask"]).to(self.device).reshape(-1).to(torch.uint8).contiguous()
            if mask.numel() != self.num_envs:
                raise ValueError("options['mask'] must have one entry per environment")
            self._mask_buf = mask  # keep alive until the launch has consumed it
            mask_ptr = mask.data_ptr()
        if seed is not None:
            self._seed = int(seed)
            self.np_random = np.random.default_rng(seed)
            self._backend.reset(mask_ptr, self._seed, True)
        else:
            self._backend.reset(mask_ptr, self._seed, False)
        if mask_ptr is None:
            self.steps_since_reset = 0
        return self._get_obs(), {}

    def step(self, action):
        """One 1-us physics step for every environment (wire_edm.py:116-157)."""
        return self.step_many(action, 1)

    def step_many(self, action, n_substeps: int):
        """``n_substeps`` consecutive ``step(action)`` calls in ONE fused kernel launch."""
        act = self._prepare_action(action)
        self._last_action = act  # keep the tensors alive while the launch is in flight
        n_substeps = int(n_substeps)
        # (`state.time` is 64-bit: the kernels carry its low word and bump the high word once per launch, so a single
        # launch may not advance an environment by 2**31 us or more; nothing limits the number of launches)
        if n_substeps * self.dt >= 2**31:
            raise ValueError(f"one launch may advance an environment by less than 2**31 us (asked: {n_substeps} x {self.dt} us)")
        self._backend.step(n_substeps, act.ptrs)
        self.steps_since_reset += n_substeps
        # obs / done / info are views of caller-owned memory the kernel has just (asynchronously)
        # updated: built once, handed out every step (`info` is a fresh dict of the same tensors)
        out = self._step_out
        if out is None:
            st = self.state
            out = self._step_out = (self._get_obs(), st.done, {
                "wire_broken": st.is_wire_broken,
                "target_reached": st.is_target_distance_reached,
                "spark_state": st.spark_state,
                "time": None,  # the exact int64 clock, composed when it is read (StepInfo)
                "control_step": st.control_step,
            })
        return out[0], self._reward, out[1], self._truncated, StepInfo(out[2], self.state)

    def step_control(self, action):
        """One control interval (``servo_interval`` physics steps, default 1000)."""
        return self.step_many(action, -(-self.servo_interval // self.dt))  # ceil: the latch tests `time_since_servo >= servo_interval`

    def close(self) -> None:
        self._backend.close()

    # ------------------------------------------------------------------ helpers
    def make_action(self, servo=0.0, target_voltage=80.0, current_mode=5, ON_time=3.0, OFF_time=80.0) -> DeviceAction:
        return self._prepare_action({
            "servo": servo,
            "generator_control": {"target_voltage": target_voltage, "current_mode": current_mode,
                                  "ON_time": ON_time, "OFF_time": OFF_time},
        })

    def _leaf(self, value, dtype) -> torch.Tensor:
        if torch.is_tensor(value):
            t = value.to(device=self.device, dtype=dtype).reshape(-1)
        else:
            arr = np.asarray(value)
            t = torch.from_numpy(np.ascontiguousarray(arr.reshape(-1))).to(device=self.device, dtype=dtype)
        if t.numel() == 1:
            t = t.expand(self.num_envs)
        elif t.numel() != self.num_envs:
            raise ValueError(f"action leaf has {t.numel()} values, expected 1 or num_envs={self.num_envs}")
        return t.contiguous()

    def _prepare_action(self, action) -> DeviceAction:
        if isinstance(action, DeviceAction):
            # the kernel reads num_envs elements through raw pointers: never let a foreign action through
            if action.servo.device != self.device or action.servo.numel() != self.num_envs:
                raise ValueError("DeviceAction belongs to another environment (device or num_envs differ); "
                                 "build it with this environment's make_action()")
            return action
        gc = action["generator_control"]
        mode = gc["current_mode"]
        if self.strict_actions:
            self._validate_modes(mode)
        return DeviceAction(
            self._leaf(action["servo"], torch.float64),
            self._leaf(gc["target_voltage"], torch.float64),
            self._leaf(gc["ON_time"], torch.float64),
            self._leaf(gc["OFF_time"], torch.float64),
            self._leaf(mode, torch.int32),
        )

    def _validate_modes(self, mode) -> None:
        """Mirror of material.py:108-113: modes without crater data are an error.

        Host-side values (Python numbers, NumPy arrays, CPU tensors) are checked here and raise at once.  A tensor that
        already lives on the device is checked ON the device and never read back: a policy that emits fresh device
        tensors every control step must not block on the previous launch (a `.cpu()` here waited for the 4-6 ms
        launch still in flight).  An invalid entry sets the environment's sticky ERROR flag -- the row the kernels
        set at the first fresh spark with such a mode -- and `check_errors()` raises for it (deferred raise)."""
        if torch.is_tensor(mode) and mode.device.type != "cpu":
            m = mode.to(self.device).reshape(-1).to(torch.int64)
            bad_dev = ~self._valid_modes_dev[torch.clamp(m, 0, MAX_MODE + 1)]  # (table built at construction: no upload here)
            if bad_dev.numel() == 1:
                bad_dev = bad_dev.expand(self.num_envs)
            elif bad_dev.numel() != self.num_envs:
                raise ValueError(f"action leaf has {bad_dev.numel()} values, expected 1 or num_envs={self.num_envs}")
            self.state.error.logical_or_(bad_dev)  # stream-ordered before the launch that latches the action
            return
        vals = mode.detach().numpy() if torch.is_tensor(mode) else np.asarray(mode)
        bad = sorted({int(v) for v in np.unique(vals.reshape(-1)) if int(v) not in VALID_CRATER_MODES})
        if bad:
            raise ValueError(
                f"Current mode I{bad[0]} is not available in crater data. "
                f"Available modes: {[f'I{m}' for m in VALID_CRATER_MODES]}"
            )

    def _get_obs(self) -> torch.Tensor:
        """``float32[num_envs, 8]``: gap, wire_velocity, voltage, current, spark_state,
        debris_density, flow_rate, max wire temperature — as of each environment's last
        control step (the reference's ``_get_obs`` is a TODO, wire_edm.py:181-183)."""
        return self.state.obs[:, : self.num_envs].t()

    def check_errors(self) -> None:
        """Synchronising check of the sticky per-environment error flag."""
        if bool(self.state.error.any().item()):
            idx = int(torch.nonzero(self.state.error)[0].item())
            raise ValueError(f"environment {idx}: a current mode that has no crater data was latched / passed in a device "
                             f"tensor (material.py:108-113 raises at the first fresh spark with it). "
                             f"Available modes: {[f'I{m}' for m in VALID_CRATER_MODES]}")

    def set_kernel(self, variant: int, lanes: int = 0) -> None:
        """0 = auto, 1 = global-memory stencil, 2 = LDS predicated, 3 = LDS fused, 4 = LDS
        fused with packed float32 math, 5 = global-memory stencil split over four waves, 6 =
        stream kernel (5 / 6: single microseconds; auto picks between them by shape), 7 = register
        kernel (one environment per lane, its whole wire in registers: wires of at most 128 segments,
        uniform geometry), 8 = wide register kernel (4 / 8 / 16 lanes per environment with 32 cells each in
        registers: wires of 9 to 512 segments, uniform geometry; auto picks it for fused launches of small
        batches), 9 = served kernel (kernel 4's walk with the float64 scalar physics of a block's environments on a wave
        of its own, one microsecond ahead of the walking waves: 4 or 8 lanes per environment, uniform geometry);
        10 = kernel 2's cell-by-cell form by name (kernel 2 is its packed form wherever the stencil is float32), 11 = the served
        form of kernel 2, 12 = the served form of kernel 7; ``lanes`` lanes per environment for 2/3/4/6/8/9/10/11 (0 = auto).
        All variants are bit-identical.  With ``stencil_dtype="float64"`` kernels 1, 2 / 10, 3, 6 (single microseconds without a
        trace sample), 7 and 8 accept the launch (the others raise ``WEDM_ERR_UNSUPPORTED``)."""
        self._backend.set_kernel(variant)
        if hasattr(self._backend, "set_lanes"):
            self._backend.set_lanes(lanes)

    def bind_trace(self, signals, *, every: int = 1, capacity: int = 1000, envs=None, wire_temperature: bool = False):
        """Record `signals` (EDMState attribute names, plus ``"wire_temperature"``) of the
        environments ``envs=(first, count)`` (default: all) every ``every`` microseconds INSIDE
        the step kernels, into a ring of ``capacity`` samples — what the reference does with
        `SimulationLogger.collect` after each 1-us step (utils/logger.py:110-160).  Returns the
        `DeviceTrace`; a previously bound trace is replaced."""
        from ..trace import DeviceTrace

        trace = DeviceTrace(self, signals, every=every, capacity=capacity, envs=envs, wire_temperature=wire_temperature)
        self._backend.bind_trace(trace.desc)
        self._trace = trace
        return trace

    def bind_rng_replay(self, table) -> None:
        """Validation mode: feed the environments caller-provided variates instead of their Philox streams —
        e.g. the draws a native-seed run of the reference made from its NumPy ``Generator(PCG64)``
        (``env.np_random``; call sites ignition.py:233,239,261,327, material.py:127).  ``table`` is
        ``float64[n_steps, 5]`` (the same variates for every environment) or ``float64[n_steps, 5, num_envs]``:
        per physics step since the reset the debris-short roll, the random-short roll, the ignition roll, the
        spark location [mm] and the crater volume [um^3], NaN where nothing is drawn (`_abi.REPLAY_SLOTS`).
        ``None`` returns to Philox.  Runs on the global-memory kernel."""
        if table is None:
            self._backend.bind_rng_replay(None, 0)
            self._replay = None
            return
        t = torch.as_tensor(table, dtype=torch.float64)
        if t.dim() == 2:
            t = t.unsqueeze(-1).expand(-1, -1, self.num_envs)
        if t.dim() != 3 or t.shape[1] != _abi.REPLAY_SLOTS or t.shape[2] != self.num_envs:
            raise ValueError(f"table must be [n_steps, {_abi.REPLAY_SLOTS}] or [n_steps, {_abi.REPLAY_SLOTS}, num_envs]")
        buf = torch.full((t.shape[0], _abi.REPLAY_SLOTS, self.state.stride), float("nan"), dtype=torch.float64,
                         device=self.device)
        buf[:, :, : self.num_envs] = t.to(self.device)
        self._replay = buf  # keep alive: the library only borrows the pointer
        self._backend.bind_rng_replay(buf.data_ptr(), int(t.shape[0]))

    def unbind_trace(self) -> None:
        self._backend.bind_trace(None)
        self._trace = None

    # ---- checkpoint / resume (the reference has none for simulation state) ---------
    def _physics_fingerprint(self) -> str:
        """sha256 over everything that determines the physics of a continuation: the whole
        `wedm_params` block the kernels receive (configuration, module parameters, derived constants,
        control mode, shard offset) and, with per-environment geometry, the geometry rows."""
        import ctypes
        import hashlib

        h = hashlib.sha256(ctypes.string_at(ctypes.addressof(self.params), ctypes.sizeof(self.params)))
        if self.per_env_geometry:
            h.update(self._geom_f64.cpu().numpy().tobytes())
            h.update(self._geom_i32.cpu().numpy().tobytes())
        return h.hexdigest()

    def state_dict(self) -> Dict[str, Any]:
        """Everything a bit-identical continuation needs: the raw state blocks (Philox key, episode
        and clocks live in them, so the random streams resume exactly), the reset seed, and a
        fingerprint of the physics parameters (tensors, ints and strings only: loads with
        ``weights_only=True``)."""
        return {"abi_version": _abi.ABI_VERSION, "blocks": self.state.clone_blocks(), "seed": self._seed, "num_envs": self.num_envs,
                "n_segments": self.n_segments, "env_id_offset": self.env_id_offset,
                "steps_since_reset": self.steps_since_reset, "physics": self._physics_fingerprint()}

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        if sd.get("abi_version") != _abi.ABI_VERSION:
            raise ValueError(f"checkpoint was written with state layout ABI {sd.get('abi_version', '<= 3')}, this build is ABI "
                             f"{_abi.ABI_VERSION} (rows and the wire-temperature layout differ): it cannot be continued here")
        if (sd["num_envs"], sd["n_segments"], sd["env_id_offset"]) != (self.num_envs, self.n_segments, self.env_id_offset):
            raise ValueError("checkpoint was taken from an environment of a different shape / shard")
        if sd.get("physics") != self._physics_fingerprint():
            raise ValueError("checkpoint was taken with different physics (configuration, module parameters, control "
                             "mode or per-environment geometry): continuing would silently change the trajectory")
        self.state.load_blocks(sd["blocks"])
        self._seed = int(sd["seed"])
        self.steps_since_reset = int(sd["steps_since_reset"])

    def save_checkpoint(self, path) -> None:
        torch.save(self.state_dict(), path)

    def load_checkpoint(self, path) -> None:
        self.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))

    def zone_mean_temperature(self) -> torch.Tensor:
        """Mean wire temperature over the workpiece zone (wire.py:390-398), per environment."""
        if self.geometry is None:
            raise NotImplementedError("zone mean needs uniform geometry")
        g = self.geometry
        lo, hi = g.az_start, g.az_end
        T = self.state.wire_temperature.tensor().t()  # [segment, env], as the reduction was written for ABI v3
        if hi > lo:
            return T[lo:hi].mean(dim=0)
        return T[: g.n_seg].mean(dim=0)

    # ---- statistics the reference's modules expose -------------------------
    def get_short_circuit_status(self) -> Dict[str, torch.Tensor]:
        """`IgnitionModule.get_short_circuit_status` (ignition.py:386-399), per environment."""
        r, d = self.state.random_short_remaining, self.state.debris_short_remaining
        return {"has_random_short": r > 0, "random_short_remaining_us": r, "has_debris_short": d > 0,
                "debris_short_remaining_us": d, "total_short_remaining_us": torch.maximum(r, d)}

    def get_debris_statistics(self) -> Dict[str, torch.Tensor]:
        """`DielectricModule.get_debris_statistics` (dielectric.py:174-182), per environment."""
        st = self.state
        return {"debris_volume_mm3": st.debris_volume, "debris_density": st.debris_density,
                "cavity_volume_mm3": st.cavity_volume, "flow_condition": st.flow_rate,
                "debris_fill_percentage": st.debris_density * 100.0}

    def get_crater_statistics(self) -> Dict[str, torch.Tensor]:
        """`MaterialRemovalModule.get_crater_statistics` (material.py:207-227), per environment, from
        the running sum / sum of squares / min / max the kernels keep at every fresh spark (the
        list of all volumes, ``volumes_um3``, is not kept; trace ``last_crater_volume`` to get it).
        Zeros while an environment has had no crater, as in the reference."""
        from .._abi import STAT

        st, n = self.state.stats[:, : self.num_envs], self.state.spark_count
        none = n == 0
        denom = torch.clamp(n, min=1).to(torch.float64)
        mean = st[STAT.CRATER_SUM] / denom
        var = torch.clamp(st[STAT.CRATER_SUMSQ] / denom - mean * mean, min=0.0)
        zero = torch.zeros_like(mean)
        return {"total_craters": n, "mean_volume_um3": mean, "std_volume_um3": torch.sqrt(var),
                "min_volume_um3": torch.where(none, zero, st[STAT.CRATER_MIN]),
                "max_volume_um3": torch.where(none, zero, st[STAT.CRATER_MAX])}

    def get_crater_volumes(self, env_index: int) -> torch.Tensor:
        """`MaterialRemovalModule.crater_volumes_um3` (material.py:133) of one environment since its reset, oldest
        first, from the ring the kernels fill at every fresh spark (needs ``crater_log_capacity``; when more craters
        were sampled than the ring holds, the newest ``capacity`` of them)."""
        log = self.state.crater_log
        if log is None:
            raise RuntimeError("construct the environment with crater_log_capacity > 0 to keep the crater volumes")
        n, cap = int(self.state.spark_count[env_index].item()), log.shape[0]
        if n <= cap:
            return log[:n, env_index].clone()
        idx = torch.arange(n - cap, n, device=log.device) % cap
        return log[idx, env_index]

    def get_crater_count(self) -> torch.Tensor:
        """`len(MaterialRemovalModule.crater_volumes_um3)` (material.py:133), per environment."""
        return self.state.spark_count

    @property
    def workpiece_height(self) -> float:
        return self.config.workpiece_height

    @property
    def wire_diameter(self) -> float:
        return self.config.wire_diameter



def _to_numpy(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
