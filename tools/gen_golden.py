#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

The reference (/root/reference, pure Python) is imported with the two in-memory
stubs of tools/ref_harness.py and driven through the scenarios
below.  Every scenario records, per 1-us step, what `WireEDMEnv.step()` left in
`env.state` and in the module-private fields, plus the exact sequence of random
variates the reference drew (`env.np_random` is wrapped by a recorder).

Two RNG sources:
  native : the reference's own NumPy Generator(PCG64) seeded by reset(seed=...).
           Fixtures carry the draw trace so the oracle can REPLAY it.
  philox : `env.np_random` is replaced by a shim that returns the build's
           Philox4x32-10 variates by slot (computed by oracle/, whose Philox is
           pinned by Random123 known-answer vectors in tests/).  The reference,
           the oracle and the GPU then consume identical variates.

Fixtures are data only (inputs + recorded outputs).  Run:
    python tools/gen_golden.py            # writes tests/golden/*.npz
"""
from __future__ import annotations

import json
import math
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

import ref_harness  # noqa: E402

wedm = ref_harness.import_reference()
from oracle import oracle as orc  # noqa: E402  (only for the Philox shim's variates)

OUT = ROOT / "tests" / "golden"

FLOAT_FIELDS = (
    "workpiece_position", "wire_position", "wire_velocity", "voltage", "current", "spark_y",
    "debris_volume", "debris_density", "flow_rate", "cavity_volume", "last_crater_volume",
    "prev_accel", "tmax", "h_base", "h_zone", "diel_last_gap", "diel_last_density", "wire_last_flow",
)
INT_FIELDS = (
    "time", "time_since_servo", "spark_state", "spark_dur", "is_short_circuit",
    "random_short_remaining", "debris_short_remaining", "time_in_critical_temp",
    "is_wire_broken", "is_target_reached", "terminated", "ctrl_step", "n_draws",
    "time_since_spark_ignition", "time_since_spark_end", "time_since_open_voltage",
)


class RecordingRNG:
    """Wraps the reference's Generator and logs every variate it hands out."""

    def __init__(self, gen):
        self.gen = gen
        self.trace = []
        self.n_step = 0

    def begin_step(self, t):
        self.n_step = 0

    def _log(self, v):
        self.trace.append(float(v))
        self.n_step += 1
        return v

    def random(self):
        return self._log(self.gen.random())

    def uniform(self, low, high):
        return self._log(self.gen.uniform(low, high))

    def normal(self, loc, scale):
        return self._log(self.gen.normal(loc, scale))


class PhiloxShim(RecordingRNG):
    """`env.np_random` stand-in returning the build's Philox variates by slot.

    Per step the reference calls `.random()` at most three times in a fixed order
    (debris roll, random-short roll, ignition roll: ignition.py:233,239,327), so
    the call position identifies the slot."""

    def __init__(self, seed, env_id, episode=0):
        super().__init__(None)
        self.seed, self.env_id, self.episode = seed, env_id, episode
        self.t = 0
        self.n_random = 0

    def begin_step(self, t):
        self.n_step = 0
        self.n_random = 0
        self.t = t

    def random(self):
        slot = self.n_random
        self.n_random += 1
        u = orc.step_uniforms(self.seed, self.env_id, self.episode, self.t)
        return self._log(u[slot])

    def uniform(self, low, high):
        u = orc.step_uniforms(self.seed, self.env_id, self.episode, self.t)
        return self._log(low + (high - low) * u[3])  # NumPy: low + (high-low)*next_double

    def normal(self, loc, scale):
        z, _ = orc.std_normal(self.seed, self.env_id, self.episode, self.t)
        return self._log(loc + scale * z)  # NumPy: loc + scale*standard_normal


def make_action(servo, tv, mode, on, off, dtype=np.float64):
    return {
        "servo": np.array([servo], dtype=dtype),
        "generator_control": {
            "target_voltage": np.array([tv], dtype=dtype),
            "current_mode": np.array([mode], dtype=np.int32),
            "ON_time": np.array([on], dtype=dtype),
            "OFF_time": np.array([off], dtype=dtype),
        },
    }


def action_row(a):
    g = a["generator_control"]
    return [float(a["servo"][0]), float(g["target_voltage"][0]), float(g["ON_time"][0]),
            float(g["OFF_time"][0]), float(int(g["current_mode"][0]))]


def snapshot(env, terminated, ctrl, n_draws):
    s = env.state
    T = s.wire_temperature
    y = s.spark_status[1]
    f = {
        "workpiece_position": s.workpiece_position, "wire_position": s.wire_position,
        "wire_velocity": s.wire_velocity,
        "voltage": 0.0 if s.voltage is None else s.voltage,
        "current": 0.0 if s.current is None else s.current,
        "spark_y": math.nan if y is None else y,
        "debris_volume": env.dielectric.debris_volume, "debris_density": s.debris_density,
        "flow_rate": s.flow_rate, "cavity_volume": s.cavity_volume,
        "last_crater_volume": s.last_crater_volume, "prev_accel": env.mechanics.prev_accel,
        "tmax": float(np.max(T)) if len(T) else math.nan,
        "h_base": float(env.wire.h_eff_zone[0]),
        "h_zone": float(env.wire.h_eff_zone[env.wire.actual_zone_start]),
        "diel_last_gap": env.dielectric._last_gap_um,
        "diel_last_density": env.dielectric._last_debris_density,
        "wire_last_flow": env.wire._last_flow_condition,
    }
    i = {
        "time": s.time, "time_since_servo": s.time_since_servo, "spark_state": int(s.spark_status[0]),
        "spark_dur": int(s.spark_status[2]), "is_short_circuit": int(s.is_short_circuit),
        "random_short_remaining": env.ignition.random_short_remaining,
        "debris_short_remaining": env.ignition.debris_short_remaining,
        "time_in_critical_temp": s.time_in_critical_temp, "is_wire_broken": int(s.is_wire_broken),
        "is_target_reached": int(s.is_target_distance_reached), "terminated": int(terminated),
        "ctrl_step": int(ctrl), "n_draws": n_draws,
        "time_since_spark_ignition": s.time_since_spark_ignition,
        "time_since_spark_end": s.time_since_spark_end,
        "time_since_open_voltage": s.time_since_open_voltage,
    }
    return f, i


def run_scenario(name, *, n_steps, seed=0, config=None, control_mode="position", ignition=None,
                 wire=None, material=None, dielectric=None, mechanics=None, rng="native", env_id=0,
                 state_init=None, module_init=None, T_init=None, action=None, action_schedule=None,
                 controller=None, forced=None, disable_ignition=False, t_snap_every=None,
                 float_stride=1, stop_on_terminate=True, resets=None, note=""):
    """Run one scenario on the reference and write tests/golden/<name>.npz.

    ``resets``: [(step, seed, state_init), ...] -- before step number `step` the SAME environment object is reset with
    the reference's own `env.reset(seed=seed)` (wire_edm.py:106-114: a new EDMState; the module objects live on) and
    `state_init` is applied to the new state, as a driver's second episode does.  ``stop_on_terminate=False`` keeps
    calling `step()` after `terminated`, as nothing in the reference forbids (wire_edm.py:116-157 has no guard)."""
    config = dict(config or {})
    mods = {"ignition": dict(ignition or {}), "wire": dict(wire or {}), "material": dict(material or {}),
            "dielectric": dict(dielectric or {}), "mechanics": dict(mechanics or {})}
    env = ref_harness.quiet(
        wedm.WireEDMEnv,
        mechanics_control_mode=control_mode,
        config=wedm.EnvironmentConfig(**config),
        ignition_params=wedm.IgnitionModuleParameters(**mods["ignition"]),
        wire_params=wedm.WireModuleParameters(**mods["wire"]),
        material_params=wedm.MaterialModuleParameters(**mods["material"]),
        dielectric_params=wedm.DielectricModuleParameters(**mods["dielectric"]),
        mechanics_params=wedm.MechanicsModuleParameters(**mods["mechanics"]),
    )
    env.reset(seed=seed)
    if rng == "native":
        rec = RecordingRNG(env.np_random)
    else:
        rec = PhiloxShim(seed, env_id)
    env.np_random = rec

    for k, v in (state_init or {}).items():
        setattr(env.state, k, v)
    for k, v in (module_init or {}).items():
        mod, attr = k.split(".")
        setattr(getattr(env, mod), attr, v)
    if T_init is not None:
        # the reference allocates T lazily in the first wire.update (wire.py:264-269)
        env.state.wire_temperature = np.full(env.wire.n_segments, env.wire.params.spool_T, dtype=np.float32)
        for lo, hi, val in T_init:
            env.state.wire_temperature[lo:hi] = val
    if disable_ignition:
        env.ignition.update = lambda state, dt=None: None

    actions, action_idx = [], []
    cur_action = action
    floats = {k: [] for k in FLOAT_FIELDS}
    ints = {k: [] for k in INT_FIELDS}
    float_steps = []
    t_snaps, t_snap_steps = [], []
    forced_rows = []
    raised = None

    resets = [(int(at), int(sd), dict(si or {})) for at, sd, si in (resets or [])]
    for step in range(n_steps):
        for at, seed2, init2 in resets:
            if at == step:
                env.reset(seed=seed2)  # the reference's own reset of a USED environment
                if rng == "native":
                    nxt = RecordingRNG(env.np_random)
                else:
                    nxt = PhiloxShim(seed2, env_id)  # reset(seed=) re-keys: key = the new seed, episode 0
                nxt.trace = rec.trace  # one continuous draw trace
                rec = nxt
                env.np_random = rec
                for k, v in init2.items():
                    setattr(env.state, k, v)
        if action_schedule is not None:
            for at, act in action_schedule:
                if at == step:
                    cur_action = act
        if controller is not None and (step == 0 or ints["ctrl_step"][-1]):
            cur_action = controller(env)
        row = action_row(cur_action)
        if not actions or actions[-1] != row:
            actions.append(row)
        action_idx.append(len(actions) - 1)

        if forced is not None:
            st, y, dur, V, I = forced(step)
            env.state.spark_status = [st, None if (isinstance(y, float) and math.isnan(y)) else y, dur]
            env.state.voltage = V
            env.state.current = I
            forced_rows.append([st, y, dur, V, I])

        rec.begin_step(env.state.time)
        try:
            obs, reward, terminated, truncated, info = env.step(cur_action)
        except ValueError as exc:
            raised = (step, str(exc))
            break
        ctrl = info.get("control_step", False)  # early return on wire break has no key
        if hasattr(controller, "observe"):
            controller.observe(env)
        f, i = snapshot(env, terminated, ctrl, rec.n_step)
        for k in INT_FIELDS:
            ints[k].append(i[k])
        if step % float_stride == 0 or step == n_steps - 1 or terminated:
            float_steps.append(step)
            for k in FLOAT_FIELDS:
                floats[k].append(f[k])
        if t_snap_every and ((step + 1) % t_snap_every == 0 or terminated):
            t_snaps.append(env.state.wire_temperature.copy())
            t_snap_steps.append(step)
        if terminated and stop_on_terminate:
            break

    if not t_snaps and len(env.state.wire_temperature):
        t_snaps.append(env.state.wire_temperature.copy())
        t_snap_steps.append(len(ints["time"]) - 1)

    meta = {
        "name": name, "note": note, "seed": seed, "rng": rng, "env_id": env_id,
        "control_mode": control_mode, "config": config, "modules": mods,
        "state_init": state_init or {}, "module_init": module_init or {},
        "T_init": T_init or [], "disable_ignition": disable_ignition, "n_steps_run": len(ints["time"]),
        "resets": [[at, sd, si] for at, sd, si in resets], "stop_on_terminate": bool(stop_on_terminate),
        "n_seg": int(env.wire.n_segments), "zone": [int(env.wire.zone_start), int(env.wire.zone_end)],
        "contacts": [int(env.wire.contact_bottom_idx), int(env.wire.contact_top_idx)],
        "raised": raised, "numpy": np.__version__,
        "wire_material_constants": {k: v for k, v in vars(env.wire.wire_material).items() if k != "name"},
        "float_fields": FLOAT_FIELDS, "int_fields": INT_FIELDS,
    }
    arrays = {
        "meta": np.array(json.dumps(meta)),
        "actions": np.array(actions, dtype=np.float64).reshape(-1, 5),
        "action_idx": np.array(action_idx, dtype=np.int32),
        "draws": np.array(rec.trace, dtype=np.float64),
        "float_steps": np.array(float_steps, dtype=np.int32),
        "floats": np.array([floats[k] for k in FLOAT_FIELDS], dtype=np.float64),
        "ints": np.array([ints[k] for k in INT_FIELDS], dtype=np.int32),
        "T_snaps": np.array(t_snaps, dtype=np.float32),
        "T_snap_steps": np.array(t_snap_steps, dtype=np.int32),
    }
    # MaterialRemovalModule.get_crater_statistics() (material.py:207-227) at the end of the run
    cs = env.material.get_crater_statistics()
    arrays["crater_volumes_um3"] = np.asarray(cs["volumes_um3"], dtype=np.float64)
    arrays["crater_stats"] = np.array([cs["total_craters"], cs["mean_volume_um3"], cs["std_volume_um3"],
                                       cs["min_volume_um3"], cs["max_volume_um3"]], dtype=np.float64)
    if forced_rows:
        arrays["forced"] = np.array(forced_rows, dtype=np.float64)
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / f"{name}.npz", **arrays)
    hist = {int(v): int(c) for v, c in zip(*np.unique(arrays["ints"][INT_FIELDS.index("spark_state")], return_counts=True))}
    print(f"{name:28s} steps={meta['n_steps_run']:6d} draws={len(rec.trace):6d} n_seg={meta['n_seg']} "
          f"states={hist} raised={raised is not None} "
          f"size={(OUT / (name + '.npz')).stat().st_size // 1024} KiB")
    return env


# ------------------------------------------------------------------ scenarios
QUICKSTART = make_action(0.1, 80.0, 5, 3.0, 80.0)  # examples/quickstart.py:25-33


def gap_controller(desired_gap=5.0):
    """experiments/run_simulation.py:24-56 (position and velocity branches)."""

    def controller(env):
        gap = env.state.workpiece_position - env.state.wire_position
        error = gap - desired_gap
        if env.mechanics.control_mode == "position":
            delta = error * 0.1
        else:
            delta = float(np.clip(error * 50.0, -1000.0, 1000.0))
        return make_action(delta, 80.0, 7, 2.0, 33.0, dtype=np.float32)

    return controller


def reference_driver_module():
    """The reference's own experiments/run_simulation.py (its controllers are used as they are)."""
    import importlib.util

    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    spec = importlib.util.spec_from_file_location("ref_run_simulation", "/root/reference/experiments/run_simulation.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class VoltageDriver:
    """The reference's PI voltage controller (`create_voltage_controller`, run_simulation.py:58-109,
    imported, not restated) fed the way `run_simulation()` feeds it (run_simulation.py:258-281):
    the voltage of every microsecond is appended to a history that keeps the samples whose time
    is >= state.time - 1000, and the controller receives a copy at each control step."""

    def __init__(self, target_voltage=30.0):
        self.ctl = reference_driver_module().create_voltage_controller(target_voltage)
        self.vh, self.th = [], []

    def observe(self, env):
        self.vh.append(env.state.voltage if env.state.voltage is not None else 0.0)
        self.th.append(env.state.time)
        cutoff = env.state.time - 1000.0
        while self.th and self.th[0] < cutoff:
            self.vh.pop(0)
            self.th.pop(0)

    def __call__(self, env):
        return self.ctl(env, list(self.vh))


def run_logger_scenario(name, *, n_steps, seed, env_id, control_mode="position", wire=None, state_init=None,
                        interval=50):
    """The reference's OWN `SimulationLogger` (utils/logger.py:54-237) fed by the reference's own driver
    loop (experiments/run_simulation.py:241-297: gap controller recomputed on control steps,
    `logger.collect(env.state, info)` after every 1-us step) over the signal list of
    `setup_logger(log_strategy="both")` (run_simulation.py:127-147), at the three log frequencies
    (every_step / interval / control_step), Philox variates injected.  The fixture holds what
    `np.array(logger.get_data()[signal])` gives, key by key: `<frequency>/<signal>`."""
    from wedm.utils.logger import SimulationLogger as RefLogger

    drv = reference_driver_module()
    cfg_file = drv.setup_logger(control_mode, log_to_file=True, log_strategy="both")  # the reference's signal list
    signals = list(cfg_file["signals_to_log"])
    assert "dielectric_flow_rate" in signals and "wire_average_temperature" in signals and "wire_temperature" in signals
    wire = dict(wire or {})
    env = ref_harness.quiet(wedm.WireEDMEnv, mechanics_control_mode=control_mode,
                            wire_params=wedm.WireModuleParameters(**wire))
    env.reset(seed=seed)
    rec = PhiloxShim(seed, env_id)
    env.np_random = rec
    for k, v in (state_init or {}).items():
        setattr(env.state, k, v)
    no_field = [s for s in signals if s != "wire_temperature"]  # the every-microsecond log without the full field (size)
    loggers = {
        "every_step": RefLogger({"signals_to_log": no_field, "log_frequency": {"type": "every_step"},
                                 "backend": {"type": "memory"}}, env_reference=env),
        "interval": RefLogger({"signals_to_log": signals, "log_frequency": {"type": "interval", "value": interval},
                               "backend": {"type": "memory"}}, env_reference=env),
        "control_step": RefLogger({"signals_to_log": signals, "log_frequency": {"type": "control_step"},
                                   "backend": {"type": "memory"}}, env_reference=env),
    }
    for lg in loggers.values():
        lg.reset()
    controller = drv.create_gap_controller()
    action = controller(env)
    for _ in range(n_steps):  # run_simulation.py:255-281
        rec.begin_step(env.state.time)
        obs, reward, terminated, truncated, info = env.step(action)
        for lg in loggers.values():
            lg.collect(env.state, info)
        if info.get("control_step", False):
            action = controller(env)
        assert not (terminated or truncated)
    arrays = {}
    for freq, lg in loggers.items():
        lg.finalize()
        for sig, values in lg.get_data().items():
            arr = np.array(values)
            assert arr.dtype != object, (freq, sig)
            arrays[f"{freq}/{sig}"] = arr
    meta = {"name": name, "seed": seed, "env_id": env_id, "rng": "philox", "control_mode": control_mode,
            "modules": {"wire": wire}, "state_init": state_init or {}, "n_steps_run": n_steps, "interval": interval,
            "signals": signals, "every_step_signals": no_field, "n_seg": int(env.wire.n_segments),
            "base_flow_rate": env.dielectric.params.base_flow_rate, "numpy": np.__version__}
    arrays["meta"] = np.array(json.dumps(meta))
    OUT.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(OUT / f"{name}.npz", **arrays)
    print(f"{name:28s} steps={n_steps} keys={len(arrays) - 1} "
          f"size={(OUT / (name + '.npz')).stat().st_size // 1024} KiB")


def env_probe_modes():
    """Current modes that have crater data in the reference (material.py:108-113)."""
    env = ref_harness.quiet(wedm.WireEDMEnv)
    return [m for m in env.material.crater_data.keys()]


def single_spark_forced(spark_time=50, duration=2, loc=25.0, ocv=80.0):
    """experiments/single_spark_animation.py:209-251."""

    def forced(step):
        t = step + 1
        if spark_time <= t < spark_time + duration:
            return 1, loc, 1, ocv * 0.3, 60.0
        return 0, math.nan, 0, ocv, 0.0

    return forced


def main():
    import os

    only = os.environ.get("WEDM_GOLDEN_ONLY")
    if only:
        global run_scenario
        _orig = run_scenario

        def run_scenario(name, **kw):  # noqa: F811
            if only in name:
                return _orig(name, **kw)
            return None
    # F1 — BASELINE config 1: brass 0.25 mm wire, seed 0, quickstart action, 10 000 us
    run_scenario("f1_config1_native", n_steps=10000, seed=0, config={"wire_diameter": 0.25},
                 action=QUICKSTART, t_snap_every=1000, float_stride=7,
                 note="BASELINE.json configs[0]; native PCG64 stream, float64 action arrays")
    run_scenario("f1_config1_f32action", n_steps=4000, seed=0, config={"wire_diameter": 0.25},
                 action=make_action(0.1, 80.0, 5, 3.0, 80.0, dtype=np.float32), t_snap_every=2000,
                 float_stride=7, note="same with float32 action leaves (action_space dtype)")

    # F2 — single-spark known-answer test, no RNG
    run_scenario("f2_single_spark", n_steps=1000, seed=42,
                 state_init={"workpiece_position": 50.0, "wire_position": 40.0, "target_position": 5000.0,
                             "wire_unwinding_velocity": 0.0, "dielectric_temperature": 293.15},
                 T_init=[(0, None, 293.15)], disable_ignition=True, forced=single_spark_forced(),
                 action=make_action(0.0, 80.0, 13, 2.0, 1000.0, dtype=np.float32),
                 t_snap_every=50, note="experiments/single_spark_animation.py scenario")

    # F3 — Philox-injected runs (reference consumes the build's variates), several env ids
    for env_id in (0, 1, 2, 3, 777, 65535):
        run_scenario(f"f3_philox_env{env_id}", n_steps=6000, seed=1234, rng="philox", env_id=env_id,
                     state_init={"workpiece_position": 25.0, "wire_position": 10.0, "target_position": 5000.0},
                     action=QUICKSTART, t_snap_every=2000, float_stride=5,
                     note="gap 15 um start: frequent sparks; Philox key 1234")
    run_scenario("f3_philox_config3_env5", n_steps=4000, seed=99, rng="philox", env_id=5,
                 wire={"segment_len": 0.625},
                 state_init={"workpiece_position": 22.0, "wire_position": 10.0, "target_position": 5000.0},
                 action=make_action(0.05, 100.0, 9, 2.0, 20.0), t_snap_every=1000, float_stride=5,
                 note="BASELINE config 3 grid: segment_len 0.625 -> 128 segments")

    # F5 — edge paths
    run_scenario("f5_hard_short", n_steps=400, seed=3,
                 state_init={"workpiece_position": 11.0, "wire_position": 10.0, "target_position": 5000.0},
                 action=make_action(-0.5, 80.0, 5, 3.0, 80.0), note="gap < 2 um: 51-step shorts, state -1 pulses")
    run_scenario("f5_debris_short", n_steps=3000, seed=4,
                 state_init={"workpiece_position": 20.0, "wire_position": 10.0, "target_position": 5000.0},
                 module_init={"dielectric.debris_volume": 0.0316},
                 action=make_action(0.0, 80.0, 5, 3.0, 80.0), float_stride=3,
                 note="debris density ~0.5 at gap 10: sigmoid short model, np.exp branch of fast_exp")
    run_scenario("f5_random_short", n_steps=3000, seed=5, ignition={"random_short_max_probability": 0.004},
                 state_init={"workpiece_position": 30.0, "wire_position": 10.0, "target_position": 5000.0},
                 action=QUICKSTART, float_stride=3, note="random gap-dependent shorts enabled")
    run_scenario("f5_collision", n_steps=200, seed=6,
                 state_init={"workpiece_position": 50.0, "wire_position": 149.5, "wire_velocity": 20000.0,
                             "target_position": 5000.0},
                 action=make_action(1.0, 80.0, 5, 3.0, 80.0), note="wire > workpiece + 100 -> broken")
    run_scenario("f5_target_reached", n_steps=4000, seed=7,
                 state_init={"workpiece_position": 25.0, "wire_position": 10.0, "target_position": 25.002},
                 action=QUICKSTART, note="workpiece_position >= target_position after a few craters")
    run_scenario("f5_critical_temp", n_steps=300, seed=8, T_init=[(180, 181, 1502.0), (300, 304, 1100.0)],
                 action=QUICKSTART, t_snap_every=100, note="hot cells below break: time_in_critical_temp counter")
    run_scenario("f5_wire_break", n_steps=50, seed=9, T_init=[(180, 186, 1600.0)],
                 action=QUICKSTART, t_snap_every=1, note="Tmax > 1500 K: early return before mechanics/clocks")
    sched = [(0, make_action(0.1, 80.0, 5, 3.0, 80.0)), (250, make_action(0.3, 60.0, 7, 2.0, 40.0)),
             (1000, make_action(-0.2, 120.0, 9, 4.0, 20.0)), (1001, make_action(0.5, 90.0, 3, 1.0, 10.0)),
             (1500, make_action(0.0, 70.0, 11, 2.5, 33.0)), (2001, make_action(0.25, 100.0, 13, 2.0, 50.0))]
    run_scenario("f5_action_latch", n_steps=3200, seed=10, action_schedule=sched,
                 state_init={"workpiece_position": 25.0, "wire_position": 10.0, "target_position": 5000.0},
                 float_stride=3, note="action latched only at calls 1001, 2001, 3001; non-integer ON time")
    run_scenario("f5_zero_fallbacks", n_steps=2600, seed=11, action=make_action(0.1, 0.0, 5, 0.0, 0.0),
                 state_init={"workpiece_position": 25.0, "wire_position": 10.0, "target_position": 5000.0},
                 float_stride=3, note="0.0 target_voltage/ON/OFF fall back to defaults (`x or default`)")
    run_scenario("f5_invalid_mode", n_steps=3000, seed=12, action=make_action(0.1, 80.0, 2, 3.0, 80.0),
                 state_init={"workpiece_position": 20.0, "wire_position": 10.0, "target_position": 5000.0},
                 float_stride=50, note="even mode: ValueError at the first fresh spark after the latch")

    # F6 — velocity-mode mechanics
    run_scenario("f6_velocity_mode", n_steps=3000, seed=13, control_mode="velocity",
                 action=make_action(200.0, 80.0, 5, 3.0, 80.0), float_stride=3,
                 note="first-order velocity servo")
    run_scenario("f6_limits", n_steps=1500, seed=14, action=make_action(1.0, 80.0, 5, 3.0, 80.0),
                 mechanics={"max_acceleration": 2.0e3, "max_jerk": 5.0e6, "max_speed": 1.5},
                 float_stride=1, note="acceleration, jerk and speed clips all active")

    # F7 — run_simulation.py driver: gap P-controller on the control step
    run_scenario("f7_gap_controller", n_steps=12000, seed=0, controller=gap_controller(),
                 state_init={"workpiece_position": 70.0, "wire_position": 10.0, "target_position": 5000.0,
                             "dielectric_temperature": 293.15},
                 t_snap_every=4000, float_stride=9,
                 note="experiments/run_simulation.py:173-297 with the gap controller (float32 leaves)")

    # F7p — the same driver with the build's Philox variates injected (two environments), so the
    # batched environment + on-device controller can be compared with the reference directly
    for env_id in (0, 9):
        run_scenario(f"f7_gap_controller_philox_env{env_id}", n_steps=9000, seed=77, rng="philox", env_id=env_id,
                     controller=gap_controller(),
                     state_init={"workpiece_position": 70.0, "wire_position": 10.0, "target_position": 5000.0},
                     t_snap_every=3000, float_stride=9, note="run_simulation.py gap controller, Philox key 77")
    run_scenario("f7_gap_controller_velocity_philox_env3", n_steps=6000, seed=78, rng="philox", env_id=3,
                 control_mode="velocity", controller=gap_controller(),
                 state_init={"workpiece_position": 70.0, "wire_position": 10.0, "target_position": 5000.0},
                 t_snap_every=3000, float_stride=9, note="velocity-mode gap controller (clip +-1000 um/s)")

    # F9 — the driver's PI voltage controller (1 ms moving-average voltage), Philox variates
    run_scenario("f9_voltage_controller_philox_env2", n_steps=8000, seed=79, rng="philox", env_id=2,
                 controller=VoltageDriver(30.0),
                 state_init={"workpiece_position": 70.0, "wire_position": 10.0, "target_position": 5000.0},
                 t_snap_every=4000, float_stride=9, note="run_simulation.py voltage controller, position mode")
    run_scenario("f9_voltage_controller_velocity_philox_env5", n_steps=6000, seed=80, rng="philox", env_id=5,
                 control_mode="velocity", controller=VoltageDriver(30.0),
                 state_init={"workpiece_position": 70.0, "wire_position": 10.0, "target_position": 5000.0},
                 t_snap_every=3000, float_stride=9, note="run_simulation.py voltage controller, velocity mode")

    # F9b — the same controller with a 2-us physics step (501-sample window) and with a 500-us servo interval
    # (the 1 ms window then spans two control intervals)
    run_scenario("f9_voltage_controller_dt2_philox_env4", n_steps=3000, seed=85, rng="philox", env_id=4,
                 config={"dt": 2}, controller=VoltageDriver(30.0),
                 state_init={"workpiece_position": 40.0, "wire_position": 10.0, "target_position": 5000.0},
                 t_snap_every=3000, float_stride=9, note="voltage controller, config.dt = 2 us")
    run_scenario("f9_voltage_controller_servo500_philox_env6", n_steps=5000, seed=86, rng="philox", env_id=6,
                 config={"servo_interval": 500}, controller=VoltageDriver(30.0),
                 state_init={"workpiece_position": 40.0, "wire_position": 10.0, "target_position": 5000.0},
                 t_snap_every=5000, float_stride=9, note="voltage controller, servo_interval = 500 us")

    # F10 — many craters: get_crater_statistics() (material.py:207-227) over a busy run
    run_scenario("f10_crater_statistics_philox_env1", n_steps=12000, seed=81, rng="philox", env_id=1,
                 state_init={"workpiece_position": 22.0, "wire_position": 10.0, "target_position": 5000.0},
                 action=make_action(0.05, 80.0, 13, 2.0, 20.0), t_snap_every=12000, float_stride=97,
                 note="dense sparking, mode I13: crater statistics")

    # F11 — randomized parameter sets over every module (seeded), Philox variates: widens the part of
    # the parameter space on which the oracle (and through it the GPU) is pinned to the reference
    prng = np.random.default_rng(20261004)
    valid_modes = sorted(int(k[1:]) for k in env_probe_modes())
    for k in range(8):
        u = prng.uniform
        cfg = {"workpiece_height": float(u(8.0, 32.0)), "wire_diameter": float(prng.choice([0.1, 0.15, 0.2, 0.25, 0.3])),
               "servo_interval": int(prng.choice([250, 500, 1000])), "initial_gap": float(u(15.0, 60.0))}
        ign = {"base_critical_density": float(u(0.05, 0.4)), "gap_coefficient": float(u(0.005, 0.03)),
               "max_critical_density": float(u(0.6, 0.99)), "hard_short_gap": float(u(1.0, 4.0)),
               "sigmoid_steepness": float(prng.choice([50.0, 200.0, 500.0])),
               "debris_short_duration": int(prng.integers(10, 80)), "random_short_duration": int(prng.integers(20, 120)),
               "random_short_min_gap": float(u(1.0, 5.0)), "random_short_max_gap": float(u(30.0, 70.0)),
               "random_short_max_probability": float(prng.choice([0.0, 0.002, 0.01])),
               "ignition_a_coeff": float(0.48 * u(0.8, 1.2)), "ignition_b_coeff": float(-3.69 * u(0.8, 1.2)),
               "ignition_c_coeff": float(14.05 * u(0.9, 1.3)), "default_target_voltage": float(u(60.0, 100.0)),
               "default_on_time": float(u(1.0, 4.0)), "default_off_time": float(u(20.0, 90.0)),
               "spark_voltage_factor": float(u(0.2, 0.5))}
        wire = {"segment_len": float(prng.choice([0.2, 0.25, 0.4, 0.5])), "buffer_len_bottom": float(u(10.0, 40.0)),
                "buffer_len_top": float(u(10.0, 40.0)), "contact_offset_bottom": float(u(5.0, 15.0)),
                "contact_offset_top": float(u(5.0, 15.0)), "base_convection_coefficient": float(u(8000.0, 20000.0)),
                "plasma_efficiency": float(u(0.05, 0.3)), "convection_velocity_factor": float(u(0.2, 0.8)),
                "convection_flow_enhancement": float(u(0.5, 1.5)), "spool_T": float(prng.choice([288.15, 293.15, 300.0])),
                "critical_temp_threshold": float(u(0.7, 0.95))}
        mat = {"base_overcut": float(u(0.08, 0.2))}
        diel = {"base_flow_rate": float(u(50.0, 200.0)), "debris_removal_efficiency": float(u(0.005, 0.05)),
                "debris_obstruction_coeff": float(u(0.5, 3.0)), "reference_gap": float(u(15.0, 40.0)),
                "dielectric_temperature": float(u(285.0, 300.0))}
        mech = {"omega_n": float(u(150.0, 400.0)), "zeta": float(u(0.2, 0.9)), "max_acceleration": float(3.0e5 * u(0.3, 1.5)),
                "max_jerk": float(1.0e8 * u(0.3, 1.5)), "max_speed": float(3.0e4 * u(0.3, 1.5))}
        mode = "velocity" if k % 3 == 2 else "position"
        servo = float(u(50.0, 400.0)) if mode == "velocity" else float(u(-0.05, 0.3))
        act = make_action(servo, float(u(60.0, 120.0)), int(prng.choice(valid_modes)), float(u(1.0, 4.0)), float(u(10.0, 60.0)))
        run_scenario(f"f11_random_params_{k}", n_steps=3000, seed=1000 + k, rng="philox", env_id=int(prng.integers(0, 64)),
                     control_mode=mode, config=cfg, ignition=ign, wire=wire, material=mat, dielectric=diel, mechanics=mech,
                     state_init={"workpiece_position": float(10.0 + u(6.0, 30.0)), "wire_position": 10.0,
                                 "target_position": 5000.0},
                     module_init={"dielectric.debris_volume": float(u(0.0, 0.02))} if k % 2 else None,
                     action=act, t_snap_every=1500, float_stride=7, note="randomized parameters over all modules")

    # geometry variants (BASELINE config 5 shapes), short Philox runs
    for i, (h, d) in enumerate(((10.0, 0.10), (15.0, 0.25), (30.0, 0.30), (12.3, 0.15))):
        run_scenario(f"f8_geometry_{i}", n_steps=2500, seed=2024, rng="philox", env_id=100 + i,
                     config={"workpiece_height": h, "wire_diameter": d},
                     state_init={"workpiece_position": 24.0, "wire_position": 10.0, "target_position": 5000.0},
                     action=make_action(0.1, 80.0, (1, 7, 15, 17)[i], 3.0, 40.0), t_snap_every=2500,
                     float_stride=5, note=f"h={h} d={d}")

    # F13 — a physics step other than 1 us (config.dt = 2: clocks and mechanics scale, wire.py keeps 1e-6 s)
    run_scenario("f13_dt2_philox_env4", n_steps=2500, seed=90, rng="philox", env_id=4,
                 config={"dt": 2, "servo_interval": 500},
                 state_init={"workpiece_position": 24.0, "wire_position": 10.0, "target_position": 5000.0},
                 action=make_action(0.1, 80.0, 9, 3.0, 30.0), t_snap_every=2500, float_stride=7,
                 note="config.dt = 2 us")

    # F14 — a wire material other than the built-in brass, registered in the material database
    copper = dict(density=8960, specific_heat=385, thermal_conductivity=401, electrical_resistivity=1.68e-8,
                  temperature_coefficient=0.00393, melting_point=1358, breaking_temperature=1600)
    if not only or only in "f14_copper_wire_philox_env6":
        from wedm.core.material_db import WireMaterial as RefWireMaterial, get_material_db as ref_db

        ref_db()._wire_materials["copper"] = RefWireMaterial(name="copper", **copper)
    run_scenario("f14_copper_wire_philox_env6", n_steps=2500, seed=91, rng="philox", env_id=6,
                 config={"wire_material": "copper", "wire_diameter": 0.25},
                 state_init={"workpiece_position": 22.0, "wire_position": 10.0, "target_position": 5000.0},
                 action=make_action(0.1, 80.0, 15, 3.0, 25.0), t_snap_every=2500, float_stride=7,
                 note="custom wire material through the material database")

    # F15 — before the first latch state.current_mode is None: the ignition module's fresh cache answers
    # 60 A (ignition.py:79-81,98-113; default_current_mode is NOT consulted), material uses "I1"
    # (material.py:104-105)
    run_scenario("f15_default_mode_philox_env7", n_steps=1800, seed=92, rng="philox", env_id=7,
                 ignition={"default_current_mode": "I13"},
                 state_init={"workpiece_position": 20.0, "wire_position": 10.0, "target_position": 5000.0},
                 action=make_action(0.1, 90.0, 9, 2.0, 25.0), t_snap_every=1800, float_stride=3,
                 note="sparks before the first control-step latch use the default modes")

    # F17 — the reference's SECOND episode: reset() of a used environment re-initialises EDMState only
    # (wire_edm.py:106-114); module-private state leaks into the next episode
    dense = {"workpiece_position": 25.0, "wire_position": 10.0, "target_position": 5000.0}
    run_scenario("f17_second_episode_philox_env2", n_steps=6000, seed=95, rng="philox", env_id=2,
                 state_init=dense, resets=[(3000, 951, dense)], action=make_action(0.1, 80.0, 9, 3.0, 30.0),
                 t_snap_every=1000, float_stride=3,
                 note="reset(seed) -> 3000 us -> reset(seed') -> 3000 us on ONE environment object: debris volume, flow / "
                      "density / convection caches, prev_accel and the crater list carry over")
    run_scenario("f17_second_episode_native", n_steps=5000, seed=96, state_init=dense, resets=[(2500, 961, dense)],
                 action=make_action(0.1, 80.0, 9, 3.0, 30.0), t_snap_every=1250, float_stride=3,
                 note="the same with the reference's own PCG64 streams (reset(seed') reseeds env.np_random)")
    run_scenario("f17_reset_during_short_philox_env4", n_steps=400, seed=97, rng="philox", env_id=4,
                 state_init={"workpiece_position": 11.5, "wire_position": 10.0, "target_position": 5000.0},
                 resets=[(130, 971, {"target_position": 5000.0})], action=make_action(0.0, 80.0, 5, 3.0, 80.0),
                 note="gap 1.5 um < hard_short_gap: a 50-us debris-short timer is running when reset() comes; it runs on "
                      "in the new episode at its 50-um gap (ignition.py:75-76,204-217)")
    run_scenario("f17_stale_current_cache_philox_env5", n_steps=4200, seed=98, rng="philox", env_id=5,
                 ignition={"default_current_mode": "I13"}, state_init=dense, resets=[(2500, 981, dense)],
                 action=make_action(0.1, 90.0, 9, 2.0, 25.0), t_snap_every=2100, float_stride=3,
                 note="after the reset state.current_mode is None again but the ignition module's current cache names the "
                      "previous episode's mode: None resolves through default_current_mode (I13), not the fresh 60 A "
                      "(ignition.py:98-113)")

    # F18 — step() called after `terminated` (wire_edm.py:116-157 has no guard)
    run_scenario("f18_past_target_philox_env1", n_steps=2400, seed=99, rng="philox", env_id=1,
                 state_init={"workpiece_position": 25.0, "wire_position": 10.0, "target_position": 25.002},
                 action=QUICKSTART, stop_on_terminate=False, t_snap_every=600, float_stride=3,
                 note="target reached, then stepped on: everything continues, terminated stays True")
    run_scenario("f18_past_wire_break_philox_env3", n_steps=600, seed=100, rng="philox", env_id=3,
                 state_init=dense, T_init=[(180, 186, 1600.0)], action=make_action(0.1, 80.0, 9, 3.0, 30.0),
                 stop_on_terminate=False, t_snap_every=100,
                 note="Tmax > 1500 K at the first step, then 599 more: ignition / material / dielectric go on, the wire module "
                      "returns at once, every step returns before mechanics and clocks (time stays 0, so the Philox shim "
                      "hands out the same variates each step)")
    run_scenario("f18_past_wire_break_native", n_steps=600, seed=101, state_init=dense, T_init=[(180, 186, 1600.0)],
                 action=make_action(0.1, 80.0, 9, 3.0, 30.0), stop_on_terminate=False, t_snap_every=100,
                 note="the same with the reference's own PCG64 stream (which does advance)")
    run_scenario("f18_past_collision_philox_env6", n_steps=700, seed=102, rng="philox", env_id=6,
                 state_init={"workpiece_position": 50.0, "wire_position": 149.5, "wire_velocity": 20000.0,
                             "target_position": 5000.0},
                 action=make_action(1.0, 80.0, 5, 3.0, 80.0), stop_on_terminate=False,
                 note="wire > workpiece + 100 -> is_wire_broken by _check_termination, then stepped on")

    # F16 — the reference's own SimulationLogger over its own driver loop and signal list
    if not only or only in "f16_logger_philox_env3":
        run_logger_scenario("f16_logger_philox_env3", n_steps=3300, seed=83, env_id=3,
                            wire={"segment_len": 0.625, "compute_zone_mean": True},
                            state_init={"workpiece_position": 24.0, "wire_position": 10.0, "target_position": 5000.0})
    if not only or only in "f16_logger_velocity_philox_env1":
        run_logger_scenario("f16_logger_velocity_philox_env1", n_steps=2200, seed=84, env_id=1, control_mode="velocity",
                            wire={"segment_len": 0.625, "compute_zone_mean": True, "zone_mean_interval": 70},
                            state_init={"workpiece_position": 30.0, "wire_position": 10.0, "target_position": 5000.0},
                            interval=35)

    # F12 — the modules' public getters over a grid (ignition.py:348-384, material.py:176-205)
    if not only or only in "f12_module_getters":
        env = ref_harness.quiet(wedm.WireEDMEnv)
        env.reset(seed=0)
        gaps = np.concatenate([np.linspace(0.25, 4.0, 16), np.linspace(5.0, 80.0, 31)])
        dens = np.linspace(0.0, 1.0, 41)
        crit = np.array([env.ignition.get_critical_density_for_gap(float(g)) for g in gaps])
        pshort = np.array([[env.ignition.get_debris_short_probability(float(g), float(d)) for d in dens] for g in gaps])
        lam = []
        for g in gaps:
            env.state.workpiece_position, env.state.wire_position, env.state.is_short_circuit = 10.0 + float(g), 10.0, False
            lam.append(env.ignition.get_lambda(env.state))
        table = env.material.get_current_mapping_table()
        rows = []
        for i in range(1, 20):
            e = table[f"I{i}"]
            cd = e["crater_data"]
            rows.append([i, e["machine_current"], 1.0 if cd else 0.0,
                         cd["ellipsoid_volume_half"] if cd else 0.0, cd["ellipsoid_volume_std"] if cd else 0.0,
                         cd["depth"] if cd else 0.0])
        np.savez_compressed(OUT / "f12_module_getters.npz", gaps=gaps, densities=dens, critical_density=crit,
                            debris_short_probability=pshort, ignition_lambda=np.array(lam, dtype=np.float64),
                            mapping=np.array(rows, dtype=np.float64))
        print("f12_module_getters", len(gaps), "gaps x", len(dens), "densities")

    # F4 — geometry table straight from WireModule.__init__
    rows = []
    for h in (5.0, 10.0, 12.3, 15.0, 20.0, 25.0, 30.0, 47.7):
        for seg in (0.1, 0.2, 0.25, 0.3, 0.5, 0.625, 1.0):
            for d in (0.1, 0.2, 0.25, 0.3):
                for bb, bt, cb, ct in ((30.0, 30.0, 10.0, 10.0), (20.0, 35.0, 5.0, 12.5), (8.0, 8.0, 10.0, 10.0)):
                    env = ref_harness.quiet(
                        wedm.WireEDMEnv, config=wedm.EnvironmentConfig(workpiece_height=h, wire_diameter=d),
                        wire_params=wedm.WireModuleParameters(segment_len=seg, buffer_len_bottom=bb,
                                                              buffer_len_top=bt, contact_offset_bottom=cb,
                                                              contact_offset_top=ct))
                    w = env.wire
                    rows.append([h, seg, d, bb, bt, cb, ct, w.n_segments, w.zone_start, w.zone_end,
                                 w.actual_zone_start, w.actual_zone_end, w.contact_bottom_idx, w.contact_top_idx,
                                 w.k_cond_coeff, w.temp_update_factor, w.A, w.S, w.joule_geom_factor,
                                 w.critical_temperature, w.breaking_temperature,
                                 env.dielectric.cavity_volume_coeff, env.dielectric.debris_removal_per_us,
                                 env.mechanics.damping_coeff, env.mechanics.stiffness_coeff,
                                 env.mechanics.max_jerk_dt, env.mechanics.dt])
    cols = ["h", "seg", "d", "buf_bottom", "buf_top", "off_bottom", "off_top", "n_seg", "zone_start", "zone_end",
            "az_start", "az_end", "contact_bottom", "contact_top", "k_cond", "tuf", "a_surf", "s_area",
            "joule_geom", "critical_temperature", "breaking_temperature", "cavity_coeff",
            "debris_removal_per_us", "damping_coeff", "stiffness_coeff", "max_jerk_dt", "dt_s"]
    np.savez_compressed(OUT / "f4_geometry_table.npz", table=np.array(rows, dtype=np.float64),
                        columns=np.array(json.dumps(cols)))
    print(f"f4_geometry_table rows={len(rows)}")


if __name__ == "__main__":
    main()
