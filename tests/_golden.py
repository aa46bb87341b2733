"""Replays a tests/golden/*.npz scenario on the CPU oracle and compares step by step.

Test infrastructure only.  The fixtures were produced by the reference itself
(tools/gen_golden.py); nothing here reads /root/reference.
"""
from __future__ import annotations

import ctypes as C
import json
import math

import numpy as np

from oracle import oracle as orc

# reference attribute -> oracle Env field
STATE_ATTR = {
    "workpiece_position": "workpiece_position", "wire_position": "wire_position",
    "wire_velocity": "wire_velocity", "target_position": "target_position",
    "wire_unwinding_velocity": "wire_unwinding_velocity",
    "dielectric_temperature": "dielectric_temperature",
}
MODULE_ATTR = {"dielectric.debris_volume": "debris_volume"}

FLOAT_GETTERS = {
    "workpiece_position": lambda e: e.workpiece_position, "wire_position": lambda e: e.wire_position,
    "wire_velocity": lambda e: e.wire_velocity, "voltage": lambda e: e.voltage, "current": lambda e: e.current,
    "spark_y": lambda e: e.spark_y, "debris_volume": lambda e: e.debris_volume,
    "debris_density": lambda e: e.debris_density, "flow_rate": lambda e: e.flow_rate,
    "cavity_volume": lambda e: e.cavity_volume, "last_crater_volume": lambda e: e.last_crater_volume,
    "prev_accel": lambda e: e.prev_accel, "tmax": lambda e: float(e.tmax), "h_base": lambda e: float(e.h_base),
    "h_zone": lambda e: float(e.h_zone), "diel_last_gap": lambda e: e.diel_last_gap,
    "diel_last_density": lambda e: e.diel_last_density, "wire_last_flow": lambda e: e.wire_last_flow,
}
INT_GETTERS = {
    "time": lambda e: e.time, "time_since_servo": lambda e: e.time_since_servo,
    "spark_state": lambda e: e.spark_state, "spark_dur": lambda e: e.spark_dur,
    "is_short_circuit": lambda e: e.is_short_circuit,
    "random_short_remaining": lambda e: e.random_short_remaining,
    "debris_short_remaining": lambda e: e.debris_short_remaining,
    "time_in_critical_temp": lambda e: e.time_in_critical_temp,
    "is_wire_broken": lambda e: e.is_wire_broken, "is_target_reached": lambda e: e.is_target_reached,
    "terminated": lambda e: e.last_terminated, "ctrl_step": lambda e: e.last_ctrl_step,
    "n_draws": lambda e: e.rng.draws_this_step,
    "time_since_spark_ignition": lambda e: e.time_since_spark_ignition,
    "time_since_spark_end": lambda e: e.time_since_spark_end,
    "time_since_open_voltage": lambda e: e.time_since_open_voltage,
}


class Fixture:
    def __init__(self, path):
        z = np.load(path, allow_pickle=False)
        self.meta = json.loads(str(z["meta"]))
        self.actions = z["actions"]
        self.action_idx = z["action_idx"]
        self.draws = np.ascontiguousarray(z["draws"])
        self.float_steps = z["float_steps"]
        self.floats = z["floats"]
        self.ints = z["ints"]
        self.T_snaps = z["T_snaps"]
        self.T_snap_steps = z["T_snap_steps"]
        self.forced = z["forced"] if "forced" in z.files else None
        self.data = {k: z[k] for k in ("crater_stats", "crater_volumes_um3") if k in z.files}
        self.float_fields = list(self.meta["float_fields"])
        self.int_fields = list(self.meta["int_fields"])
        self.n_steps = int(self.meta["n_steps_run"])

    def int_row(self, name):
        return self.ints[self.int_fields.index(name)]

    def float_row(self, name):
        return self.floats[self.float_fields.index(name)]


def config_from_meta(meta) -> orc.Config:
    cfg = orc.default_config()
    for k, v in meta["config"].items():
        if k == "wire_material":
            if v != "brass":  # a material registered in the reference's database: its constants travel in the meta
                for ck, cv in meta["wire_material_constants"].items():
                    setattr(cfg, ck, cv)
            continue
        setattr(cfg, k, v)
    for mod, params in meta["modules"].items():
        for k, v in params.items():
            if k == "default_current_mode":
                v = int(str(v)[1:])
            setattr(cfg, k, v)
    cfg.control_mode = 0 if meta["control_mode"] == "position" else 1
    return cfg


def oracle_env_for(fx: Fixture, *, math_mode=orc.MATH_LIBM, stencil_mode=orc.STENCIL_F32) -> orc.Env:
    meta = fx.meta
    env = orc.new_env(config_from_meta(meta))
    env.math_mode = math_mode
    env.stencil_mode = stencil_mode
    if meta["rng"] == "native":
        env.rng.mode = orc.RNG_REPLAY
        env.rng.replay = fx.draws.ctypes.data_as(C.POINTER(C.c_double))
        env.rng.replay_len = len(fx.draws)
        env.rng.replay_pos = 0
    else:
        env.rng.mode = orc.RNG_PHILOX
        env.rng.seed = int(meta["seed"])
        env.rng.env_id = int(meta["env_id"])
        env.rng.episode = 0
    for k, v in meta["state_init"].items():
        setattr(env, STATE_ATTR[k], v)
    for k, v in meta["module_init"].items():
        setattr(env, MODULE_ATTR[k], v)
    for lo, hi, val in meta["T_init"]:
        hi = env.c.n_seg if hi is None else hi
        for i in range(lo, hi):
            env.T[i] = val
    env.disable_ignition = 1 if meta["disable_ignition"] else 0
    return env


def action_for(fx: Fixture, step: int) -> orc.Action:
    servo, tv, on, off, mode = fx.actions[fx.action_idx[step]]
    return orc.Action(servo, tv, on, off, int(mode))


def same(a: float, b: float) -> bool:
    return (a == b) or (math.isnan(a) and math.isnan(b))


def replay(fx: Fixture, *, math_mode=orc.MATH_LIBM, stencil_mode=orc.STENCIL_F32,
           exact_floats=True, float_rtol=0.0, T_atol=0.0, skip_floats=(), check_draw_count=True):
    """Step the oracle through the fixture; return a list of mismatch strings (empty = parity)."""
    env = oracle_env_for(fx, math_mode=math_mode, stencil_mode=stencil_mode)
    bad = []
    fpos = {int(s): i for i, s in enumerate(fx.float_steps)}
    tpos = {int(s): i for i, s in enumerate(fx.T_snap_steps)}
    resets = {int(at): (int(sd), si) for at, sd, si in fx.meta.get("resets", [])}
    for step in range(fx.n_steps):
        if step in resets:  # the reference's reset() of a USED environment: EDMState only (wire_edm.py:106-114)
            seed2, init2 = resets[step]
            orc.lib().wedm_oracle_reset_reference(C.byref(env))
            if fx.meta["rng"] != "native":  # reset(seed=) re-keys the stream; a native trace simply continues
                env.rng.seed, env.rng.episode = seed2, 0
            for k, v in init2.items():
                setattr(env, STATE_ATTR[k], v)
        if fx.forced is not None:
            st, y, dur, V, I = fx.forced[step]
            env.spark_state, env.spark_y, env.spark_dur = int(st), float(y), int(dur)
            env.voltage, env.current = float(V), float(I)
        act = action_for(fx, step)
        orc.step(env, act)
        early = bool(env.last_early_return)
        for name in fx.int_fields:
            if name == "ctrl_step" and early:
                continue  # the reference's early return carries no control_step key
            if name == "n_draws" and not check_draw_count:
                continue
            want = int(fx.int_row(name)[step])
            got = int(INT_GETTERS[name](env))
            if want != got:
                bad.append(f"step {step} {name}: reference {want} oracle {got}")
        if step in fpos:
            j = fpos[step]
            for name in fx.float_fields:
                if name in skip_floats:
                    continue
                want = float(fx.float_row(name)[j])
                got = float(FLOAT_GETTERS[name](env))
                if exact_floats:
                    ok = same(want, got)
                else:
                    ok = same(want, got) or abs(want - got) <= float_rtol * max(abs(want), abs(got), 1e-300)
                if not ok:
                    bad.append(f"step {step} {name}: reference {want!r} oracle {got!r}")
        if step in tpos:
            want = fx.T_snaps[tpos[step]]
            got = env.temperature().copy()
            if T_atol == 0.0:
                if not np.array_equal(want, got):
                    k = int(np.argmax(np.abs(want.astype(np.float64) - got)))
                    bad.append(f"step {step} T: max |dT| {np.max(np.abs(want.astype(np.float64) - got)):.3e} at seg {k}")
            elif np.max(np.abs(want.astype(np.float64) - got)) > T_atol:
                bad.append(f"step {step} T: max |dT| {np.max(np.abs(want.astype(np.float64) - got)):.3e}")
        if len(bad) > 20:
            break
    if fx.meta["rng"] == "native" and env.rng.replay_pos != len(fx.draws):
        bad.append(f"draw trace: consumed {env.rng.replay_pos} of {len(fx.draws)}")
    return bad, env


def replay_table(fx: Fixture) -> np.ndarray:
    """The reference's recorded draw trace (native NumPy PCG64 fixtures) as the per-step slot table of
    `wedm_bind_rng_replay` (include/wedm_hip.h): float64[n_steps, 5], NaN where nothing was drawn.  The number of
    draws of a step identifies the calls: 1 = debris roll only (it succeeded), 2 = + random-short
    roll, 3 = + ignition roll, 5 = + spark location and crater volume; 0 = a short timer was running."""
    n_draws = fx.int_row("n_draws")
    table = np.full((fx.n_steps, 5), np.nan)
    pos = 0
    for step in range(fx.n_steps):
        k = int(n_draws[step])
        assert k in (0, 1, 2, 3, 5), (step, k)
        table[step, :k] = fx.draws[pos:pos + k]
        pos += k
    assert pos == len(fx.draws)
    return table
