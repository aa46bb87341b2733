"""SURVEY.md §8f-1/-3 on top of the device trace: the per-microsecond signal ring, the logger fed
from it, and the PI voltage controller of the reference's driver.  CPU tests run the host logic
against the oracle test seam (which restates the sampling schedule of include/wedm_hip.h); the
GPU tests in test_gpu_parity.py compare the kernels' rings with these, bit for bit."""
from __future__ import annotations

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from sparc_amd import GapController, SimulationLogger, VoltageController, WireEDMEnv, run_controlled
from tests._fixture_env import check_step
from tests._golden import Fixture
from tests._oracle_backend import OracleBackend


class LibmOracleBackend(OracleBackend):
    math_mode = orc.MATH_LIBM  # glibc pow/exp like the reference -> bit-exact against its fixtures


def driver_env(n, seed, mode="position", backend=LibmOracleBackend):
    env = WireEDMEnv(num_envs=n, device="cpu", backend=backend, mechanics_control_mode=mode)
    env.reset(seed=seed)
    env.state.workpiece_position = 70.0   # experiments/run_simulation.py:199-201
    env.state.wire_position = 10.0
    env.state.target_position = 5000.0
    return env


def sparking_env(n, seed=5, backend=OracleBackend):
    env = WireEDMEnv(num_envs=n, device="cpu", backend=backend)
    env.reset(seed=seed)
    env.state.workpiece_position = 25.0
    env.state.wire_position = 10.0
    env.state.target_position = 5000.0
    return env


def test_trace_equals_per_microsecond_reads_and_ring_wraps():
    a, b = sparking_env(6), sparking_env(6)
    names = ["voltage", "current", "time", "spark_state", "is_short_circuit", "wire_position", "spark_status"]
    trace = a.bind_trace(names, every=1, capacity=700, envs=(2, 3), wire_temperature=True)
    act_a, act_b = a.make_action(), b.make_action()
    want = {k: [] for k in names + ["wire_temperature"]}
    for launch in (300, 250, 400):       # 950 samples through a 700-slot ring
        a.step_many(act_a, launch)
        for _ in range(launch):
            b.step(act_b)
            for k in names:
                # (a trace records the rows the kernels carry: the clock's low 32 bits; `state.time` is the 64-bit value)
                v = getattr(b.state, {"spark_status": "spark_state", "time": "time_low32"}.get(k, k))
                want[k].append(v[2:5].clone())
            want["wire_temperature"].append(b.state.wire_temperature[2:5].clone())
    assert trace.count == 950
    got = trace.read()
    for k in names + ["wire_temperature"]:
        ref = torch.stack(want[k])[-700:]
        assert got[k].shape == ref.shape and got[k].dtype == ref.dtype, k
        assert torch.equal(got[k], ref), k
    assert int((got["spark_state"] == 1).sum()) > 0            # the window really contains sparks
    last = trace.read(last=10, names=["time"])["time"]
    assert last[:, 0].tolist() == list(range(941, 951))
    mid = trace.read_range(400, 420, ["voltage"])["voltage"]
    assert torch.equal(mid, torch.stack(want["voltage"])[400:420])
    with pytest.raises(RuntimeError, match="overrun"):
        trace.read_range(100, 120)
    assert trace.sample_times(0, 3).tolist() == [1, 2, 3]


def test_trace_interval_phase_across_launches_and_rebind():
    a, b = sparking_env(4), sparking_env(4)
    trace = a.bind_trace(["time", "workpiece_position"], every=7, capacity=64)
    act = a.make_action()
    for launch in (10, 3, 1, 30, 6):     # 50 us -> samples at 7, 14, ..., 49
        a.step_many(act, launch)
    assert trace.count == 7
    assert trace.read()["time"][:, 0].tolist() == [7, 14, 21, 28, 35, 42, 49]
    wp = []
    for t in range(50):
        b.step(b.make_action())
        if (t + 1) % 7 == 0:
            wp.append(b.state.workpiece_position.clone())
    assert torch.equal(trace.read()["workpiece_position"], torch.stack(wp))
    # binding again restarts the count and the phase
    t2 = a.bind_trace(["time"], every=5, capacity=4)
    assert trace.count == 0 and t2.count == 0
    a.step_many(act, 12)
    assert t2.read()["time"][:, 0].tolist() == [55, 60]
    a.unbind_trace()
    a.step_many(act, 5)
    assert t2.count == 0
    for bad in (dict(every=0), dict(capacity=0), dict(envs=(3, 2)), dict(envs=(-1, 2))):
        with pytest.raises(ValueError):
            a.bind_trace(["time"], **bad)
    with pytest.raises(ValueError, match="unknown signal"):
        a.bind_trace(["no_such_field"])
    with pytest.raises(ValueError, match="nothing"):
        a.bind_trace([])


def test_terminated_environments_keep_being_sampled():
    env = sparking_env(3)
    env.state.target_position[0] = 25.0005          # the first crater finishes environment 0
    trace = env.bind_trace(["time", "done", "workpiece_position"], capacity=3000)
    env.step_many(env.make_action(), 3000)
    got = trace.read()
    done0 = got["done"][:, 0]
    assert bool(done0[-1]) and not bool(done0[0]) and not got["done"][:, 1:].any()
    first = int(torch.nonzero(done0)[0])
    assert (got["time"][first:, 0] == got["time"][first, 0]).all()      # frozen state, sampled on
    assert got["time"][:, 1].tolist() == list(range(1, 3001))


@pytest.mark.parametrize("name,n,use_ring", [
    ("f9_voltage_controller_philox_env2", 4, False),
    ("f9_voltage_controller_velocity_philox_env5", 8, False),
    ("f9_voltage_controller_dt2_philox_env4", 6, False),        # config.dt = 2: 501-sample window
    ("f9_voltage_controller_servo500_philox_env6", 8, False),   # the 1 ms window spans two control intervals
    ("f9_voltage_controller_philox_env2", 4, True),             # the per-microsecond ring kept as a fallback
    ("f9_voltage_controller_dt2_philox_env4", 6, True),
])
def test_voltage_controller_reproduces_the_reference_driver(golden_dir, name, n, use_ring):
    """run_simulation.py's loop with the reference's OWN `create_voltage_controller` (fixture
    generated by importing it) against the on-device VoltageController reading the running voltage
    sum the step kernels publish at control steps (rows VOLT_ACC / VOLT_SUM): every recorded quantity
    and every servo command equal, bit for bit."""
    from tests._fixture_env import env_from_fixture

    fx = Fixture(golden_dir / (name + ".npz"))
    env_id = int(fx.meta["env_id"])
    env = env_from_fixture(fx, n, device="cpu", backend=LibmOracleBackend)
    ctl = VoltageController(30.0, use_ring=use_ring)
    action = ctl(env)
    assert float(action.servo[env_id]) == fx.actions[fx.action_idx[0], 0]
    checked = 0
    for step in range(fx.n_steps):
        env.step(action)
        check_step(env, fx, env_id, step, exact_floats=True)
        if bool(env.state.control_step[0]):
            action = ctl(env)
            if step + 1 < fx.n_steps:   # the float32 servo leaf the reference computed at this control step
                assert float(action.servo[env_id]) == fx.actions[fx.action_idx[step + 1], 0], step
                checked += 1
    assert checked >= 5 and len(fx.actions) >= 5
    assert np.array_equal(env.state.wire_temperature[env_id].numpy(), fx.T_snaps[-1])


def test_voltage_controller_fused_launches_equal_per_microsecond_driver():
    a, b = driver_env(8, 6, backend=OracleBackend), driver_env(8, 6, backend=OracleBackend)
    for env in (a, b):
        env.state.workpiece_position = 22.0       # close enough to spark: a non-trivial voltage average
    ctl_a, ctl_b = VoltageController(30.0), VoltageController(30.0)
    assert run_controlled(a, ctl_a, 5300) == 5300
    action = ctl_b(b)
    for _ in range(5300):
        b.step(action)
        if bool(b.state.control_step[0]):
            action = ctl_b(b)
    A, B = a.state.clone_blocks(), b.state.clone_blocks()
    for k in ("i32", "i8", "T", "obs"):
        assert torch.equal(A[k], B[k]), k
    assert bool(((A["f64"] == B["f64"]) | (A["f64"].isnan() & B["f64"].isnan())).all())
    assert torch.equal(ctl_a.integral_error, ctl_b.integral_error)
    avg = ctl_a.average_voltage()
    assert bool(((avg > 0) & (avg < 80)).all()) and int(a.state.spark_count.sum()) > 50


def test_logger_attached_to_the_device_trace(tmp_path):
    """every_step / interval logging of a fused run == the reference-style collect() after every
    1-us step, including the signals of run_simulation.py:127-147."""
    signals = ["time", "voltage", "current", "wire_position", "wire_velocity", "workpiece_position", "target_delta",
               "debris_concentration", "dielectric_flow_rate", "is_short_circuit", "flow_rate", "wire_temperature",
               "wire_average_temperature", "spark_status"]
    a, b = driver_env(3, 4, backend=OracleBackend), driver_env(3, 4, backend=OracleBackend)
    for env in (a, b):
        env.state.workpiece_position = 24.0
    cfg = {"signals_to_log": signals, "log_frequency": {"type": "every_step"},
           "backend": {"type": "numpy", "filepath": str(tmp_path / "run.npz"), "compress": True}}
    fused = SimulationLogger(cfg, env_reference=a)
    fused.attach(a)
    run_controlled(a, GapController(), 2300, logger=fused)
    per_us = SimulationLogger({**cfg, "backend": {"type": "memory"}}, env_reference=b)
    ctl = GapController()
    action = ctl(b)
    for _ in range(2300):
        _, _, _, _, info = b.step(action)
        per_us.collect(b.state, info)
        if bool(b.state.control_step[0]):
            action = ctl(b)
    F, P = fused.get_data(), per_us.get_data()
    for k in signals:
        assert F[k].shape == P[k].shape and F[k].shape[:2] == (2300, 3), k
        if k == "wire_average_temperature":
            assert np.allclose(F[k], P[k], rtol=0, atol=1e-4), k
        else:
            assert np.array_equal(F[k], P[k], equal_nan=True), k
    assert (F["dielectric_flow_rate"] == F["flow_rate"] * 100.0 / 1e9).all()   # dielectric.py:160-162
    z = np.load(tmp_path / "run.npz")
    assert np.array_equal(z["voltage"], F["voltage"]) and z["wire_temperature"].shape == (2300, 3, a.n_segments)
    # interval logging: every 50th microsecond (logger.py:130-132), on a sub-range of environments
    c = driver_env(5, 4, backend=OracleBackend)
    lg = SimulationLogger({"signals_to_log": ["time", "voltage"], "log_frequency": {"type": "interval", "value": 50}})
    lg.attach(c, envs=(1, 2))
    run_controlled(c, GapController(), 1230, logger=lg)
    t = lg.get_data()["time"]
    assert t.shape == (24, 2) and t[:, 0].tolist() == list(range(50, 1201, 50))
    with pytest.raises(ValueError):
        SimulationLogger({"log_frequency": {"type": "control_step"}}).attach(c)


@pytest.mark.parametrize("name", ["f16_logger_philox_env3", "f16_logger_velocity_philox_env1"])
def test_logger_output_equals_the_reference_loggers_own_output(golden_dir, name):
    """F16: the reference's `SimulationLogger` over its own driver loop and signal list (every_step,
    interval, control_step; incl. `dielectric_flow_rate`'s 1e-9 scaling and the zone mean cached every
    `zone_mean_interval` wire updates) against the build's logger fed from the device trace of fused
    launches — every key, bit for bit on the LIBM oracle seam."""
    from tests._fixture_env import run_logger_fixture

    assert run_logger_fixture(golden_dir / f"{name}.npz", device="cpu", backend=LibmOracleBackend, exact=True) == 38


def test_logger_and_voltage_controller_share_one_trace():
    env = driver_env(4, 11, backend=OracleBackend)
    env.state.workpiece_position = 22.0
    trace = env.bind_trace(["voltage", "time", "current"], every=1, capacity=2048)
    ctl = VoltageController(30.0).bind(env, trace)
    lg = SimulationLogger({"signals_to_log": ["time", "voltage"], "log_frequency": {"type": "every_step"}})
    lg.attach(env, trace=trace)
    run_controlled(env, ctl, 3200, logger=lg)
    d = lg.get_data()
    assert d["time"][:, 0].tolist() == list(range(1, 3201))
    assert float(ctl.average_voltage()[0]) == float(d["voltage"][-1001:, 0].sum() / 1001)
    with pytest.raises(ValueError):
        VoltageController().bind(env, env.bind_trace(["voltage"], every=2, capacity=2000))


# ------------------------------------------------------------------ running statistics (§8f-4)
def crater_env(n, backend, device="cpu", **kw):
    env = WireEDMEnv(num_envs=n, device=device, backend=backend, **kw)
    env.reset(seed=81)
    env.state.workpiece_position = 22.0
    env.state.wire_position = 10.0
    env.state.target_position = 5000.0
    return env


def test_crater_statistics_match_the_reference(golden_dir):
    """`MaterialRemovalModule.get_crater_statistics()` of the reference after 12 000 us of dense
    sparking (fixture F10, 140 craters) against the running statistics kept at every fresh spark:
    count / min / max exact, mean and std to 1e-12 (np.mean / np.std sum pairwise, the kernel
    updates Welford's recurrence; both are float64)."""
    fx = Fixture(golden_dir / "f10_crater_statistics_philox_env1.npz")
    env = crater_env(3, LibmOracleBackend, crater_log_capacity=256)
    small = crater_env(3, LibmOracleBackend, crater_log_capacity=32)     # a ring shorter than the list
    act = env.make_action(0.05, 80.0, 13, 2.0, 20.0)
    assert (env.get_crater_statistics()["total_craters"] == 0).all()
    trace = env.bind_trace(["last_crater_volume", "spark_state", "spark_duration"], capacity=fx.n_steps, envs=(1, 1))
    for _ in range(12):
        env.step_many(act, 1000)
    check_step(env, fx, 1, fx.n_steps - 1, exact_floats=True)
    total, mean, std, vmin, vmax = fx.data["crater_stats"].tolist()
    got = {k: v[1].item() for k, v in env.get_crater_statistics().items()}
    assert got["total_craters"] == total == 140
    assert got["min_volume_um3"] == vmin and got["max_volume_um3"] == vmax
    assert abs(got["mean_volume_um3"] - mean) <= 1e-12 * mean
    assert abs(got["std_volume_um3"] - std) <= 1e-12 * std
    # the full list of the reference (`volumes_um3`) is recoverable from the per-microsecond trace
    t = trace.read()
    fresh = (t["spark_state"][:, 0] == 1) & (t["spark_duration"][:, 0] == 0)
    vols_mm3 = t["last_crater_volume"][:, 0][fresh].numpy()
    assert np.array_equal(vols_mm3, fx.data["crater_volumes_um3"] / 1e9)
    # ... and directly from the crater log the kernels keep (`crater_volumes_um3`, material.py:133)
    assert np.array_equal(env.get_crater_volumes(1).numpy(), fx.data["crater_volumes_um3"])
    for _ in range(12):
        small.step_many(small.make_action(0.05, 80.0, 13, 2.0, 20.0), 1000)
    assert np.array_equal(small.get_crater_volumes(1).numpy(), fx.data["crater_volumes_um3"][-32:])
    with pytest.raises(RuntimeError):
        crater_env(2, OracleBackend).get_crater_volumes(0)
    # environments without craters report zeros, like the reference's empty case
    idle = WireEDMEnv(num_envs=2, device="cpu", backend=OracleBackend)
    idle.reset(seed=1)
    idle.step_many(idle.make_action(), 100)
    assert all(float(v.abs().sum()) == 0 for v in idle.get_crater_statistics().values())
    # a partial reset clears the statistics of the selected environments only
    env.reset(options={"mask": torch.tensor([False, True, False])})
    st = env.get_crater_statistics()
    assert st["total_craters"].tolist()[1] == 0 and st["mean_volume_um3"][1] == 0 and st["total_craters"][0] > 0


@pytest.mark.parametrize("k", range(8))
def test_randomized_parameter_fixtures_on_the_batched_environment(golden_dir, k):
    """F11: eight reference runs with randomized parameters in every module (random shorts on,
    odd servo intervals, velocity mode, coarse wires ...).  The batched environment (host
    derivation of all constants + SoA layout + fused launches) must reproduce the reference's
    whole trajectory, read back through the device trace: bit for bit in LIBM math mode."""
    from tests._fixture_env import env_from_fixture, run_fixture_through_trace

    fx = Fixture(golden_dir / f"f11_random_params_{k}.npz")
    env = env_from_fixture(fx, 64, device="cpu", backend=LibmOracleBackend)
    got = run_fixture_through_trace(env, fx, exact_floats=True)
    assert (got["spark_state"] == 1).sum() > 30


def test_two_microsecond_physics_step_fixture(golden_dir):
    """F13: config.dt = 2 (clocks and the servo integrate 2 us per step, the wire keeps its 1 us
    update factor, wire.py:192-196) on the batched environment, whole trajectory, bit for bit."""
    from tests._fixture_env import env_from_fixture, run_fixture_through_trace

    fx = Fixture(golden_dir / "f13_dt2_philox_env4.npz")
    env = env_from_fixture(fx, 8, device="cpu", backend=LibmOracleBackend)
    got = run_fixture_through_trace(env, fx, exact_floats=True)
    assert got["time"][:3].tolist() == [2, 4, 6] and (got["spark_state"] == 1).sum() > 30


def test_custom_wire_material_fixture(golden_dir):
    """F14: a copper wire registered in the material database (core/material_db.py:67-73) — every
    derived constant of the wire module changes; oracle-seam run == reference, whole trajectory."""
    from tests._fixture_env import env_from_fixture, run_fixture_through_trace

    fx = Fixture(golden_dir / "f14_copper_wire_philox_env6.npz")
    env = env_from_fixture(fx, 8, device="cpu", backend=LibmOracleBackend)
    assert env.wire_material.name == "copper" and env.wire_material.thermal_conductivity == 401
    got = run_fixture_through_trace(env, fx, exact_floats=True)
    assert (got["spark_state"] == 1).sum() > 50


def test_default_modes_before_the_first_latch_fixture(golden_dir):
    """F15: until the first control step `state.current_mode` is None.  The reference's ignition
    module then HITS its freshly initialised cache (`_cached_current_mode = None`, 60 A,
    ignition.py:79-81) and never consults `default_current_mode` (set to I13 here on purpose); the
    material module falls back to I1."""
    from tests._fixture_env import env_from_fixture, run_fixture_through_trace

    fx = Fixture(golden_dir / "f15_default_mode_philox_env7.npz")
    env = env_from_fixture(fx, 8, device="cpu", backend=LibmOracleBackend)
    got = run_fixture_through_trace(env, fx, exact_floats=True)
    early = (got["spark_state"][:1000] == 1)
    assert early.sum() > 5 and set(np.unique(got["current"][:1000][early]).tolist()) == {60.0}   # not I13's 215 A


COMPAT_FIXTURES = ["f17_second_episode_philox_env2", "f17_reset_during_short_philox_env4", "f17_stale_current_cache_philox_env5",
                   "f18_past_target_philox_env1", "f18_past_wire_break_philox_env3", "f18_past_collision_philox_env6"]


@pytest.mark.parametrize("name", COMPAT_FIXTURES)
def test_reference_compatible_reset_and_stepping_past_termination(golden_dir, name):
    """F17 / F18 on the batched environment: `WireEDMEnv(reset_semantics="reference")` re-initialises EDMState only, as
    the reference's reset() of a used environment does (wire_edm.py:106-114: short timers, current cache, debris, flow /
    density / convection caches, prev_accel and the crater list live on in its module objects), and
    `freeze_terminated=False` keeps stepping a terminated environment as the reference's unguarded step() does
    (wire_edm.py:116-157; after a wire break: wire.py:260-261 + the early return of wire_edm.py:129-130).  Whole
    trajectories, bit for bit on the LIBM seam: once microsecond by microsecond through `step()`, once in fused launches
    read back through the device trace."""
    from tests._fixture_env import compat_env_from_fixture, run_fixture_stepwise, run_fixture_through_trace

    fx = Fixture(golden_dir / f"{name}.npz")
    env = compat_env_from_fixture(fx, 8, device="cpu", backend=LibmOracleBackend)
    assert run_fixture_stepwise(env, fx, exact_floats=True, every=7) > fx.n_steps // 8
    env = compat_env_from_fixture(fx, 8, device="cpu", backend=LibmOracleBackend)
    got = run_fixture_through_trace(env, fx, exact_floats=True)
    if name.startswith("f18"):
        first = int(np.argmax(fx.int_row("terminated") != 0))
        assert fx.int_row("terminated")[first:].all() and fx.n_steps - first >= 500   # >= 500 us past `terminated`
        assert got["done"][first:].all() and not got["done"][:first].any()


def test_default_reset_and_freeze_differ_from_the_reference_where_documented(golden_dir):
    """The defaults stay what they were (DESIGN.md deviations 1 and 2): a full reset forgets the module state the
    reference would carry over, and a terminated environment is frozen."""
    from tests._fixture_env import env_from_fixture, fixture_segments, apply_fixture_reset

    fx = Fixture(golden_dir / "f17_second_episode_philox_env2.npz")
    env = env_from_fixture(fx, 8, device="cpu", backend=LibmOracleBackend)        # reset_semantics="full"
    (lo, hi, _), (lo2, hi2, reset) = fixture_segments(fx)
    act = env.make_action(*[fx.actions[0][i] for i in (0, 1)], int(fx.actions[0][4]), fx.actions[0][2], fx.actions[0][3])
    env.step_many(act, hi - lo)
    assert float(env.state.debris_volume[2]) > 1e-5
    apply_fixture_reset(env, reset)
    assert float(env.state.debris_volume[2]) == 0.0 and float(env.state.h_eff_zone[2]) == 0.0 and int(env.state.spark_count[2]) == 0
    fz = Fixture(golden_dir / "f18_past_target_philox_env1.npz")
    env = env_from_fixture(fz, 8, device="cpu", backend=LibmOracleBackend)        # freeze_terminated=True
    a = fz.actions[0]
    env.step_many(env.make_action(a[0], a[1], int(a[4]), a[2], a[3]), fz.n_steps)
    first = int(np.argmax(fz.int_row("terminated") != 0))
    assert int(env.state.time[1]) == first + 1 and bool(env.state.done[1])      # frozen at the terminating step
