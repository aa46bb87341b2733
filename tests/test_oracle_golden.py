"""Pins the CPU oracle against the reference's own outputs (tests/golden, made by
tools/gen_golden.py from the reference).  Bit-exact in math mode LIBM."""
from __future__ import annotations

import json

import numpy as np
import pytest

from tests._golden import Fixture, replay

NATIVE = [
    "f1_config1_native", "f1_config1_f32action", "f2_single_spark", "f5_hard_short", "f5_debris_short",
    "f5_random_short", "f5_collision", "f5_target_reached", "f5_critical_temp", "f5_wire_break",
    "f5_action_latch", "f5_zero_fallbacks", "f6_velocity_mode", "f6_limits", "f7_gap_controller",
    # the reference's second episode (reset() of a used environment) and step() after `terminated`
    "f17_second_episode_native", "f18_past_wire_break_native",
]
SECOND_EPISODE = ["f17_second_episode_philox_env2", "f17_reset_during_short_philox_env4", "f17_stale_current_cache_philox_env5"]
PAST_TERMINATION = ["f18_past_target_philox_env1", "f18_past_wire_break_philox_env3", "f18_past_collision_philox_env6"]
PHILOX = [f"f3_philox_env{i}" for i in (0, 1, 2, 3, 777, 65535)] + ["f3_philox_config3_env5"] + [
    f"f8_geometry_{i}" for i in range(4)
] + ["f7_gap_controller_philox_env0", "f7_gap_controller_philox_env9", "f7_gap_controller_velocity_philox_env3",
     "f9_voltage_controller_philox_env2", "f9_voltage_controller_velocity_philox_env5",
     "f10_crater_statistics_philox_env1", "f13_dt2_philox_env4", "f14_copper_wire_philox_env6", "f15_default_mode_philox_env7"] + [f"f11_random_params_{k}" for k in range(8)] + SECOND_EPISODE + PAST_TERMINATION


@pytest.mark.parametrize("name", NATIVE)
def test_oracle_replays_reference_bit_exact(orc, golden_dir, name):
    """Reference's own NumPy PCG64 draws replayed into the oracle: every recorded
    integer, float64 scalar and float32 temperature must be identical."""
    fx = Fixture(golden_dir / f"{name}.npz")
    bad, _ = replay(fx, math_mode=orc.MATH_LIBM)
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("name", PHILOX)
def test_oracle_philox_matches_injected_reference(orc, golden_dir, name):
    """The reference consumed the build's Philox variates (injection shim); the oracle
    generates the same variates itself and must reproduce the reference exactly."""
    fx = Fixture(golden_dir / f"{name}.npz")
    bad, env = replay(fx, math_mode=orc.MATH_LIBM)
    assert not bad, "\n".join(bad[:20])
    if "crater_stats" in fx.data:  # MaterialRemovalModule.get_crater_statistics() of the reference run
        total, mean, std, vmin, vmax = fx.data["crater_stats"].tolist()
        assert env.spark_count == total
        if total:
            assert (env.crater_stat_min, env.crater_stat_max) == (vmin, vmax)
            assert abs(env.crater_stat_sum / total - mean) <= 1e-12 * mean
            var = max(env.crater_stat_sumsq / total - (env.crater_stat_sum / total) ** 2, 0.0)
            assert abs(var ** 0.5 - std) <= 1e-9 * max(std, 1.0)


def test_second_episode_fixtures_really_carry_module_state_over(orc, golden_dir):
    """F17 is not a fixture of two fresh episodes glued together: right after the reference's reset() the recorded
    module state is the previous episode's (wire_edm.py:106-114 re-initialises EDMState only)."""
    fx = Fixture(golden_dir / "f17_second_episode_philox_env2.npz")
    (at, _seed, _init), = fx.meta["resets"]
    pos = int(np.searchsorted(fx.float_steps, at))
    assert fx.float_steps[pos] == at
    row = {k: float(fx.float_row(k)[pos]) for k in fx.float_fields}
    assert row["debris_volume"] > 1e-5 and row["h_zone"] > 0 and row["wire_last_flow"] > 0 and row["diel_last_gap"] > 0
    assert int(fx.int_row("time")[at]) == 1 and int(fx.int_row("time")[at - 1]) == at     # a new EDMState: the clock restarts
    assert fx.data["crater_stats"][0] == (fx.int_row("spark_state") == 1).sum() / 3 or fx.data["crater_stats"][0] > 40  # one list, two episodes
    short = Fixture(golden_dir / "f17_reset_during_short_philox_env4.npz")
    (at, _seed, _init), = short.meta["resets"]
    rem = short.int_row("debris_short_remaining")
    assert rem[at - 1] > 5 and rem[at] == rem[at - 1] - 1 and short.int_row("is_short_circuit")[at] == 1   # the timer runs on
    assert float(short.float_row("workpiece_position")[np.searchsorted(short.float_steps, at)]) == 50.0     # at the new episode's gap
    stale = Fixture(golden_dir / "f17_stale_current_cache_philox_env5.npz")
    (at, _seed, _init), = stale.meta["resets"]
    cur = stale.float_row("current")
    fs = stale.float_steps
    before_latch_1 = cur[(fs < 1000) & (cur > 0)]
    before_latch_2 = cur[(fs >= at) & (fs < at + 1000) & (cur > 0)]
    assert set(before_latch_1.tolist()) == {60.0} and len(before_latch_2) and set(before_latch_2.tolist()) != {60.0}


def test_known_answers_from_survey(orc, golden_dir):
    """SURVEY.md §8c F1 known answers, recorded independently of this build."""
    fx = Fixture(golden_dir / "f1_config1_native.npz")
    bad, env = replay(fx)
    assert not bad
    assert env.c.n_seg == 400
    states = fx.int_row("spark_state")
    assert {int(k): int(v) for k, v in zip(*np.unique(states, return_counts=True))} == {0: 9004, 1: 36, -2: 960}
    assert env.workpiece_position == 50.006647428091696
    T = env.temperature()
    assert float(T.max()) == 301.2557067871094 and int(T.argmax()) == 166
    assert float(T.mean(dtype=np.float32)) == pytest.approx(293.6468200683594, abs=1e-4)
    assert env.debris_volume == 4.965168720421775e-05
    assert (float(env.h_base), float(env.h_zone)) == (15400.0, 30800.0)


def test_single_spark_known_answers(orc, golden_dir):
    """plasma index 273, T at t=50 and t=1000."""
    fx = Fixture(golden_dir / "f2_single_spark.npz")
    bad, env = replay(fx)
    assert not bad
    snaps = dict(zip(fx.T_snap_steps.tolist(), fx.T_snaps))
    t50 = snaps[49]  # after the 50th step
    assert [float(x) for x in t50[272:275]] == [293.2237243652344, 300.46075439453125, 293.2237243652344]
    T = env.temperature()
    assert [float(x) for x in T[272:275]] == [296.1844787597656, 297.6618957519531, 296.1844787597656]
    assert (float(env.h_base), float(env.h_zone)) == (14000.0, 14896.0)


def test_invalid_mode_flags_where_reference_raises(orc, golden_dir):
    """material.py:108-113 raises ValueError at the first fresh spark with an even
    mode; the oracle sets its error flag on exactly that step."""
    fx = Fixture(golden_dir / "f5_invalid_mode.npz")
    step_raised, _msg = fx.meta["raised"]
    bad, env = replay(fx)
    # the aborted step drew u_debris, u_random, u_ignite, y before material.update raised
    assert bad == [f"draw trace: consumed {len(fx.draws) - 4} of {len(fx.draws)}"]
    assert env.error == 0
    from tests._golden import action_for

    orc.step(env, action_for(fx, fx.n_steps - 1))
    assert env.error & 1 and fx.n_steps == step_raised


def test_portable_math_mode_stays_within_stated_tolerance(orc, golden_dir):
    """PORTABLE math (what the GPU computes) vs the reference: decisions identical,
    float64 state within 1e-12 relative, temperatures within 1e-4 K."""
    for name in ("f1_config1_native", "f5_debris_short", "f3_philox_env0", "f7_gap_controller", "f9_voltage_controller_philox_env2",
                 "f10_crater_statistics_philox_env1", "f13_dt2_philox_env4", "f14_copper_wire_philox_env6",
                 "f15_default_mode_philox_env7") + tuple(f"f11_random_params_{k}" for k in range(8)):
        fx = Fixture(golden_dir / f"{name}.npz")
        bad, _ = replay(fx, math_mode=orc.MATH_PORTABLE, exact_floats=False, float_rtol=1e-12, T_atol=1e-4)
        assert not bad, name + "\n" + "\n".join(bad[:20])


def test_numba_typing_of_the_stencil_is_within_tolerance(orc, golden_dir):
    """float64-intermediate stencil (how real Numba types wire.py:58-123) vs the
    float32 path the stubbed reference runs: |dT| <= 1.3e-4 K over 10 000 steps."""
    fx = Fixture(golden_dir / "f1_config1_native.npz")
    bad, _ = replay(fx, stencil_mode=orc.STENCIL_F64, exact_floats=False, float_rtol=1e-6, T_atol=1.3e-4,
                    skip_floats=("tmax",))
    assert not bad, "\n".join(bad[:20])


def test_geometry_table(orc, golden_dir):
    """F4: derived constants of WireModule/Dielectric/Mechanics constructors."""
    z = np.load(golden_dir / "f4_geometry_table.npz")
    cols = json.loads(str(z["columns"]))
    import ctypes as C

    for row in z["table"]:
        r = dict(zip(cols, row))
        cfg = orc.default_config(workpiece_height=r["h"], wire_diameter=r["d"], segment_len=r["seg"],
                                 buffer_len_bottom=r["buf_bottom"], buffer_len_top=r["buf_top"],
                                 contact_offset_bottom=r["off_bottom"], contact_offset_top=r["off_top"])
        c = orc.Consts()
        assert orc.lib().wedm_oracle_derive(C.byref(cfg), C.byref(c)) == 0
        for k in ("n_seg", "zone_start", "zone_end", "az_start", "az_end", "contact_bottom", "contact_top"):
            assert getattr(c, k) == int(r[k]), (k, r)
        for k in ("k_cond", "tuf", "a_surf", "s_area", "joule_geom", "critical_temperature",
                  "breaking_temperature", "cavity_coeff", "debris_removal_per_us", "damping_coeff",
                  "stiffness_coeff", "max_jerk_dt", "dt_s"):
            assert getattr(c, k) == r[k], (k, getattr(c, k), r[k], r)


def test_native_draw_trace_converts_to_the_replay_slot_table(golden_dir):
    """Host side of the device's variate-injection mode (wedm_bind_rng_replay): the reference's own PCG64 draws of
    fixture F1, laid out by step and slot."""
    from tests._golden import Fixture, replay_table

    fx = Fixture(golden_dir / "f1_config1_native.npz")
    table = replay_table(fx)
    assert table.shape == (10000, 5)
    drawn = ~np.isnan(table)
    assert drawn.sum() == len(fx.draws) == 29028        # 29 004 random + 12 uniform + 12 normal
    assert (drawn.sum(axis=1) == fx.int_row("n_draws")).all()
    sparks = drawn[:, 4].sum()
    assert sparks == 12 and np.all(table[drawn[:, 3], 3] < 20.0) and np.all(table[drawn[:, 4], 4] > 0)
