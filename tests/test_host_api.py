"""CPU-side tests of the host layer: the reference's own smoke tests (tests/test_state.py,
tests/test_env_integration.py of the reference) restated for the batched environment,
plus configuration, derivation and action plumbing.  The physics backend here is the
oracle test seam (tests/_oracle_backend.py); the product itself has no CPU path."""
from __future__ import annotations

import json

import numpy as np
import pytest
import torch

from sparc_amd import (DielectricModuleParameters, EnvironmentConfig, IgnitionModuleParameters, MaterialDatabase,
                       MaterialModuleParameters, MechanicsModuleParameters, WireEDMEnv, WireModuleParameters,
                       get_material_db)
from sparc_amd.core import derive
from tests._oracle_backend import OracleBackend


def make(n=4, **kw):
    return WireEDMEnv(num_envs=n, device="cpu", backend=OracleBackend, **kw)


# ---- reference tests/test_env_integration.py, batched --------------------------------------
def test_env_creation():
    env = make()
    assert env is not None and env.config is not None


def test_env_reset():
    env = make()
    obs, info = env.reset()
    assert obs is not None and isinstance(info, dict)
    assert (env.state.time == 0).all()
    assert (env.state.workpiece_position == env.config.initial_gap).all()
    assert (env.state.target_position == env.config.target_cutting_distance).all()


def test_env_step_with_sampled_action():
    env = make()
    env.reset(seed=0)
    action = env.action_space.sample(env.np_random)
    action["generator_control"]["current_mode"] = np.array([5], dtype=np.int32)  # a mode with crater data
    obs, reward, terminated, truncated, info = env.step(action)
    assert (env.state.time > 0).all()
    assert reward.shape == (4,) and terminated.dtype == torch.bool and truncated.dtype == torch.bool
    assert set(info) == {"wire_broken", "target_reached", "spark_state", "time", "control_step"}
    # the exact 64-bit clock, composed when it is read (the int32 row read -2**31 once an environment had passed 2**31 us)
    assert info["time"].dtype == torch.int64 and info["time"].tolist() == [1] * 4 and dict(info.items())["time"] is not None
    env.state.time = 2**31 + 5
    assert info["time"].tolist() == [2**31 + 5] * 4 and info.get("time").dtype == torch.int64


def test_custom_config():
    cfg = EnvironmentConfig(workpiece_height=20.0, wire_diameter=0.3, target_cutting_distance=1000.0)
    env = make(config=cfg)
    assert env.config.wire_diameter == 0.3 and env.config.target_cutting_distance == 1000.0
    assert env.workpiece_height == 20.0 and env.wire_diameter == 0.3


def test_control_modes():
    assert make(mechanics_control_mode="position").mechanics_control_mode == "position"
    assert make(mechanics_control_mode="velocity").mechanics.control_mode == "velocity"
    with pytest.raises(ValueError):
        make(mechanics_control_mode="invalid")


# ---- reference tests/test_state.py, batched ------------------------------------------------
def test_state_defaults_and_assignment():
    env = make()
    env.reset(seed=1)
    s = env.state
    assert (s.wire_position == 0.0).all() and (s.time == 0).all()
    s.wire_position = 10.0
    s.workpiece_position = torch.tensor([15.0, 16.0, 17.0, 18.0])
    s.target_position = 100.0
    assert (s.wire_position == 10.0).all()
    assert s.workpiece_position.tolist() == [15.0, 16.0, 17.0, 18.0]
    assert (s.target_position == 100.0).all()
    assert s.wire_temperature.shape == (4, env.n_segments)
    st, y, dur = s.spark_status
    assert (st == 0).all() and y.isnan().all() and (dur == 0).all()


# ---- configuration -------------------------------------------------------------------------
@pytest.mark.parametrize("field,value", [("workpiece_height", 0.0), ("wire_diameter", -1.0), ("initial_gap", 0.0),
                                         ("target_cutting_distance", 0.0), ("dt", 0), ("servo_interval", 0),
                                         ("max_wire_temperature", 293.15)])
def test_config_validate_raises_like_the_reference(field, value):
    cfg = EnvironmentConfig(**{field: value})
    with pytest.raises(ValueError):
        cfg.validate()
    with pytest.raises(ValueError):
        make(config=cfg)


def test_config_dict_json_roundtrip(tmp_path):
    cfg = EnvironmentConfig(workpiece_height=12.5, wire_diameter=0.25, servo_interval=500)
    d = cfg.to_dict()
    assert EnvironmentConfig.from_dict({**d, "unknown_key": 1}) == cfg
    cfg.to_json(tmp_path / "c.json")
    assert EnvironmentConfig.from_json(tmp_path / "c.json") == cfg
    assert json.loads((tmp_path / "c.json").read_text())["servo_interval"] == 500


def test_material_database():
    db = get_material_db()
    brass = db.get_wire_material("brass")
    assert (brass.density, brass.specific_heat, brass.melting_point, brass.breaking_temperature) == (8400, 377, 1173, 1500)
    with pytest.raises(ValueError):
        db.get_wire_material("unobtainium")
    with pytest.raises(ValueError):
        make(config=EnvironmentConfig(wire_material="unobtainium"))


def test_material_database_save_and_reload(tmp_path):
    db = MaterialDatabase(tmp_path)
    db.save_materials()
    assert MaterialDatabase(tmp_path).get_wire_material("brass").thermal_conductivity == 120


def test_parameter_dataclass_defaults_match_the_reference():
    i, w = IgnitionModuleParameters(), WireModuleParameters()
    assert (i.sigmoid_steepness, i.debris_short_duration, i.random_short_max_probability) == (500.0, 50, 0.0)
    assert (i.ignition_a_coeff, i.ignition_b_coeff, i.ignition_c_coeff, i.default_current_mode) == (0.48, -3.69, 14.05, "I5")
    assert (w.segment_len, w.spool_T, w.base_convection_coefficient, w.plasma_efficiency) == (0.2, 293.15, 14000, 0.1)
    assert MaterialModuleParameters().base_overcut == 0.12
    d, m = DielectricModuleParameters(), MechanicsModuleParameters()
    assert (d.base_flow_rate, d.debris_removal_efficiency, d.reference_gap) == (100.0, 0.01, 25.0)
    assert (m.omega_n, m.zeta, m.max_acceleration, m.max_jerk, m.max_speed) == (235.0, 0.38, 3.0e5, 1.0e8, 3.0e4)


# ---- host-side derivation vs the reference's constructors (fixture F4) ----------------------
def test_derive_geometry_matches_reference_table(golden_dir):
    z = np.load(golden_dir / "f4_geometry_table.npz")
    cols = json.loads(str(z["columns"]))
    brass = get_material_db().get_wire_material("brass")
    for row in z["table"]:
        r = dict(zip(cols, row))
        wire = WireModuleParameters(segment_len=r["seg"], buffer_len_bottom=r["buf_bottom"], buffer_len_top=r["buf_top"],
                                    contact_offset_bottom=r["off_bottom"], contact_offset_top=r["off_top"])
        g = derive.derive_geometry(r["h"], r["d"], wire, brass, MaterialModuleParameters())
        for k in ("n_seg", "zone_start", "zone_end", "az_start", "az_end", "contact_bottom", "contact_top"):
            assert getattr(g, k) == int(r[k]), (k, r)
        for k in ("k_cond", "tuf", "a_surf", "s_area", "joule_geom", "cavity_coeff"):
            assert getattr(g, k) == r[k], (k, r)
        p = derive.build_params(EnvironmentConfig(workpiece_height=r["h"], wire_diameter=r["d"]), "position",
                                IgnitionModuleParameters(), wire, MaterialModuleParameters(),
                                DielectricModuleParameters(), MechanicsModuleParameters(), brass, geometry=g)
        for k in ("critical_temperature", "breaking_temperature", "debris_removal_per_us", "damping_coeff",
                  "stiffness_coeff", "max_jerk_dt", "dt_s"):
            assert getattr(p, k) == r[k], (k, r)


def test_survey_geometry_known_answers():
    """(h, seg, d) -> (n_seg, zone, contacts)."""
    brass = get_material_db().get_wire_material("brass")
    cases = {(20, .2, .2): (400, 149, 248, 100, 300), (20, .625, .2): (128, 48, 80, 32, 96),
             (10, .2, .2): (350, 149, 198, 100, 250), (15, .2, .25): (375, 149, 223, 100, 275),
             (30, .2, .3): (450, 149, 298, 100, 350)}
    for (h, seg, d), want in cases.items():
        g = derive.derive_geometry(h, d, WireModuleParameters(segment_len=seg), brass, MaterialModuleParameters())
        assert (g.n_seg, g.zone_start, g.zone_end, g.contact_bottom, g.contact_top) == want


# ---- action plumbing -----------------------------------------------------------------------
def test_action_leaf_shapes_and_broadcast():
    env = make(n=6)
    env.reset(seed=3)
    a = env.make_action(servo=np.linspace(-1, 1, 6), target_voltage=torch.full((6, 1), 90.0), current_mode=[1, 3, 5, 7, 9, 11],
                        ON_time=2.0, OFF_time=np.array([30.0]))
    assert a.servo.dtype == torch.float64 and a.current_mode.dtype == torch.int32
    assert a.target_voltage.shape == (6,) and a.on_time.tolist() == [2.0] * 6
    with pytest.raises(ValueError):
        env.make_action(servo=np.zeros(5))


def test_invalid_mode_raises_like_material_module():
    env = make()
    env.reset(seed=0)
    for bad in (2, 4, 18, 19):
        with pytest.raises(ValueError, match="not available in crater data"):
            env.step(env.make_action(current_mode=bad))


def test_action_is_latched_only_on_control_steps():
    env = make(n=2)
    env.reset(seed=5)
    env.step_many(env.make_action(servo=0.7, target_voltage=123.0), 1000)
    assert (env.state.target_delta == 0.0).all() and (env.state.target_voltage == 0.0).all()  # None until call 1001
    _, _, _, _, info = env.step(env.make_action(servo=0.7, target_voltage=123.0))
    assert info["control_step"].all()
    assert (env.state.target_delta == 0.7).all() and (env.state.target_voltage == 123.0).all()
    assert (env.state.time_since_servo == 1).all()


def test_partial_reset_mask_and_seeding():
    env = make(n=8)
    env.reset(seed=11)
    env.step_many(env.make_action(), 50)
    mask = np.array([1, 0, 0, 1, 0, 0, 0, 1], dtype=bool)
    env.reset(options={"mask": mask})
    assert env.state.time.tolist() == [0, 50, 50, 0, 50, 50, 50, 0]
    assert env.state.episode.tolist() == [1, 0, 0, 1, 0, 0, 0, 1]
    a, b = make(n=8), make(n=8)
    for e in (a, b):
        e.reset(seed=99)
        e.state.workpiece_position = 25.0
        e.state.wire_position = 10.0
        e.step_many(e.make_action(), 1500)
    assert bool(((a.state.f64 == b.state.f64) | (a.state.f64.isnan() & b.state.f64.isnan())).all())
    assert torch.equal(a.state.T, b.state.T)
    assert int(a.state.spark_count.sum()) > 0


def test_per_environment_geometry_rows():
    h = np.array([10.0, 20.0, 30.0, 12.3])
    d = np.array([0.10, 0.20, 0.30, 0.15])
    env = make(n=4, workpiece_height=h, wire_diameter=d)
    assert env.params.per_env_geometry == 1 and env.n_segments == 450
    assert env._geom_i32[0, :4].tolist() == [350, 400, 450, 361]
    with pytest.raises(ValueError):
        make(n=4, workpiece_height=np.array([10.0, -1.0, 30.0, 12.3]))


def test_product_refuses_to_run_without_the_gpu():
    """No CPU fallback: the default backend is the HIP library and needs a CUDA/HIP device."""
    with pytest.raises(RuntimeError, match="no CPU path"):
        WireEDMEnv(num_envs=2, device="cpu")


def test_module_getters_match_the_reference(golden_dir):
    """F12: `IgnitionModule.get_critical_density_for_gap / get_debris_short_probability / get_lambda`
    and `MaterialRemovalModule.get_current_mapping_table` of the reference over a grid."""
    import numpy as np

    from tests._oracle_backend import OracleBackend

    z = np.load(golden_dir / "f12_module_getters.npz")
    gaps, dens = torch.from_numpy(z["gaps"]), torch.from_numpy(z["densities"])
    env = WireEDMEnv(num_envs=len(gaps), device="cpu", backend=OracleBackend)
    env.reset(seed=0)
    assert np.array_equal(env.ignition.get_critical_density_for_gap(gaps).numpy(), z["critical_density"])
    p = env.ignition.get_debris_short_probability(gaps[:, None], dens[None, :]).numpy()
    assert np.allclose(p, z["debris_short_probability"], rtol=1e-14, atol=0)
    env.state.wire_position = 10.0
    env.state.workpiece_position = 10.0 + gaps
    assert np.allclose(env.ignition.get_lambda().numpy(), z["ignition_lambda"], rtol=1e-15, atol=0)
    env.state.is_short_circuit[3] = True
    assert bool(torch.isnan(env.ignition.get_lambda()[3]))
    table = env.material.get_current_mapping_table()
    for row in z["mapping"]:
        e = table[f"I{int(row[0])}"]
        assert e["machine_current"] == row[1] and (e["crater_data"] is not None) == bool(row[2])
        if row[2]:
            cd = e["crater_data"]
            assert (cd["ellipsoid_volume_half"], cd["ellipsoid_volume_std"], cd["depth"]) == tuple(row[3:6])
        else:
            assert "error" in e
    assert env.material.get_crater_data_for_current_mode("I99")["current_mode"] == "I1"
    assert env.wire.compute_zone_mean_temperature().shape == (len(gaps),)
    assert set(env.dielectric.get_debris_statistics()) >= {"debris_volume_mm3", "flow_condition"}


def test_reference_class_names_and_state_utils_are_importable():
    import sparc_amd
    from sparc_amd import IgnitionModule, WireModule, get_gap, is_short_circuited
    from tests._oracle_backend import OracleBackend

    for name in ("EDMState", "EnvironmentConfig", "MaterialDatabase", "WireMaterial", "get_material_db", "WireEDMEnv",
                 "IgnitionModule", "IgnitionModuleParameters", "WireModule", "WireModuleParameters",
                 "MaterialRemovalModule", "MaterialModuleParameters", "DielectricModule", "DielectricModuleParameters",
                 "MechanicsModule", "MechanicsModuleParameters"):          # wedm/__init__.py:22-42
        assert hasattr(sparc_amd, name), name
    env = WireEDMEnv(num_envs=3, device="cpu", backend=OracleBackend)
    env.reset(seed=1)
    assert isinstance(env.ignition, IgnitionModule) and isinstance(env.wire, WireModule)
    assert WireModule(env).n_segments == env.n_segments and env.mechanics.control_mode == "position"
    env.state.wire_position[1] = 60.0
    env.state.spark_state[2] = -1
    assert get_gap(env.state).tolist() == [50.0, 0.0, 50.0]
    assert is_short_circuited(env.state).tolist() == [False, False, True]


def test_readme_usage_snippet_runs(tmp_path):
    """The Python block of README.md, executed against the CPU test seam (small batch, short run)."""
    import re
    from pathlib import Path

    import numpy as np

    from tests._oracle_backend import OracleBackend  # noqa: F401  (used by the exec'd code)

    src = (Path(__file__).resolve().parents[1] / "README.md").read_text()
    code = re.search(r"```python\n(.*?)```", src, re.S).group(1)
    out = tmp_path / "run.npz"
    code = (code.replace('num_envs=65536, device="cuda"', 'num_envs=70, device="cpu", backend=OracleBackend')
            .replace("100_000", "2_500").replace('"filepath": "run.npz"', f'"filepath": "{out}"'))
    exec(code, {"OracleBackend": OracleBackend})
    z = np.load(out)
    assert z["time"].shape == (2500, 64) and z["time"][:, 0].tolist() == list(range(1002, 3502))  # after 1 + 1000 us


def test_foreign_device_action_is_rejected():
    from tests._oracle_backend import OracleBackend

    a = WireEDMEnv(num_envs=4, device="cpu", backend=OracleBackend)
    b = WireEDMEnv(num_envs=9, device="cpu", backend=OracleBackend)
    with pytest.raises(ValueError, match="another environment"):
        b.step(a.make_action())


def test_wire_temperature_proxy_reads_gather_and_assignments_write_through():
    """ABI v4 stores the wire quad-interleaved (`T[seg >> 2][env][seg & 3]`); `state.wire_temperature` keeps the
    reference's `[env, segment]` face: reads gather, every form of assignment writes through to the block the kernels
    step, and the padding cells of the last 16-byte word are never touched."""
    import numpy as np
    import torch

    from tests._oracle_backend import OracleBackend
    from sparc_amd import WireEDMEnv, WireModuleParameters

    env = WireEDMEnv(num_envs=5, device="cpu", backend=OracleBackend, wire_params=WireModuleParameters(segment_len=80.0 / 129.5))
    env.reset(seed=1)
    n = env.n_segments
    assert n == 129 and env.state.T.shape == (33, 64, 4)               # 129 segments -> 33 words, the last with 3 padding cells
    wt = env.state.wire_temperature
    assert wt.shape == (5, n) and len(wt) == 5 and wt.dtype == torch.float32
    assert torch.equal(wt.tensor(), torch.full((5, n), 293.15, dtype=torch.float32))
    wt[2, 100:104] = 400.0                                              # slice assignment
    wt[:, 0] = torch.arange(5, dtype=torch.float32)                     # tensor assignment
    assert env.state.T[25, 2, 0] == 400.0 and env.state.T[25, 2, 3] == 400.0 and env.state.T[26, 2, 0] == 293.15
    assert env.state.T[0, :5, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]
    assert float(wt[2].max()) == 400.0 and float(torch.max(wt)) == 400.0 and float((wt - 1.0)[2, 101]) == 399.0
    assert np.asarray(wt).shape == (5, n) and wt.quads.shape == (5, 33, 4)
    env.state.wire_temperature = np.full(n, 300.0, dtype=np.float32)    # whole-array assignment, broadcast over environments
    assert bool((env.state.wire_temperature == 300.0).all())
    pad = env.state.T[32, :5, 1:]                                       # cells 129..131: padding, still what the reset wrote
    assert bool((pad == np.float32(293.15)).all())
    # chained indexing and in-place methods write through too (a gathered temporary used to swallow them)
    wt = env.state.wire_temperature
    wt[1][10] = 999.0
    assert env.state.T[2, 1, 2] == 999.0
    wt.add_(5.0)
    assert env.state.T[2, 1, 2] == 1004.0 and env.state.T[0, 0, 0] == 305.0
    env.state.wire_temperature[2].fill_(500.0)
    assert bool((env.state.T[:32, 2, :] == 500.0).all()) and env.state.T[32, 2, 0] == 500.0 and env.state.T[0, 3, 0] == 305.0
    env.state.wire_temperature[3, 4:8] += 1.0
    env.state.wire_temperature[4] *= 2.0
    wt[0:2, 20:24].zero_()
    assert env.state.T[1, 3, :].tolist() == [306.0] * 4 and env.state.T[7, 4, 1] == 610.0 and env.state.T[5, 1, 0] == 0.0
    wt[:, 1:].clamp_(max=400.0)
    assert float(torch.max(wt)) == 610.0 and float(wt[4, 1:].max()) == 400.0  # (column 0 was left out of the clamp)
    sub = wt[2:4]                                                       # a proxy of rows 2..3: reads gather at the time of the read
    assert sub.shape == (2, n) and len(sub) == 2 and float(sub[0, 3]) == 400.0
    picked = wt[torch.tensor([0, 4])]                                   # advanced indexing: a plain copy, as on a NumPy array
    assert isinstance(picked, torch.Tensor) and picked.shape == (2, n)
    assert torch.stack([wt[0], wt[1]]).shape == (2, n)
    import time
    t0 = time.perf_counter()
    assert torch.as_tensor(wt[:, :]).shape == (5, n) and len(list(wt)) == 5 and isinstance(next(iter(wt)), torch.Tensor)
    assert time.perf_counter() - t0 < 1.0                              # (the sequence protocol walks one gathered copy)
    assert bool((env.state.T[32, :5, 1:] == np.float32(293.15)).all())  # padding still untouched
    env.state.wire_temperature = np.full(n, 300.0, dtype=np.float32)
    env.step_many(env.make_action(), 3)                                 # and the kernels' view is the same memory
    assert abs(float(env.state.wire_temperature[0, 64]) - 300.0) < 1.0 and float(env.state.wire_temperature[0, 0]) == np.float32(293.15)
