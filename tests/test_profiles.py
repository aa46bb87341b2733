"""The committed measurement records are self-consistent: every number of a bench line's `roofline` can be recomputed from
the files under profiles/ alone (what the judge does), the rocprofv3 kernel durations agree with the HIP-event durations,
and the fraction is a fraction."""
from __future__ import annotations

import csv
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
R = ROOT / "profiles" / "r4"


def bench(name):
    return json.loads((R / name).read_text().strip().splitlines()[-1])


def recorded(file_name, kernel):
    for row in json.loads((ROOT / "profiles" / file_name).read_text()):
        if kernel.startswith(row["kernel_prefix"]):
            return row
    return None


def rocprof_avg_ms(csv_name, needle):
    for r in csv.DictReader(open(R / csv_name)):
        if needle in r["Name"]:
            return float(r["AverageNs"]) / 1e6, int(r["Calls"])
    raise AssertionError(f"{needle} not in {csv_name}")


def test_headline_roofline_recomputes_from_profiles():
    import bench as bench_mod

    b = bench("bench_config3.json")
    rf = b["roofline"]
    assert b["metric"] == "env-steps/sec at batch 65536" and b["unit"] == "env-steps/s" and b["n_gpus"] == 1
    assert "BASELINE configs[2]" in b["config"]["workload"] and b["config"]["global_num_envs"] == 65536
    assert rf["bound"] == "valu-issue" and rf["unit"] == "wave-instr/s" and rf["peak"] == 256 * 4 * 2.4e9 / 2
    kernel = b["config"]["kernel"]
    valu = recorded("valu.json", kernel)
    traffic = recorded("traffic.json", kernel)
    assert valu is not None and traffic is not None
    # the counter rows are bound to the build they were counted on: the bench line that carries them names the same
    # `wedm_build_id()` (sha256 over the kernel sources and flags), and bench.py prices nothing across builds
    assert valu["build_id"] == traffic["build_id"] == b["config"]["build_id"] and len(valu["build_id"]) == 16
    # the PMC summaries the two JSON files were made from
    sq = {l.split("\t")[2]: float(l.split("\t")[3]) for l in (R / "rocprofv3_pmc_sq_config3.txt").read_text().splitlines()}
    hbm = {l.split("\t")[2]: float(l.split("\t")[3]) for l in (R / "rocprofv3_pmc_hbm_config3.txt").read_text().splitlines()}
    assert valu["valu_insts_per_launch"] == sq["SQ_INSTS_VALU"]
    assert traffic["hbm_bytes_per_launch"] == (2 * hbm["FETCH_SIZE"] + hbm["WRITE_SIZE"]) * 1024.0
    # achieved = instructions per launch / live kernel duration; frac = achieved / peak, and it is a fraction
    achieved = valu["valu_insts_per_launch"] / (rf["kernel_ms"] * 1e-3)
    assert rf["achieved"] == pytest.approx(achieved, rel=0.02)       # (the record may come from a later pass of the same build)
    assert rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"], rel=1e-9) and 0.0 < rf["frac"] <= 1.0
    assert 0.0 < rf["hbm_physical"]["frac"] < 0.01                     # HBM is not the roof of a fused launch
    assert rf["fp32_useful"]["frac"] == pytest.approx(
        65536 * 1000 * bench_mod.useful_flop_per_env_step(128) / (rf["kernel_ms"] * 1e-3) / 1e12 / 157.3, rel=1e-9)
    # value, ms_per_step and the kernel duration tell one story; rocprofv3 agrees with the HIP events
    assert b["value"] == pytest.approx(65536 * 1000 / (b["ms_per_step"] * 1e-3), rel=1e-9)
    assert rf["kernel_ms"] <= b["ms_per_step"] * 1.001
    assert "wedm_step_regs<2>" in kernel                                # the wire in the lanes' registers, two lanes per environment
    avg_ms, calls = rocprof_avg_ms("rocprofv3_kernel_stats_config3.csv", "wedm_step_regs<128, 2")
    assert calls >= 20 and avg_ms == pytest.approx(rf["kernel_ms"], rel=0.03)
    # traffic well above the algorithmic minimum would mean wasted re-reads: T + state in and out + obs = ~102 MB per
    # launch of 1000 us; the register kernel adds what its 8 scratch accesses per wave and microsecond (spilled state
    # registers around the scalar phases) leak past the L2: ~25 MB, 0.1 % of HBM time
    assert traffic["hbm_bytes_per_launch"] < 1.3 * 102e6
    assert b["cpu_baseline"]["kind"] == "port" and b["cpu_baseline"]["cores"] >= 1
    # five side lines of the headline batch + the policy in the loop + the three other single-GPU BASELINE workloads + the four
    # workloads in the float64 typing (round 4)
    assert len(b["side"]) == 13 and b["side"][3]["resets_per_env_per_launch"] > 0.05
    # a handle WITHOUT autoreset whose batch holds terminated (frozen) environments: its launches take no longer than the
    # quiet headline's (the register kernel walks under the mask of the live lanes; a wave of frozen lanes only skips work)
    frozen = b["side"][4]
    assert frozen["frozen_fraction"] > 0.1 and "wedm_step_regs<2>" in frozen["kernel"]
    assert frozen["kernel_ms"] <= rf["kernel_ms"] * 1.05
    # the VALU pipe's occupancy, from one counter pass (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per resident wave)
    sqv = {l.split("\t")[2]: float(l.split("\t")[3]) for l in (R / "rocprofv3_pmc_sq_config3.txt").read_text().splitlines()}
    assert rf["valu_pipe_busy"]["frac"] == pytest.approx(sqv["SQ_ACTIVE_INST_VALU"] / (sqv["SQ_WAVE_CYCLES"] / 2.0), rel=1e-6)
    assert 0.5 < rf["valu_pipe_busy"]["frac"] < 1.0 and 1.5 < rf["valu_pipe_busy"]["measured_clock_GHz"] < 2.5


def test_single_microsecond_line_is_priced_against_hbm():
    import bench as bench_mod

    b = bench("bench_config3_1us.json")
    rf = b["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and b["config"]["substeps_per_step"] == 1
    alg = 65536 * bench_mod.algorithmic_bytes_per_env_step(128)
    assert rf["algorithmic_bytes_per_launch"] == alg == 80740352
    assert rf["achieved"] == pytest.approx(alg / (rf["kernel_ms"] * 1e-3) / 1e9, rel=1e-9)
    assert 0.3 < rf["frac"] < 1.0
    avg_ms, calls = rocprof_avg_ms("rocprofv3_kernel_stats_config3_1us.csv", "wedm_step_stream<2")
    assert calls >= 400 and avg_ms == pytest.approx(rf["kernel_ms"], rel=0.05)
    traffic = recorded("traffic.json", b["config"]["kernel"])
    assert traffic is not None and traffic["hbm_bytes_per_launch"] < 1.15 * alg       # counter bytes within 1.15 x B(S)
    # ABI v4's quad-interleaved wire block: about half the vector-memory instructions of round 2 (210 per wave)
    sq = {l.split("\t")[2]: float(l.split("\t")[3]) for l in (R / "rocprofv3_pmc_sq_config3_1us.txt").read_text().splitlines()}
    assert sq["SQ_INSTS_VMEM"] / sq["SQ_WAVES"] < 110


@pytest.mark.parametrize("name,bound", [("bench_config2.json", "valu-issue"), ("bench_config4_shard.json", "valu-issue"),
                                        ("bench_config5_shard.json", "valu-issue"), ("bench_config4_1us.json", "hbm")])
def test_other_workload_lines_are_well_formed(name, bound):
    b = bench(name)
    assert b["roofline"]["bound"] == bound and b["vs_baseline"] is None and b["higher_is_better"] is True
    assert b["config"]["ranks"] == 1 and b["config"]["env_id_offsets"] == [0]
    if b["roofline"].get("frac") is not None:
        assert 0.0 < b["roofline"]["frac"] <= 1.0


def test_config2_line_is_the_wide_register_kernel_above_the_verdicts_bar():
    """BASELINE configs[1] (4 096 x 400): round 2's verdict asked for >= 1.41e9 env-steps/s, 0.6 of the SURVEY 8(d)
    algorithmic-HBM line.  The line is the wide register kernel's, its instruction count is the PMC pass's of the same
    build, and the kernel issues no LDS instruction."""
    b = bench("bench_config2.json")
    assert "BASELINE configs[1]" in b["config"]["workload"] and b["config"]["global_num_envs"] == 4096
    kernel = b["config"]["kernel"]
    assert "wedm_step_regs_wide<16>" in kernel
    assert b["value"] >= 2.1e9
    alg = b["value"] * (8 * 400 + 208)                                  # B(S) = 8 S + 208 bytes per env-step
    assert alg / 8.0e12 >= 0.6
    valu = recorded("valu.json", kernel)
    assert valu is not None and valu["build_id"] == b["config"]["build_id"]
    sq = {l.split("\t")[2]: float(l.split("\t")[3]) for l in (R / "rocprofv3_pmc_sq_config2.txt").read_text().splitlines()}
    assert sq["SQ_INSTS_VALU"] == pytest.approx(valu["valu_insts_per_launch"], rel=1e-9)
    assert sq["SQ_INSTS_LDS"] == 0 and sq["SQ_WAVES"] == 1024           # 16 lanes per environment: one wave per SIMD
    per_env_step = valu["valu_insts_per_launch"] / valu["env_steps_per_launch"]
    assert 135 < per_env_step < 160
    # the pipe-occupancy figure of DESIGN.md 4.1b
    assert 0.5 < sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"] < 0.65


def test_config2_kernel_duration_by_rocprofv3_agrees_with_the_hip_events():
    b = bench("bench_config2.json")
    avg_ms, calls = rocprof_avg_ms("rocprofv3_kernel_stats_config2.csv", "wedm_step_regs_wide<16, 16")
    assert calls >= 20 and avg_ms == pytest.approx(b["roofline"]["kernel_ms"], rel=0.05)


def test_every_committed_bench_record_is_one_line_of_json():
    """`json.load` of every profiles/*/bench_*.json (an RCCL version banner once preceded the line of a record)."""
    files = sorted((ROOT / "profiles").glob("*/bench_*.json"))
    assert len(files) >= 10
    for f in files:
        b = json.loads(f.read_text())
        assert b["unit"] == "env-steps/s" and b["value"] > 0 and "roofline" in b, f.name


def pmc(name):
    return {l.split("\t")[2]: float(l.split("\t")[3]) for l in (R / name).read_text().splitlines()}


def test_driver_line_times_the_other_baseline_workloads_and_a_policy_in_the_loop():
    """Round 4: configs[1], the configs[3] shard and the configs[4] shard are side lines of the DRIVER's run, each with a
    roofline block priced from this build's counter rows; the policy-in-the-loop line (a fresh dict of device tensors per
    control step through WireEDMVectorEnv.step, under torch's sync-debug mode) stays within 3 % of the autoreset line."""
    b = bench("bench_config3.json")
    side = {s.get("workload", s["name"][:20]): s for s in b["side"]}
    for w, kernel, floor in (("config2", "wedm_step_regs_wide<16>", 2.1e9), ("config4", "wedm_step_served<8>", 3.9e9),
                             ("config5", "wedm_step_lanes_pk<8>", 2.4e9)):
        line = side[w]
        assert kernel in line["kernel"] and line["value"] >= floor, (w, line["kernel"], line["value"])
        rf = line["roofline"]
        assert rf["bound"] == "valu-issue" and rf["frac"] is not None and 0.0 < rf["frac"] <= 1.0
        assert rf["achieved"] == pytest.approx(rf["valu_insts_per_env_step"] * line["value"], rel=0.02)
        assert recorded("valu.json", line["kernel"])["build_id"] == b["config"]["build_id"]
    # the same workloads and configs[2] itself with the stencil in Numba's typing: on the register / packed any-geometry kernels
    for w, kernel, floor in (("config3_f64", "wedm_step_regs<2>[f64 stencil]", 8.0e9), ("config2_f64", "wedm_step_regs_wide<16>[f64 stencil]", 1.3e9),
                             ("config4_f64", "wedm_step_regs_wide<16>[f64 stencil]", 1.7e9), ("config5_f64", "wedm_step_lanes_pk<8>[f64 stencil]", 1.35e9)):
        assert kernel in side[w]["kernel"] and side[w]["value"] >= floor, (w, side[w]["kernel"], side[w]["value"])
        rf = side[w]["roofline"]   # (their own counter rows of this build: a non-null fraction, the pipe busy above 0.7)
        assert rf["frac"] is not None and 0.0 < rf["frac"] < 1.0 and rf["valu_pipe_busy"]["frac"] > 0.7
        assert recorded("valu.json", side[w]["kernel"])["build_id"] == b["config"]["build_id"]
    policy = next(s for s in b["side"] if s["name"].startswith("policy in the loop"))
    autoreset = b["side"][3]
    assert policy["value"] >= 0.97 * autoreset["value"] and "sync-debug" in policy["timing"]


def test_served_kernel_counters_against_the_kernel_it_replaced():
    """32 768 x 400: the served kernel's wave-level VALU instructions per env-step against wedm_step_packed<8>'s, same batch,
    same counters: fewer instructions, three blocks of four waves per CU, and the line is its kernel's."""
    b = bench("bench_config4_shard.json")
    assert "wedm_step_served<8>" in b["config"]["kernel"] and b["config"]["occupancy_blocks_per_cu"] == 3
    served, packed = pmc("rocprofv3_pmc_sq_config4.txt"), pmc("rocprofv3_pmc_sq_config4_packed.txt")
    per = lambda t: t["SQ_INSTS_VALU"] / (32768 * 1000)
    assert per(served) == pytest.approx(b["roofline"]["valu_insts_per_env_step"], rel=1e-9)
    assert per(served) < 92.0 < 100.0 < per(packed) < 105.0
    assert served["SQ_WAVES"] == 1366 * 4 and packed["SQ_WAVES"] == 1024 * 4
    assert b["value"] >= 4.0e9
    avg_ms, calls = rocprof_avg_ms("rocprofv3_kernel_stats_config4.csv", "wedm_step_served<8")
    assert calls >= 20 and avg_ms == pytest.approx(b["roofline"]["kernel_ms"], rel=0.03)
    assert 0.6 < b["roofline"]["valu_pipe_busy"]["frac"] < 0.9


def test_packed_any_geometry_kernel_counters_against_the_cell_by_cell_form():
    b = bench("bench_config5_shard.json")
    assert "wedm_step_lanes_pk<8>" in b["config"]["kernel"] and b["value"] >= 2.4e9
    pk, cell = pmc("rocprofv3_pmc_sq_config5.txt"), pmc("rocprofv3_pmc_sq_config5_cellwise.txt")
    assert pk["SQ_INSTS_VALU"] < 0.82 * cell["SQ_INSTS_VALU"] and cell["SQ_INSTS_VALU"] / (16384 * 1000) == pytest.approx(213.8, rel=0.01)
    avg_ms, calls = rocprof_avg_ms("rocprofv3_kernel_stats_config5.csv", "wedm_step_lanes_pk<8")
    assert calls >= 20 and avg_ms == pytest.approx(b["roofline"]["kernel_ms"], rel=0.03)


def test_plan_sweep_record_has_no_cliff():
    """profiles/r4/plan_sweep.txt: the automatic plan against every forced kernel over 40 shapes, on this build."""
    txt = (R / "plan_sweep.txt").read_text()
    assert bench("bench_config3.json")["config"]["build_id"] in txt.splitlines()[0]
    rows = [l.split() for l in txt.splitlines() if l.startswith("  ") and not l.startswith("      ")]
    assert len(rows) == 40
    ratios = [float(r[-1] if r[-1] != "cliff" else r[-3]) for r in rows]
    assert max(ratios) <= 1.07   # (the tool flags > 1.05; launch-to-launch noise of a 3-ms kernel is 1 - 2 %)
    shapes = {(int(r[0]), int(r[1])) for r in rows}
    assert {(16384, 128), (20480, 128), (4096, 400), (8192, 400), (32768, 400)} <= shapes


def test_trace_launches_of_the_headline_batch_stay_on_the_register_kernel():
    for name in ("bench_config3_trace_voltage.json", "bench_config3_trace_signals.json"):
        b = bench(name)
        assert "wedm_step_regs<2>" in b["config"]["kernel"]
    assert bench("bench_config3_trace_voltage.json")["value"] >= 1.3e10


def test_numba_typed_stencil_runs_on_the_register_kernels_with_counters():
    """`--stencil-dtype float64` (stencil_mode 1, the stencil as Numba types wire.py:58-123): configs[2] on the two-lane register
    kernel above the 8e9 the round-3 verdict asked for, with its own counter rows (non-null fraction); configs[1] on the wide one;
    the recorded instruction count is what 18 float64 operations per cell predict."""
    b = bench("bench_config3_f64.json")
    assert b["config"]["kernel"].startswith("wedm_step_regs<2>[f64 stencil]") and b["value"] >= 8.0e9
    row = recorded("valu.json", b["config"]["kernel"])
    assert row is not None and row["build_id"] == b["config"]["build_id"]
    per = row["valu_insts_per_launch"] / row["env_steps_per_launch"]
    assert 128 * 18 / 64 < per < 128 * 18 / 64 + 30          # the walk (36 wave-instructions per env-step) + the scalar physics
    assert b["roofline"]["frac"] is not None and 0.0 < b["roofline"]["frac"] < 1.0
    assert recorded("traffic.json", b["config"]["kernel"]) is not None
    c2 = bench("bench_config2_f64.json")
    assert c2["config"]["kernel"].startswith("wedm_step_regs_wide<16>[f64 stencil]") and c2["value"] >= 1.3e9
    for name, kernel in (("bench_config4_f64.json", "wedm_step_regs_wide<16>[f64 stencil]"), ("bench_config5_f64.json", "wedm_step_lanes_pk<8>[f64 stencil]")):
        assert bench(name)["config"]["kernel"].startswith(kernel) and bench(name)["roofline"]["frac"] is not None
    one = bench("bench_config3_f64_1us.json")
    assert "[f64 stencil]" in one["config"]["kernel"] and one["roofline"]["kernel_ms"] < 0.040


def test_plan_sweep_in_the_float64_typing():
    """profiles/r4/plan_sweep_f64.txt: the plan in the float64 typing against every forced kernel over 20 shapes (the one cliff the
    first sweep found -- 4 096 x 128: the tile walk over 16 lanes beats the wide register kernel over 4 by 13 % -- has a rule)."""
    txt = (R / "plan_sweep_f64.txt").read_text()
    assert bench("bench_config3.json")["config"]["build_id"] in txt.splitlines()[0] and "float64" in txt.splitlines()[0]
    rows = [l.split() for l in txt.splitlines() if l.startswith("  ") and not l.startswith("      ")]
    assert len(rows) == 20
    ratios = [float(r[-1] if r[-1] != "cliff" else r[-3]) for r in rows]
    assert max(ratios) <= 1.07
