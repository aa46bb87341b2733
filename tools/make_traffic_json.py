#!/usr/bin/env python
"""profiles/traffic.json and profiles/valu.json from the PMC summaries of tools/profile_round.sh.

    python tools/make_traffic_json.py profiles/r2
    python tools/make_traffic_json.py gpurun_out/r2/summary profiles/r2   (second argument: where the summaries will be committed)

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes).  FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-B requests at 64 B);
that calibration is for 16-B-per-lane streams, ours are 4-8 B per lane, so the doubled figure is
an upper estimate.  VALU instructions per launch = SQ_INSTS_VALU (wave-level instructions, summed
over the launch's waves) from its own pass.  bench.py reports the row whose kernel_prefix matches
the kernel that ran, and only when the row's build_id (wedm_build_id() of the library, read from the bench line) is the
loaded library's."""
import json
import sys
from pathlib import Path

d = Path(sys.argv[1])
label = sys.argv[2] if len(sys.argv) > 2 else str(d)
traffic, valu = [], []
CASES = (("config3", "bench_config3.json"), ("config4", "bench_config4_shard.json"), ("config2", "bench_config2.json"),
         ("config5", "bench_config5_shard.json"), ("config3_1us", "bench_config3_1us.json"), ("config3_f64", "bench_config3_f64.json"),
         ("config2_f64", "bench_config2_f64.json"), ("config4_f64", "bench_config4_f64.json"), ("config5_f64", "bench_config5_f64.json"))


def table(path):
    vals = {}
    for line in path.read_text().splitlines():
        parts = line.split("\t")
        if len(parts) >= 4:
            vals[parts[2]] = float(parts[3])
    return vals


for cfg, bench in CASES:
    if not (d / bench).exists():
        continue
    b = json.loads((d / bench).read_text().strip().splitlines()[-1])
    env_steps = b["config"]["global_num_envs"] * b["config"]["substeps_per_step"]
    build_id = b["config"].get("build_id")  # wedm_build_id() of the library the counters were taken on
    f = d / f"rocprofv3_pmc_hbm_{cfg}.txt"
    if f.exists():
        vals = table(f)
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            traffic.append({
                "kernel_prefix": b["config"]["kernel"], "workload": b["config"]["workload"], "build_id": build_id,
                "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
                "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
                "env_steps_per_launch": env_steps,
                "source": f"{label}/rocprofv3_pmc_hbm_{cfg}.txt (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)",
            })
    f = d / f"rocprofv3_pmc_sq_{cfg}.txt"
    if f.exists():
        vals = table(f)
        if "SQ_INSTS_VALU" in vals:
            valu.append({
                "kernel_prefix": b["config"]["kernel"], "workload": b["config"]["workload"], "build_id": build_id,
                "valu_insts_per_launch": vals["SQ_INSTS_VALU"], "env_steps_per_launch": env_steps,
                **{k.lower(): v for k, v in vals.items() if k != "SQ_INSTS_VALU"},
                "source": f"{label}/rocprofv3_pmc_sq_{cfg}.txt (separate rocprofv3 --pmc SQ_* pass of bench.py)",
            })
(Path("profiles") / "traffic.json").write_text(json.dumps(traffic, indent=1) + "\n")
(Path("profiles") / "valu.json").write_text(json.dumps(valu, indent=1) + "\n")
for r in traffic:
    print("traffic", r["kernel_prefix"], f'{r["hbm_bytes_per_launch"] / 1e6:.1f} MB per launch')
for r in valu:
    print("valu   ", r["kernel_prefix"], f'{r["valu_insts_per_launch"] / r["env_steps_per_launch"]:.2f} wave-VALU instructions per env-step')
