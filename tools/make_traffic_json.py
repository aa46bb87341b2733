#!/usr/bin/env python
"""profiles/traffic.json from the PMC summaries of tools/profile_round.sh.

    python tools/make_traffic_json.py profiles/r1

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes).  FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-B requests at 64 B);
that calibration is for 16-B-per-lane streams, ours are 4-8 B per lane, so the doubled figure is
an upper estimate.  bench.py reports the row whose kernel_prefix matches the kernel that ran."""
import json
import sys
from pathlib import Path

d = Path(sys.argv[1])
rows = []
for cfg, bench in (("config3", "bench_config3.json"), ("config4", "bench_config4_shard.json"), ("config2", "bench_config2.json")):
    f = d / f"rocprofv3_pmc_hbm_{cfg}.txt"
    if not f.exists():
        continue
    vals = {}
    for line in f.read_text().splitlines():
        parts = line.split("\t")
        vals[parts[2]] = float(parts[3])
    b = json.loads((d / bench).read_text().strip().splitlines()[-1])
    rows.append({
        "kernel_prefix": b["config"]["kernel"],
        "workload": b["config"]["workload"],
        "FETCH_SIZE_KB": vals["FETCH_SIZE"], "WRITE_SIZE_KB": vals["WRITE_SIZE"],
        "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0,
        "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
        "source": f"{d}/rocprofv3_pmc_hbm_{cfg}.txt (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)",
    })
(Path("profiles") / "traffic.json").write_text(json.dumps(rows, indent=1) + "\n")
for r in rows:
    print(r["kernel_prefix"], f'{r["hbm_bytes_per_launch"] / 1e6:.1f} MB per launch vs algorithmic {r["algorithmic_bytes_per_launch"] / 1e9:.1f} GB')
