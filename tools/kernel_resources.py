#!/usr/bin/env python
"""Per-kernel register / scratch / LDS usage of the gfx950 build (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py [extra hipcc flags...]
"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402

cmd = ["hipcc", *g.HIPCC_FLAGS, "-Rpass-analysis=kernel-resource-usage", *sys.argv[1:],
       "-o", "/tmp/_wedm_resources.so", str(ROOT / "sparc_amd/csrc/wedm_kernels.hip")]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    body = m.group(1).strip()
    if body.startswith("Function Name:"):
        cur = {"name": body.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in body:
        k, v = body.split(":", 1)
        cur[k.strip()] = v.strip()
keys = ["TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "SGPRs Spill", "VGPRs Spill", "Occupancy [waves/SIMD]",
        "LDS Size [bytes/block]"]
print(f"{'kernel':44s} " + " ".join(f"{k.split(' ')[0][:10]:>10s}" for k in keys))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
    print(f"{name[:44]:44s} " + " ".join(f"{r.get(k, '-'):>10s}" for k in keys))
