#!/usr/bin/env python
"""One readable line of a bench.py JSON line: `python tools/bench_line.py TAG FILE` (used by the tools/ab_*.sh scripts)."""
import json
import sys

tag, path = sys.argv[1], sys.argv[2]
b = json.loads(open(path).read().strip().splitlines()[-1])
print(f"{tag:38s} {b['value']:.4e} env-steps/s  {b['roofline']['kernel_ms']:.3f} ms  {b['config']['kernel']}  sparks {b['check']['sparks']}")
