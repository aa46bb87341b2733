#!/bin/bash
set -o pipefail
OUT=gpurun_out/r2b
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc; tail -15 $OUT/pytest.log
for w in config3; do
  for kl in "3 2" "3 4" "3 8" "3 16" "4 1" "4 2" "4 4" "4 8" "2 4" "2 8" "1 0"; do
    set -- $kl
    python bench.py --steps 1000 --warmup 100 --substeps 1 --kernel $1 --lanes $2 --workload $w --no-cpu-baseline > $OUT/1us_${w}_k$1_l$2.json 2>$OUT/1us_${w}_k$1_l$2.err
    python - <<PY
import json
d=json.load(open("$OUT/1us_${w}_k$1_l$2.json")); print("$w k$1 l$2", d["config"]["kernel"], "us/launch %.2f" % (d["roofline"]["kernel_ms"]*1e3), "value %.3e" % d["value"])
PY
  done
done
