#!/bin/bash
set -o pipefail
OUT=gpurun_out/r2c
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc; tail -15 $OUT/pytest.log
run() {  # workload kernel lanes
    python bench.py --steps 1000 --warmup 100 --substeps 1 --kernel $2 --lanes $3 --workload $1 --no-cpu-baseline > $OUT/1us_$1_k$2_l$3.json 2>$OUT/1us_$1_k$2_l$3.err || tail -3 $OUT/1us_$1_k$2_l$3.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/1us_$1_k$2_l$3.json")); print("$1 k$2 l$3", d["config"]["kernel"], "us/launch %.2f" % (d["roofline"]["kernel_ms"]*1e3), "value %.3e" % d["value"])
except Exception as e: print("$1 k$2 l$3 failed", e)
PY
}
run config3 5 0
for l in 1 2 4 8; do run config3 9 $l; done
run config4 5 0
for l in 4 8 16; do run config4 9 $l; done
run config2 5 0
for l in 4 8 16; do run config2 9 $l; done
