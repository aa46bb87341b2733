#!/bin/bash
# round 2, GPU call 1: parity suite, then the single-microsecond kernel variants side by side, then the headline
set -o pipefail
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee $OUT/pytest.rc; tail -5 $OUT/pytest.log
for w in config3 config4; do
  for k in 5 6 7 8; do
    python bench.py --steps 2000 --warmup 200 --substeps 1 --kernel $k --workload $w --no-cpu-baseline > $OUT/1us_${w}_k$k.json 2>$OUT/1us_${w}_k$k.err
    python - <<PY
import json
d=json.load(open("$OUT/1us_${w}_k$k.json")); print("$w k$k", d["config"]["kernel"], "us/launch %.2f" % (d["roofline"]["kernel_ms"]*1e3), "value %.3e" % d["value"])
PY
  done
done
python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_config3.json 2>$OUT/bench_config3.err
python bench.py --steps 10 --warmup 3 --gap 15 --no-cpu-baseline > $OUT/bench_config3_gap15.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --workload config2 --no-cpu-baseline > $OUT/bench_config2.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --workload config4 --no-cpu-baseline > $OUT/bench_config4.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --workload config5 --no-cpu-baseline > $OUT/bench_config5.json 2>/dev/null
for f in bench_config3 bench_config3_gap15 bench_config2 bench_config4 bench_config5; do
python - <<PY
import json
d=json.load(open("$OUT/$f.json")); print("$f", d["config"]["kernel"], "ms %.3f" % d["roofline"]["kernel_ms"], "value %.4e" % d["value"])
PY
done
