#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "default_config_fused or config3_grid or ragged or tiny_wires or randomized_configurations" 2>&1 | tail -3
for rep in 1 2; do
  for extra in "" "--workload config4" "--workload config2" "--substeps 1 --steps 2000 --warmup 200" "--substeps 1 --steps 2000 --warmup 200 --workload config2"; do
  python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side $extra 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$extra', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 4), d['config']['kernel'].split('<<<')[0])"
  done
done
