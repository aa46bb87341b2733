#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "default_config_fused or ragged or per_environment or randomized_configurations or edge_scenarios" 2>&1 | tail -3
for rep in 1 2; do
for lib in "" build/ablate/libwedm_FPLAIN.so; do
  for extra in "--workload config4" "--workload config4 --gap 15" "--workload config5" "--workload config2"; do
  WEDM_HIP_LIB=${lib:+$PWD/$lib} python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-side $extra 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$extra', '${lib:-in-tree(dense)}', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 3), d['config']['kernel'].split('<<<')[0])"
  done
done
done
