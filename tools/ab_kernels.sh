#!/bin/bash
# tools/ab_kernels.sh <workload> <lib or ""> "<kernel>:<lanes> ..."   -> one line per kernel choice
WL="$1"; LIB="$2"; shift 2
for kl in $1; do
  k=${kl%%:*}; l=${kl##*:}
  WEDM_HIP_LIB=${LIB:+$PWD/$LIB} python bench.py --steps 8 --warmup 2 --workload $WL --kernel $k --lanes $l --no-cpu-baseline 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$WL', '${LIB:-in-tree}', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 3), d['config']['kernel'].split(' n_sub')[0])"
done
