#!/bin/bash
# tools/ab_gap.sh "<gap> ..." "<workload> ..." lib1 lib2 ...   ("" = in-tree): dense-spark workloads
GAPS="$1"; WL="$2"; shift 2
for rep in 1 2; do for g in $GAPS; do for w in $WL; do for lib in "$@"; do
  WEDM_HIP_LIB=${lib:+$PWD/$lib} python bench.py --steps 8 --warmup 2 --gap $g --workload $w --no-cpu-baseline 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gap $g $w', '${lib:-in-tree}', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 3))"
done; done; done; done
