#!/bin/bash
# A/B timing of alternative builds of libwedm_hip.so on the GPU box:
#   tools/ab_bench.sh "<workload> ..." lib1.so lib2.so ...   ("" = the in-tree library)
# Each (workload, library) pair is run twice, alternating, to expose run-to-run noise.
WL="$1"; shift
for rep in 1 2; do
  for w in $WL; do
    for lib in "$@"; do
      WEDM_HIP_LIB=${lib:+$PWD/$lib} python bench.py --steps 10 --warmup 2 --workload $w --no-cpu-baseline 2>/dev/null |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', '${lib:-in-tree}', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 3), d['config']['kernel'].split('<<<')[0])"
    done
  done
done
