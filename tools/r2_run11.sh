#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
for lib in "" build/ablate/libwedm_NOEARLY.so; do
  for extra in "" "--gap 15" "--workload config4"; do
  WEDM_HIP_LIB=${lib:+$PWD/$lib} python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side $extra 2>/dev/null |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$extra', '${lib:-in-tree}', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 3), d['config']['kernel'].split('<<<')[0])"
  done
  WEDM_HIP_LIB=${lib:+$PWD/$lib} python tools/closed_loop.py voltage 10 config3 100 2>/dev/null | sed "s|^|${lib:-in-tree} |"
done
done
