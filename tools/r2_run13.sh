#!/bin/bash
export TMPDIR=/tmp
for rep in 1 2; do
  for h in plain dense; do
    python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-side --workload-hint $h 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench default $h', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 3), d['config']['kernel'].split('<<<')[0])"
    python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-side --gap 15 --workload-hint $h 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gap15 $h', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 3), d['config']['kernel'].split('<<<')[0])"
  done
  WEDM_HINT=1 python tools/closed_loop.py voltage 10 config3 100 2>/dev/null | sed "s|^|hint dense |"
  WEDM_HINT=0 python tools/closed_loop.py voltage 10 config3 100 2>/dev/null | sed "s|^|hint plain |"
done
