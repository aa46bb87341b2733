#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x -k "dense_sparking or randomized_configurations or config3_grid or full_headline" 2>&1 | tail -4
for rep in 1 2; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench default', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 3), d['config']['kernel'].split('<<<')[0])"
  python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-side --gap 15 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gap15 auto', '%.4e' % d['value'], 'ms', round(d['roofline']['kernel_ms'], 3), d['config']['kernel'].split('<<<')[0])"
  python tools/closed_loop.py voltage 10 config3 100 2>/dev/null | sed "s|^|auto |"
  WEDM_HINT=0 python tools/closed_loop.py voltage 10 config3 100 2>/dev/null | sed "s|^|hint off |"
done
