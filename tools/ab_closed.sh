#!/bin/bash
# tools/ab_closed.sh [lib ...]: the bench with its side lines (dense sparking, closed loop, autoreset, frozen), in-tree library first
OUT=gpurun_out/ab_closed; mkdir -p $OUT
for lib in sparc_amd/libwedm_hip.so "$@"; do
  tag=$(basename $lib .so)
  WEDM_HIP_LIB=$lib python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/$tag.json 2>$OUT/$tag.err || tail -3 $OUT/$tag.err
  python -c "
import json; d=json.load(open('$OUT/$tag.json')); print('$tag', '%.4g' % d['value'], d['config']['kernel'], 'ms %.3f' % d['roofline']['kernel_ms'])
for s in d['side']: print('  ', s['name'][:50], '%.4g' % s['value'], s.get('kernel'))"
done
