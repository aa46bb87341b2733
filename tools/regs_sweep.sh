#!/bin/bash
# tools/regs_sweep.sh: 128-segment batches of 8 192 ... 131 072 environments, fused launches of 1000 us: the automatic kernel
# choice against the register kernel forced (--kernel 7), then the full bench line with its side measurements on kernel 7
OUT=gpurun_out/regs_sweep; mkdir -p $OUT
for N in 8192 16384 32768 49152 65536 131072; do
 for K in 0 7; do
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side --num-envs $N --kernel $K > $OUT/n$N.k$K.json 2>$OUT/err.txt || tail -3 $OUT/err.txt
  python -c "
import json; d=json.load(open('$OUT/n$N.k$K.json')); print($N, d['config']['kernel'], 'env-steps/s %.4g' % d['value'], 'ms %.3f' % d['roofline']['kernel_ms'])"
 done
done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --kernel 7 > $OUT/full_k7.json 2>$OUT/err.txt || tail -3 $OUT/err.txt
python -c "
import json; d=json.load(open('$OUT/full_k7.json')); print(d['value'], d['config']['kernel'])
for s in d.get('side',[]): print('  ', s['name'][:60], '%.4g' % s['value'], s.get('kernel', s.get('config',{}).get('kernel','')))"
