#!/bin/bash
# tools/ab_config2_1us.sh [lib ...]: single microseconds at 4 096 x 400 with 8 and 16 lanes per environment, in-tree library first
OUT=gpurun_out/ab_c2; mkdir -p $OUT
for lib in sparc_amd/libwedm_hip.so "$@"; do
  tag=$(basename $lib .so)
  for L in 8 16; do
    WEDM_HIP_LIB=$lib python bench.py --steps 1000 --warmup 100 --substeps 1 --kernel 6 --lanes $L --workload config2 --no-cpu-baseline --no-side > $OUT/$tag.l$L.json 2>$OUT/$tag.l$L.err || tail -3 $OUT/$tag.l$L.err
    python -c "
import json; d=json.load(open('$OUT/$tag.l$L.json')); print('$tag lanes $L', d['config']['kernel'], 'us/launch %.2f' % (d['roofline']['kernel_ms']*1e3))"
  done
done
