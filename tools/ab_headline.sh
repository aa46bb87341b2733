#!/bin/bash
# tools/ab_headline.sh "<bench flags>" [lib ...]: the headline workload (65 536 x 128, 1000 us per launch) with extra bench flags
# (e.g. "--kernel 4 --lanes 1"), in-tree library first, then every library named (A/B on ONE box)
FLAGS=$1; shift
OUT=gpurun_out/ab_headline; mkdir -p $OUT
run() {
    local tag=$(basename $1 .so)
    WEDM_HIP_LIB=$1 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side $FLAGS > $OUT/$tag.json 2>$OUT/$tag.err || { tail -3 $OUT/$tag.err; return 1; }
    python -c "
import json; d=json.load(open('$OUT/$tag.json')); print('$tag [$FLAGS]', d['config']['kernel'], 'env-steps/s %.4g' % d['value'], 'ms/launch %.3f' % d['roofline']['kernel_ms'])"
}
for lib in sparc_amd/libwedm_hip.so "$@"; do run $lib || exit 1; done
