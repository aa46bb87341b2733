#!/bin/bash
# tools/ab_config2.sh [lib ...]: BASELINE configs[1] (4 096 x 400, fused launches), in-tree library first, then every library
# named, then the in-tree library again (A/B on ONE box)
OUT=gpurun_out/ab_config2; mkdir -p $OUT
run() {
    local tag=$(basename $1 .so)
    WEDM_HIP_LIB=$1 python bench.py --steps 20 --warmup 3 --workload config2 --no-cpu-baseline --no-side > $OUT/$tag.json 2>$OUT/$tag.err || { tail -3 $OUT/$tag.err; return 1; }
    python -c "
import json; d=json.load(open('$OUT/$tag.json')); print('$tag', d['config']['kernel'], 'env-steps/s %.4g' % d['value'], 'ms/launch %.3f' % d['roofline']['kernel_ms'])"
}
for lib in sparc_amd/libwedm_hip.so "$@" sparc_amd/libwedm_hip.so; do run $lib || exit 1; done
