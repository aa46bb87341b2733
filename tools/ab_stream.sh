#!/bin/bash
# tools/ab_stream.sh [lib ...]: single-microsecond launches of the stream kernel (variant 6), in-tree library first,
# then every library named (A/B on ONE box), then the phase stamps if build/ablate/libwedm_STAMPS.so exists
OUT=gpurun_out/ab_stream; mkdir -p $OUT
run() {  # lib workload lanes
    local tag=$(basename $1 .so)
    WEDM_HIP_LIB=$1 python bench.py --steps 1000 --warmup 100 --substeps 1 --kernel 6 --lanes $3 --workload $2 --no-cpu-baseline --no-side > $OUT/$tag.$2.json 2>$OUT/$tag.err || { tail -3 $OUT/$tag.err; return 1; }
    python -c "
import json; d=json.load(open('$OUT/$tag.$2.json')); print('$tag $2', d['config']['kernel'], 'us/launch %.2f' % (d['roofline']['kernel_ms']*1e3))"
}
for lib in sparc_amd/libwedm_hip.so "$@" sparc_amd/libwedm_hip.so; do
    run $lib config3 2 && run $lib config2 8 || exit 1
done
if [ -f build/ablate/libwedm_STAMPS.so ]; then
    WEDM_HIP_LIB=build/ablate/libwedm_STAMPS.so python tools/stamps_stream.py 65536 config3 2
fi
