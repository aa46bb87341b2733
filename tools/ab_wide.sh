#!/bin/bash
# tools/ab_wide.sh [lib ...]: the wide register kernel (kernel 8) against the LDS kernels on ONE box: BASELINE configs[1]
# (4 096 x 400), other batch sizes of the same wire, a 15 um gap (dense sparking), single-microsecond launches; then the
# config-2 line of every library named (variants built with tools/build_variant.py)
OUT=gpurun_out/ab_wide; mkdir -p $OUT
line() {  # tag lib flags...
    local tag=$1 lib=$2; shift 2
    WEDM_HIP_LIB=$lib python bench.py --workload config2 --no-cpu-baseline --no-side "$@" > $OUT/$tag.json 2>$OUT/$tag.err || { echo "$tag FAILED"; tail -3 $OUT/$tag.err; return 0; }
    python -c "
import json; d=json.load(open('$OUT/$tag.json')); print('%-28s %-44s env-steps/s %.4g  ms/step %.4f' % ('$tag', d['config']['kernel'], d['value'], d['ms_per_step']))"
}
T=sparc_amd/libwedm_hip.so
for n in 256 1024 2048 4096 8192 16384; do
    line n${n}_auto $T --steps 10 --warmup 2 --num-envs $n
    line n${n}_wide $T --steps 10 --warmup 2 --num-envs $n --kernel 8
    line n${n}_fused16 $T --steps 10 --warmup 2 --num-envs $n --kernel 3 --lanes 16
    line n${n}_packed8 $T --steps 10 --warmup 2 --num-envs $n --kernel 4 --lanes 8
done
line gap15_wide $T --steps 10 --warmup 2 --gap 15
line gap15_fused16 $T --steps 10 --warmup 2 --gap 15 --kernel 3 --lanes 16
line gap7_wide $T --steps 10 --warmup 2 --gap 7.4
line gap7_fused16 $T --steps 10 --warmup 2 --gap 7.4 --kernel 3 --lanes 16
line us1_stream $T --steps 2000 --warmup 100 --substeps 1
line us1_wide $T --steps 2000 --warmup 100 --substeps 1 --kernel 8
for lib in "$@"; do
    tag=$(basename $lib .so)
    for n in 4096 8192 16384; do line ${tag}_n$n $lib --steps 10 --warmup 2 --num-envs $n --kernel 8; done
done
line n4096_auto_again $T --steps 10 --warmup 2
