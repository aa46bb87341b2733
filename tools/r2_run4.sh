#!/bin/bash
# single-microsecond launches: kernel durations from rocprofv3 --kernel-trace --stats (HIP events around a
# host-paced stream of 20-us launches measure the host, not the kernel)
export TMPDIR=/tmp
OUT=gpurun_out/r2d
mkdir -p $OUT
run() {  # workload kernel lanes
    d=$OUT/kt_$1_k$2_l$3
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o kt -- python3 bench.py --steps 400 --warmup 50 --substeps 1 --kernel $2 --lanes $3 --workload $1 --no-cpu-baseline > $d.log 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    grep wedm_step $f | awk -F, -v tag="$1 k$2 l$3" '{gsub(/"/,""); print tag, $1, "calls", $2, "avg_ns", $4}'
}
run config3 5 0
run config3 1 0
for l in 1 2 4; do run config3 9 $l; done
run config3 6 0
run config4 5 0
for l in 4 8; do run config4 9 $l; done
run config2 5 0
for l in 4 8 16; do run config2 9 $l; done
