#!/bin/bash
# tools/ab_1us.sh "<lib or - for in-tree> ..." : single-microsecond launches, kernel durations from rocprofv3
export TMPDIR=/tmp
for lib in "$@"; do
  [ "$lib" = "-" ] && lib=""
  for kk in 1 5; do
    for w in config3 config4; do
      d=gpurun_out/kt1us/$(basename ${lib:-intree} .so)_${kk}_$w
      WEDM_HIP_LIB=${lib:+$PWD/$lib} rocprofv3 --kernel-trace --stats --output-format csv -d $d -o kt -- python3 bench.py --steps 500 --warmup 20 --substeps 1 --kernel $kk --workload $w --no-cpu-baseline > $d.log 2>&1
      grep wedm_step $d/kt_kernel_stats.csv | cut -d, -f1,4 | sed "s|^|$w k$kk ${lib:-in-tree} |"
    done
  done
done
