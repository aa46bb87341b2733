#!/bin/bash
# tools/ab_served_libs.sh "<bench flags>" lib...: one served-kernel workload on the in-tree library and on ablation builds
# (tools/build_variant.py TAG --only-part 3 -D...), A/B on ONE box
FLAGS=$1; shift
for lib in sparc_amd/libwedm_hip.so "$@"; do
  tag=$(basename $lib .so)
  WEDM_HIP_LIB=$lib timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side $FLAGS > /tmp/ab_line.json 2> /tmp/ab_line.err || { echo "$tag FAILED"; tail -3 /tmp/ab_line.err; exit 1; }
  python tools/bench_line.py "$tag [$FLAGS]" /tmp/ab_line.json
done
