#!/bin/bash
# A/B of the float64-typed wide register kernel (GPU box): 4 096 / 32 768 x 400, in-tree build and build/ablate/libwedm_<TAG>.so.
#     bash tools/ab_f64_wide.sh [TAG ...]
run() {  # tag, lib ("" = in-tree), extra bench flags
  tag=$1; lib=$2; shift 2
  WEDM_HIP_LIB=$lib timeout -k 10 120 python bench.py --stencil-dtype float64 --no-side --no-cpu-baseline --steps 4 --warmup 1 "$@" > /tmp/ab_f64.json 2>/tmp/ab_f64.err \
    && python tools/bench_line.py "$tag" /tmp/ab_f64.json || { echo "$tag failed"; tail -2 /tmp/ab_f64.err; }
}
for t in "" "$@"; do
  lib=${t:+build/ablate/libwedm_$t.so}
  run "${t:-in-tree} 4096 x 400" "$lib" --workload config2 --kernel 8 --lanes 16
  run "${t:-in-tree} 32768 x 400" "$lib" --workload config4 --kernel 8 --lanes 16
done
