#!/bin/bash
# A/B of the float64-typed register kernel on configs[2] (GPU box): the in-tree build and build/ablate/libwedm_<TAG>.so variants.
#     bash tools/ab_f64.sh [TAG ...]
run() {  # tag, lib ("" = in-tree), extra bench flags
  tag=$1; lib=$2; shift 2
  WEDM_HIP_LIB=$lib timeout -k 10 120 python bench.py --stencil-dtype float64 --no-side --no-cpu-baseline --steps 6 --warmup 2 "$@" > /tmp/ab_f64.json 2>/tmp/ab_f64.err \
    && python tools/bench_line.py "$tag" /tmp/ab_f64.json || { echo "$tag failed"; tail -2 /tmp/ab_f64.err; }
}
run "in-tree auto" ""
run "in-tree regs<1>" "" --kernel 7 --lanes 1
run "in-tree fused (kernel 3)" "" --kernel 3
for t in "$@"; do run "$t auto" build/ablate/libwedm_$t.so; done
run "in-tree auto (again)" ""
