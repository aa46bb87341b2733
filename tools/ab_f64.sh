#!/bin/bash
# A/B of the float64-typed kernels (GPU box): configs[2] and configs[1] on the automatic plan and by name, in-tree build and
# build/ablate/libwedm_<TAG>.so variants.
#     bash tools/ab_f64.sh [TAG ...]
run() {  # tag, lib ("" = in-tree), extra bench flags
  tag=$1; lib=$2; shift 2
  WEDM_HIP_LIB=$lib timeout -k 10 120 python bench.py --stencil-dtype float64 --no-side --no-cpu-baseline --steps 6 --warmup 2 "$@" > /tmp/ab_f64.json 2>/tmp/ab_f64.err \
    && python tools/bench_line.py "$tag" /tmp/ab_f64.json || { echo "$tag failed"; tail -2 /tmp/ab_f64.err; }
}
run "configs[2] auto" ""
run "configs[2] regs<1>" "" --kernel 7 --lanes 1
run "configs[2] fused (kernel 3)" "" --kernel 3
run "configs[1] auto" "" --workload config2
run "configs[1] fused (kernel 3)" "" --workload config2 --kernel 3
run "configs[3] shard auto" "" --workload config4
run "configs[3] shard fused (kernel 3)" "" --workload config4 --kernel 3
run "configs[4] shard auto" "" --workload config5
run "configs[4] shard cell by cell (kernel 10)" "" --workload config5 --kernel 10
for t in "$@"; do run "$t configs[2] auto" build/ablate/libwedm_$t.so; run "$t configs[1] auto" build/ablate/libwedm_$t.so --workload config2; done
