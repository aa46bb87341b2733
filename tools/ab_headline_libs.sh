#!/bin/bash
# A/B of the headline launch (GPU box): the in-tree build and build/ablate/libwedm_<TAG>.so variants.   bash tools/ab_headline_libs.sh TAG...
run() { tag=$1; lib=$2; shift 2
  WEDM_HIP_LIB=$lib timeout -k 10 120 python bench.py --no-side --no-cpu-baseline --steps 10 --warmup 2 "$@" > /tmp/ab_h.json 2>/tmp/ab_h.err \
    && python tools/bench_line.py "$tag" /tmp/ab_h.json || { echo "$tag failed"; tail -2 /tmp/ab_h.err; }; }
for t in "" "$@" ""; do lib=${t:+build/ablate/libwedm_$t.so}; run "${t:-in-tree}" "$lib"; done
