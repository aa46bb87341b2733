#!/bin/bash
# tools/ab_full.sh lib ... : the default bench line (headline + side measurements: 1-us cadence, 15-um gap, closed
# loop) with the in-tree library and each library named, alternating, twice -- A/B on ONE box
OUT=gpurun_out/ab_full; mkdir -p $OUT
for rep in 1 2; do
  for lib in "" "$@"; do
    tag=$(basename ${lib:-intree} .so)
    WEDM_HIP_LIB=${lib:+$PWD/$lib} python bench.py --no-cpu-baseline > $OUT/$tag.$rep.json 2>$OUT/$tag.$rep.err || { tail -3 $OUT/$tag.$rep.err; exit 1; }
    python - <<PY
import json
d = json.load(open("$OUT/$tag.$rep.json"))
print("$tag", "headline %.4e" % d["value"], "|", " | ".join("%s %.4g %s" % (s["name"][:28], s["value"], s.get("kernel_us", "")) for s in d.get("side", [])))
PY
  done
done
