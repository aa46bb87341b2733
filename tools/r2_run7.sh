#!/bin/bash
export TMPDIR=/tmp
OUT=gpurun_out/r2g
mkdir -p $OUT
python -m pytest tests -m gpu -q -x -k "default_config_fused or config3_grid or tiny_wires or in_kernel_autoreset or randomized_configurations or float64" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
WEDM_HIP_LIB=build/ablate/libwedm_STAMPS.so python tools/stamps_stream.py 65536 config3 2 2>&1 | grep -v amdgpu.ids
WEDM_HIP_LIB=build/ablate/libwedm_STAMPS.so python tools/stamps_stream.py 256 config3 2 2>&1 | grep -v amdgpu.ids
run() {  # workload kernel lanes
    d=$OUT/kt_$1_k$2_l$3
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o kt -- python3 bench.py --steps 400 --warmup 50 --substeps 1 --kernel $2 --lanes $3 --workload $1 --no-cpu-baseline > $d.log 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    if "wedm_step" in r["Name"]: print("$1 k$2 l$3", r["Name"], "calls", r["Calls"], "avg_us %.2f" % (float(r["AverageNs"])/1e3))
PY
}
run config3 6 2
run config4 6 8
run config2 6 8
