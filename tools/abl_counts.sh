#!/bin/bash
# dynamic instruction counts per wave-step of ablated builds (attribution of the VALU budget)
export TMPDIR=/tmp
for lib in "" $(ls build/ablate/libwedm_ABL_*.so); do
  d=gpurun_out/abl/$(basename ${lib:-full} .so)
  WEDM_HIP_LIB=${lib:+$PWD/$lib} rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $d -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $d.log 2>&1
  python tools/pmc_summary.py $d | awk -v n="$(basename ${lib:-full} .so)" '{printf "%s %s %.0f per wave-step\n", n, $3, $4/2048/1000}'
  WEDM_HIP_LIB=${lib:+$PWD/$lib} python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   kernel ms', round(d['roofline']['kernel_ms'],3))"
done
