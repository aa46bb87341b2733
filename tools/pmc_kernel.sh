#!/bin/bash
# tools/pmc_kernel.sh <tag> "<bench flags>": issue counters of one bench configuration (separate rocprofv3 --pmc passes)
TAG=$1; FLAGS=$2; OUT=gpurun_out/pmc_$TAG; mkdir -p $OUT; export TMPDIR=/tmp
pmc() { tag=$1; shift; rocprofv3 --output-format csv --pmc "$@" -d $OUT/$tag -o p -- python3 bench.py --no-cpu-baseline --no-side --steps 3 --warmup 1 $FLAGS > $OUT/$tag.log 2>&1; }
pmc a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
pmc b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM
python tools/pmc_summary.py $OUT/a $OUT/b
