#!/bin/bash
# tools/pmc_1us.sh <round>: what holds the single-microsecond kernel's memory phase back -- vector-memory issue cycles and the
# texture-addresser / L1 busy and stall counters of `bench.py --substeps 1` (separate rocprofv3 --pmc passes).
R=${1:-r3}; OUT=gpurun_out/$R; mkdir -p $OUT/summary; export TMPDIR=/tmp
pmc() { tag=$1; shift; rocprofv3 --output-format csv --pmc "$@" -d $OUT/pmc1_$tag -o p -- python3 bench.py --no-cpu-baseline --no-side --steps 40 --warmup 5 --substeps 1 > $OUT/pmc1_$tag.log 2>&1; }
pmc sqvm SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc ta TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pmc tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum GRBM_GUI_ACTIVE
python tools/pmc_summary.py $OUT/pmc1_sqvm $OUT/pmc1_ta $OUT/pmc1_tcp > $OUT/summary/rocprofv3_pmc_vmem_config3_1us.txt
cat $OUT/summary/rocprofv3_pmc_vmem_config3_1us.txt
