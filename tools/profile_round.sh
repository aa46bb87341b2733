#!/bin/bash
# Measurement pass of one round, run ON THE GPU BOX from the repository root:
#     gpurun --timeout 1100 -- 'bash tools/profile_round.sh r1'
# Writes raw output under gpurun_out/<round>/ and the summaries the DESIGN.md tables quote
# under gpurun_out/<round>/summary/ (copy those into profiles/<round>/ and commit them).
set -e -o pipefail
R=${1:-r1}
OUT=gpurun_out/$R
S=$OUT/summary
mkdir -p $S
export TMPDIR=/tmp

echo "[profile] bench lines"; date +%T
python bench.py --steps 20 --warmup 2 > $S/bench_config3.json 2> $OUT/bench_config3.err
python bench.py --steps 10 --warmup 2 --workload config2 --no-cpu-baseline > $S/bench_config2.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --workload config4 --no-cpu-baseline > $S/bench_config4_shard.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --workload config5 --no-cpu-baseline > $S/bench_config5_shard.json 2>/dev/null
python bench.py --steps 2000 --warmup 100 --substeps 1 --no-cpu-baseline > $S/bench_config3_1us.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --trace voltage --no-cpu-baseline > $S/bench_config3_trace_voltage.json 2>/dev/null
WEDM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $S/bench_config3_rccl_world1.json 2>/dev/null

echo "[profile] rocprofv3 kernel trace"; date +%T
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $S/rocprofv3_kernel_stats_config3.csv

echo "[profile] PMC passes (one counter group per run)"; date +%T
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
    -d $OUT/pmc_sq -o p -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
python tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write > $S/rocprofv3_pmc_hbm_config3.txt
python tools/pmc_summary.py $OUT/pmc_sq > $S/rocprofv3_pmc_sq_config3.txt
for w in config4 config2; do
  rocprofv3 --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch_$w -o p -- python3 bench.py --steps 4 --warmup 1 --workload $w --no-cpu-baseline > $OUT/pmc_fetch_$w.log 2>&1
  rocprofv3 --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write_$w -o p -- python3 bench.py --steps 4 --warmup 1 --workload $w --no-cpu-baseline > $OUT/pmc_write_$w.log 2>&1
  python tools/pmc_summary.py $OUT/pmc_fetch_$w $OUT/pmc_write_$w > $S/rocprofv3_pmc_hbm_$w.txt
done
echo "[profile] done"; date +%T
ls -la $S
