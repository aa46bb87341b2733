#!/bin/bash
# Measurement pass of one round, run ON THE GPU BOX from the repository root:
#     gpurun --timeout 1100 -- 'bash tools/profile_round.sh r3'
# Writes raw output under gpurun_out/<round>/ and the summaries the DESIGN.md tables quote
# under gpurun_out/<round>/summary/ (copy those into profiles/<round>/ and commit them).
set -o pipefail
R=${1:-r3}
OUT=gpurun_out/$R
S=$OUT/summary
mkdir -p $S
export TMPDIR=/tmp

bench_lines() {  # $1: extra flags of the headline run ("" = with the CPU baseline leg)
python bench.py --steps 20 --warmup 2 $1 > $S/bench_config3.json 2> $OUT/bench_config3.err
python bench.py --steps 10 --warmup 2 --workload config2 --no-cpu-baseline > $S/bench_config2.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --workload config4 --no-cpu-baseline > $S/bench_config4_shard.json 2>/dev/null
python bench.py --steps 10 --warmup 2 --workload config5 --no-cpu-baseline > $S/bench_config5_shard.json 2>/dev/null
python bench.py --steps 2000 --warmup 100 --substeps 1 --no-cpu-baseline --no-side > $S/bench_config3_1us.json 2>/dev/null
python bench.py --steps 2000 --warmup 100 --substeps 1 --workload config4 --no-cpu-baseline > $S/bench_config4_1us.json 2>/dev/null
python bench.py --steps 2000 --warmup 100 --substeps 1 --workload config2 --no-cpu-baseline > $S/bench_config2_1us.json 2>/dev/null
}
echo "[profile] bench lines, first pass (kernel names and units per launch for the PMC tables)"; date +%T
bench_lines "--no-cpu-baseline --no-side"

echo "[profile] PMC passes (one counter group per run)"; date +%T
pmc() {  # tag counters... -- bench args
    tag=$1; shift
    ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
    rocprofv3 --output-format csv --pmc "${ctrs[@]}" -d $OUT/pmc_$tag -o p -- python3 bench.py --no-cpu-baseline --no-side "$@" > $OUT/pmc_$tag.log 2>&1
}
for w in config3 config4 config2 config5; do
  pmc fetch_$w FETCH_SIZE -- --steps 4 --warmup 1 --workload $w
  pmc write_$w WRITE_SIZE -- --steps 4 --warmup 1 --workload $w
  pmc sq_$w SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- --steps 4 --warmup 1 --workload $w
  python tools/pmc_summary.py $OUT/pmc_fetch_$w $OUT/pmc_write_$w > $S/rocprofv3_pmc_hbm_$w.txt
  python tools/pmc_summary.py $OUT/pmc_sq_$w > $S/rocprofv3_pmc_sq_$w.txt
done
pmc fetch_1us FETCH_SIZE -- --steps 40 --warmup 5 --substeps 1
pmc write_1us WRITE_SIZE -- --steps 40 --warmup 5 --substeps 1
pmc sq_1us SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS -- --steps 40 --warmup 5 --substeps 1
python tools/pmc_summary.py $OUT/pmc_fetch_1us $OUT/pmc_write_1us > $S/rocprofv3_pmc_hbm_config3_1us.txt
python tools/pmc_summary.py $OUT/pmc_sq_1us > $S/rocprofv3_pmc_sq_config3_1us.txt
echo "[profile] recorded counter tables -> profiles/traffic.json, profiles/valu.json (bench.py reads them)"; date +%T
python tools/make_traffic_json.py $S profiles/$R
cp profiles/traffic.json profiles/valu.json $S/

echo "[profile] bench lines, final pass (roofline objects use the counters just recorded)"; date +%T
bench_lines ""
python bench.py --steps 5 --warmup 1 --stencil-dtype float64 --no-cpu-baseline > $S/bench_config3_f64.json 2>/dev/null
python bench.py --steps 5 --warmup 1 --stencil-dtype float64 --workload config2 --no-cpu-baseline > $S/bench_config2_f64.json 2>/dev/null
WEDM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-side > $S/bench_config3_rccl_world1.json 2>/dev/null

echo "[profile] rocprofv3 kernel trace"; date +%T
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-side > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $S/rocprofv3_kernel_stats_config3.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1 -o kt -- python3 bench.py --steps 400 --warmup 50 --substeps 1 --no-cpu-baseline --no-side > $OUT/kt1.log 2>&1
cp $(find $OUT/kt1 -name "*kernel_stats.csv" | head -1) $S/rocprofv3_kernel_stats_config3_1us.csv

rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt2 -o kt -- python3 bench.py --steps 20 --warmup 2 --workload config2 --no-cpu-baseline --no-side > $OUT/kt2.log 2>&1
cp $(find $OUT/kt2 -name "*kernel_stats.csv" | head -1) $S/rocprofv3_kernel_stats_config2.csv

echo "[profile] done"; date +%T
ls -la $S
