#!/bin/bash
# Measurement pass of one round, run ON THE GPU BOX from the repository root:
#     gpurun --timeout 1100 -- 'bash tools/profile_round.sh r4'
# Writes raw output under gpurun_out/<round>/ and the summaries the DESIGN.md tables quote
# under gpurun_out/<round>/summary/ (copy those into profiles/<round>/ and commit them).
#
# Every leg is checked: a bench leg whose output is empty, is not JSON or names another build than the loaded library's
# stops the script (stderr of every leg is kept under $OUT/*.err); the pmc_* / kt* directories are removed before each
# pass, so no counter or trace file of an earlier run can be picked up under the new build id.
set -o pipefail
R=${1:-r4}
OUT=gpurun_out/$R
S=$OUT/summary
rm -rf $OUT
mkdir -p $S
export TMPDIR=/tmp
BUILD=$(python -c "from sparc_amd import _lib; print(_lib.build_id())" 2>/dev/null)
[ -n "$BUILD" ] || { echo "[profile] cannot load the library"; exit 1; }
echo "[profile] build $BUILD"

check_line() {  # file: one JSON bench line of THIS build
  python - "$1" "$BUILD" <<'PY' || { echo "[profile] bad bench record $1 (see ${1%.json}.err)"; exit 1; }
import json, sys
b = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
assert b["config"]["build_id"] == sys.argv[2], (b["config"]["build_id"], sys.argv[2])
assert b["value"] > 0
PY
}
bench() {  # name, flags...
  name=$1; shift
  python bench.py "$@" > $S/$name.json.tmp 2> $OUT/$name.err || { echo "[profile] bench leg $name failed"; tail -3 $OUT/$name.err; exit 1; }
  grep '^{' $S/$name.json.tmp | tail -1 > $S/$name.json; rm -f $S/$name.json.tmp   # (an RCCL banner may precede the line)
  check_line $S/$name.json
}
bench_lines() {  # $1: extra flags of the headline run ("" = with the CPU baseline leg and the side lines)
  bench bench_config3 --steps 20 --warmup 5 $1
  bench bench_config2 --steps 10 --warmup 2 --workload config2 --no-cpu-baseline
  bench bench_config4_shard --steps 10 --warmup 2 --workload config4 --no-cpu-baseline
  bench bench_config5_shard --steps 10 --warmup 2 --workload config5 --no-cpu-baseline
  bench bench_config3_1us --steps 2000 --warmup 100 --substeps 1 --no-cpu-baseline --no-side
  bench bench_config4_1us --steps 2000 --warmup 100 --substeps 1 --workload config4 --no-cpu-baseline
  bench bench_config2_1us --steps 2000 --warmup 100 --substeps 1 --workload config2 --no-cpu-baseline
  bench bench_config3_f64 --steps 5 --warmup 1 --stencil-dtype float64 --no-cpu-baseline --no-side
  bench bench_config2_f64 --steps 5 --warmup 1 --stencil-dtype float64 --workload config2 --no-cpu-baseline
  bench bench_config4_f64 --steps 4 --warmup 1 --stencil-dtype float64 --workload config4 --no-cpu-baseline
  bench bench_config5_f64 --steps 4 --warmup 1 --stencil-dtype float64 --workload config5 --no-cpu-baseline
}
echo "[profile] bench lines, first pass (kernel names and units per launch for the PMC tables)"; date +%T
bench_lines "--no-cpu-baseline --no-side"

echo "[profile] PMC passes (one counter group per run)"; date +%T
pmc() {  # tag counters... -- bench args
  tag=$1; shift
  ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rm -rf $OUT/pmc_$tag
  rocprofv3 --output-format csv --pmc "${ctrs[@]}" -d $OUT/pmc_$tag -o p -- python3 bench.py --no-cpu-baseline --no-side "$@" > $OUT/pmc_$tag.log 2>&1 \
    || { echo "[profile] pmc pass $tag failed"; tail -3 $OUT/pmc_$tag.log; exit 1; }
  ls $OUT/pmc_$tag/*counter_collection.csv > /dev/null 2>&1 || { echo "[profile] pmc pass $tag wrote no counters"; exit 1; }
}
for w in config3 config4 config2 config5; do
  pmc fetch_$w FETCH_SIZE -- --steps 4 --warmup 1 --workload $w
  pmc write_$w WRITE_SIZE -- --steps 4 --warmup 1 --workload $w
  pmc sq_$w SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- --steps 4 --warmup 1 --workload $w
  python tools/pmc_summary.py $OUT/pmc_fetch_$w $OUT/pmc_write_$w > $S/rocprofv3_pmc_hbm_$w.txt
  python tools/pmc_summary.py $OUT/pmc_sq_$w > $S/rocprofv3_pmc_sq_$w.txt
done
# the stencil in Numba's typing (stencil_mode 1) on the register kernel
pmc fetch_config3_f64 FETCH_SIZE -- --steps 4 --warmup 1 --stencil-dtype float64
pmc write_config3_f64 WRITE_SIZE -- --steps 4 --warmup 1 --stencil-dtype float64
pmc sq_config3_f64 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- --steps 4 --warmup 1 --stencil-dtype float64
python tools/pmc_summary.py $OUT/pmc_fetch_config3_f64 $OUT/pmc_write_config3_f64 > $S/rocprofv3_pmc_hbm_config3_f64.txt
python tools/pmc_summary.py $OUT/pmc_sq_config3_f64 > $S/rocprofv3_pmc_sq_config3_f64.txt
for w in config2 config4 config5; do   # ... and on the wide register kernel (one / two blocks per CU) and the packed any-geometry kernel
  pmc fetch_${w}_f64 FETCH_SIZE -- --steps 4 --warmup 1 --workload $w --stencil-dtype float64
  pmc write_${w}_f64 WRITE_SIZE -- --steps 4 --warmup 1 --workload $w --stencil-dtype float64
  pmc sq_${w}_f64 SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- --steps 4 --warmup 1 --workload $w --stencil-dtype float64
  python tools/pmc_summary.py $OUT/pmc_fetch_${w}_f64 $OUT/pmc_write_${w}_f64 > $S/rocprofv3_pmc_hbm_${w}_f64.txt
  python tools/pmc_summary.py $OUT/pmc_sq_${w}_f64 > $S/rocprofv3_pmc_sq_${w}_f64.txt
done
# the kernel the served kernel replaced at 32 768 x 400, same counters (what the served form saves)
pmc sq_config4_packed SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- --steps 4 --warmup 1 --workload config4 --kernel 4 --lanes 8
python tools/pmc_summary.py $OUT/pmc_sq_config4_packed > $S/rocprofv3_pmc_sq_config4_packed.txt
pmc sq_config5_cellwise SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -- --steps 4 --warmup 1 --workload config5 --kernel 10
python tools/pmc_summary.py $OUT/pmc_sq_config5_cellwise > $S/rocprofv3_pmc_sq_config5_cellwise.txt
pmc fetch_1us FETCH_SIZE -- --steps 40 --warmup 5 --substeps 1
pmc write_1us WRITE_SIZE -- --steps 40 --warmup 5 --substeps 1
pmc sq_1us SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS -- --steps 40 --warmup 5 --substeps 1
python tools/pmc_summary.py $OUT/pmc_fetch_1us $OUT/pmc_write_1us > $S/rocprofv3_pmc_hbm_config3_1us.txt
python tools/pmc_summary.py $OUT/pmc_sq_1us > $S/rocprofv3_pmc_sq_config3_1us.txt
echo "[profile] recorded counter tables -> profiles/traffic.json, profiles/valu.json (bench.py reads them)"; date +%T
python tools/make_traffic_json.py $S profiles/$R || exit 1
cp profiles/traffic.json profiles/valu.json $S/

echo "[profile] bench lines, final pass (roofline objects use the counters just recorded)"; date +%T
bench_lines ""
bench bench_config3_f64_1us --steps 2000 --warmup 100 --substeps 1 --stencil-dtype float64 --no-cpu-baseline --no-side
bench bench_config3_trace_voltage --steps 5 --warmup 1 --trace voltage --no-cpu-baseline
bench bench_config3_trace_signals --steps 5 --warmup 1 --trace signals --no-cpu-baseline
WEDM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
    bench bench_config3_rccl_world1 --steps 10 --warmup 2 --no-cpu-baseline --no-side
for w in config4 config5; do
  WEDM_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29534 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
      bench bench_${w}_rccl_world1 --steps 10 --warmup 2 --no-cpu-baseline --no-side --workload $w
done

echo "[profile] rocprofv3 kernel traces"; date +%T
ktrace() {  # tag, csv name, bench args...
  tag=$1; csv=$2; shift 2
  rm -rf $OUT/$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag -o kt -- python3 bench.py --no-cpu-baseline --no-side "$@" > $OUT/$tag.log 2>&1 \
    || { echo "[profile] kernel trace $tag failed"; tail -3 $OUT/$tag.log; exit 1; }
  f=$(find $OUT/$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "[profile] kernel trace $tag wrote no stats"; exit 1; }
  cp $f $S/$csv
}
ktrace kt rocprofv3_kernel_stats_config3.csv --steps 20 --warmup 2
ktrace kt1 rocprofv3_kernel_stats_config3_1us.csv --steps 400 --warmup 50 --substeps 1
ktrace kt2 rocprofv3_kernel_stats_config2.csv --steps 20 --warmup 2 --workload config2
ktrace kt4 rocprofv3_kernel_stats_config4.csv --steps 20 --warmup 2 --workload config4
ktrace kt5 rocprofv3_kernel_stats_config5.csv --steps 20 --warmup 2 --workload config5

echo "[profile] launch plan sweep, served kernel A/B and stamps"; date +%T
python tools/plan_sweep.py > $S/plan_sweep.txt 2> $OUT/plan_sweep.err || { echo "[profile] plan sweep failed"; tail -3 $OUT/plan_sweep.err; exit 1; }
python tools/plan_sweep.py --f64 > $S/plan_sweep_f64.txt 2> $OUT/plan_sweep_f64.err || { echo "[profile] float64 plan sweep failed"; tail -3 $OUT/plan_sweep_f64.err; exit 1; }
bash tools/ab_served.sh > $S/ab_served.txt 2>&1 || { echo "[profile] ab_served failed"; tail -3 $S/ab_served.txt; }
if [ -f build/ablate/libwedm_SVSTAMPS.so ]; then
  for n in 8192 32768; do WEDM_HIP_LIB=build/ablate/libwedm_SVSTAMPS.so python tools/stamps_served.py 8 $n 2>/dev/null; done > $S/stamps_served.txt
fi

echo "[profile] done"; date +%T
ls -la $S
