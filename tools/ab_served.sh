#!/bin/bash
# served kernel (9) against the automatic choice, S = 400, several batch sizes (run ON THE GPU BOX):
#     gpurun --timeout 600 -- 'bash tools/ab_served.sh > gpurun_out/ab_served.txt 2>&1'
set -o pipefail
run() {  # tag, bench flags...
  tag=$1; shift
  timeout -k 10 120 python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-side > /tmp/ab_line.json 2> /tmp/ab_line.err || { echo "$tag FAILED"; tail -3 /tmp/ab_line.err; return 1; }
  python tools/bench_line.py "$tag" /tmp/ab_line.json
}
for n in 32768 16384 8192 4096 65536; do
  for k in "0 0" "9 8" "9 4" "4 8"; do
    set -- $k
    run "$n x 400 kernel $1 lanes $2" --workload config4 --num-envs $n --kernel $1 --lanes $2 || exit 1
  done
done
for g in 15 7.4; do
  for k in "0 0" "9 8"; do
    set -- $k
    run "32768 x 400 gap $g kernel $1 lanes $2" --workload config4 --gap $g --kernel $1 --lanes $2 || exit 1
  done
done
