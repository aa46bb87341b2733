#!/bin/bash
# kernel duration of single-microsecond launches vs batch size (rocprofv3 kernel trace)
export TMPDIR=/tmp
mkdir -p gpurun_out/kt1us
for kk in ${KERNELS:-1 5}; do
  for n in 4096 16384 65536 262144 1048576; do
    d=gpurun_out/kt1us/scale_${kk}_$n
    rocprofv3 --kernel-trace --stats --output-format csv -d $d -o kt -- python3 bench.py --steps 200 --warmup 20 --substeps 1 --kernel $kk --num-envs $n --no-cpu-baseline > $d.log 2>&1
    grep wedm_step $d/kt_kernel_stats.csv | cut -d, -f1,4 | sed "s|^|N=$n k$kk |"
  done
done
