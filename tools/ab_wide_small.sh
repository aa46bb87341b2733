#!/bin/bash
# tools/ab_wide_small.sh: the wide register kernel (kernel 8; 4 / 8 lanes per environment for wires of up to 128 / 256
# segments) against the automatic choice among the other kernels, the packed LDS kernel and the two-lane register kernel,
# over batch sizes of the headline grid (128 segments) and of a 200-segment wire, on ONE box
for n in 64 1024 4096 8192 16384 24576 32768; do
    for kl in "8 0" "4 0" "3 0" "7 2"; do python tools/probe_shape.py $n 0.625 $kl 2>&1 | grep -v amdgpu.ids | tail -1; done
done
for n in 1024 4096 8192 16384; do
    for kl in "8 0" "4 0" "3 0"; do python tools/probe_shape.py $n 0.4 $kl 2>&1 | grep -v amdgpu.ids | tail -1; done
done
