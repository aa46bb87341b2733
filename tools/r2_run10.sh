#!/bin/bash
export TMPDIR=/tmp
echo "== config2 kernel variants"
bash tools/ab_kernels.sh config2 "" "3:16 3:8 4:8 4:4 2:16 2:8"
echo "== config2 stamps (fused<16>, fused<8>, packed<8>)"
export WEDM_HIP_LIB=build/ablate/libwedm_STAMPS.so
python tools/stamps.py 3 16 config2 2>&1 | grep -v amdgpu
python tools/stamps.py 3 8 config2 2>&1 | grep -v amdgpu
python tools/stamps.py 4 8 config2 2>&1 | grep -v amdgpu
echo "== config3 stamps packed<2> (bench workload, then 15 um gap)"
python tools/stamps.py 4 2 config3 2>&1 | grep -v amdgpu
python tools/stamps.py 4 2 config3 65536 15 2>&1 | grep -v amdgpu
python tools/stamps.py 3 8 config4 2>&1 | grep -v amdgpu
