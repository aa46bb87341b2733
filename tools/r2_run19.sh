#!/bin/bash
# A/B of an alternative library ($1) against the in-tree one: parity subset with the ALTERNATIVE, then the bench lines
set -o pipefail
WEDM_HIP_LIB=$1 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fuzz or ragged or dense or config3_grid or default_config or autoreset" 2>&1 | tail -3 || exit 1
bash tools/ab_full.sh $1
