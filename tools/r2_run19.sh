#!/bin/bash
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fuzz or ragged or dense or config3_grid or default_config" 2>&1 | tail -3 || exit 1
bash tools/ab_full.sh build/ablate/libwedm_PREV.so
