#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "6-0 or 6-4 or 6-8 or 6-16 or single_microsecond or autoreset or randomized or trace or F16 or fixture" 2>&1 | tail -3 || exit 1
bash tools/ab_stream.sh build/ablate/libwedm_PREV.so
