#!/bin/bash
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 || exit 1
bash tools/ab_bench.sh "config5 config3 config4 config2" "" build/ablate/libwedm_PREV.so
