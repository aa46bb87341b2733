#!/bin/bash
# in-tree library against build/ablate/libwedm_PREV.so: full GPU suite, a batch whose environments keep terminating
# (autoreset handle), and the standard A/B
set -o pipefail
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 || exit 1
python tools/terminating_batch.py 30 0.002
WEDM_HIP_LIB=build/ablate/libwedm_PREV.so python tools/terminating_batch.py 30 0.002
bash tools/ab_full.sh build/ablate/libwedm_PREV.so
bash tools/ab_bench.sh "config2 config4" "" build/ablate/libwedm_PREV.so
