#!/bin/bash
# A/B of the in-tree library against build/ablate/libwedm_PREV.so on one box: parity subset first, then the default
# bench line (headline + side measurements), the single-microsecond kernels and the 400-segment workloads
set -o pipefail
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fuzz or ragged or dense or config3_grid or per_environment or default_config or autoreset or single_microsecond or config4 or trace" 2>&1 | tail -3 || exit 1
bash tools/ab_full.sh build/ablate/libwedm_PREV.so
bash tools/ab_stream.sh build/ablate/libwedm_PREV.so | grep us/launch
bash tools/ab_bench.sh "config2 config4" "" build/ablate/libwedm_PREV.so
