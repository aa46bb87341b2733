#!/bin/bash
# batched cold loads in the general prelude: parity subset, A/B against the previous build, stamps
set -o pipefail
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fuzz or ragged or dense or config3_grid or per_environment or default_config or autoreset" 2>&1 | tail -3 || exit 1
bash tools/ab_full.sh build/ablate/libwedm_PREV.so
bash tools/ab_stream.sh build/ablate/libwedm_PREV.so | grep us/launch
bash tools/ab_bench.sh "config2 config4" "" build/ablate/libwedm_PREV.so
export WEDM_HIP_LIB=build/ablate/libwedm_STAMPS.so; python tools/stamps.py 4 2 config3 65536 7.4
