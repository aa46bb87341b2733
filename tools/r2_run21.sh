#!/bin/bash
set -o pipefail
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" | tail -1 || exit 1
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3 || exit 1
bash tools/ab_bench.sh "config3 config2 config4 config5" "" build/ablate/libwedm_PREV.so
bash tools/ab_stream.sh build/ablate/libwedm_PREV.so | grep us/launch
