#!/bin/bash
# tools/ab_shape.sh "<num_envs segment_len kernel lanes>" lib...: fused launches of one shape (tools/probe_shape.py), the libraries
# named alternating three times on ONE box (box-to-box variance is 1 - 2 %, the effects being chased are often smaller)
SHAPE=$1; shift
for i in 1 2 3; do
  for lib in "$@"; do
    echo -n "$(basename $lib .so) "; WEDM_HIP_LIB=$lib python tools/probe_shape.py $SHAPE 2>&1 | tail -1
  done
done
