#!/bin/bash
# A/B several builds of libarvx.so on ONE box, interleaved (device-to-device spread is ~10 %).
# usage (on the GPU box): tools/ab_compare.sh "<grids>" <libA.so> <libB.so> ...
GRIDS=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    ARVX_LIB_PATH=$lib python tools/carve_stats.py $GRIDS 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/'
  done
done
