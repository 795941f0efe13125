#!/bin/bash
# A/B several builds on one box: full grids (tools/carve_stats.py) and one Z slab of the 8-GPU grid
# usage (GPU box): tools/ab_slab.sh "<grids>" <libA.so> <libB.so> ...
GRIDS=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib"
    ARVX_LIB_PATH=$lib python tools/carve_stats.py $GRIDS 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/' | cut -c1-100
    ARVX_LIB_PATH=$lib python tools/pack_time.py 1024 8 2>&1 | grep slab
  done
done
