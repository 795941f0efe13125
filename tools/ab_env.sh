#!/bin/bash
# A/B on one box, interleaved: this build with the block map, with the row map
# (ARVX_EXACT_ROWS=1), and optionally another build.   usage: tools/ab_env.sh "<grids>" [other.so]
for round in 1 2 3; do
  echo "== blocks"; python tools/carve_stats.py $1 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/' | cut -c1-100
  echo "== rows";   ARVX_EXACT_ROWS=1 python tools/carve_stats.py $1 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/' | cut -c1-100
  if [ -n "$2" ]; then echo "== $2"; ARVX_LIB_PATH=$2 python tools/carve_stats.py $1 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/' | cut -c1-100; fi
done
