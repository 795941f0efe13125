#!/bin/bash
# A/B an environment switch of the current build on one box: tools/ab_flag.sh "<grids>" VAR=1
GRIDS=$1; FLAG=$2
for round in 1 2; do
  echo "== default"
  python tools/carve_stats.py $GRIDS 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/' | cut -c1-100
  python tools/pack_time.py 1024 8 2>&1 | grep slab
  echo "== $FLAG"
  env $FLAG python tools/carve_stats.py $GRIDS 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/' | cut -c1-100
  env $FLAG python tools/pack_time.py 1024 8 2>&1 | grep slab
done
