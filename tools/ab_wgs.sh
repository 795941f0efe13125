#!/bin/bash
# Persistent workgroups launched per CU for the exact kernel (experiments build: ARVX_EXACT_WGS_PER_CU),
# interleaved on one box: 512^3 and 1024^3 sphere, carve by HIP events (median / min of 7).
EXP=ar_voxel_project_amd/lib/libarvx_experiments.so
for round in 1 2; do
  for w in 4 5 6 8; do
    echo "== wgs $w"
    ARVX_EXACT_WGS_PER_CU=$w ARVX_LIB_PATH=$EXP python tools/carve_stats.py 512 1024 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/' | cut -c1-120
  done
done
