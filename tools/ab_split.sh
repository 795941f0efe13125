#!/bin/bash
# A/B of the small-grid exact kernel's occupancy on one box, interleaved (C1 128^3, C2 256^3):
#   split4.so   experiments build with -DARVX_EXACT_SPLIT_WAVES_PER_SIMD=4 (128 registers + 60 B scratch)
#   exp         libarvx_experiments.so as shipped (149 registers, 3 workgroups per CU)
#   exp+wgs4    the same kernel launched with 4 workgroups per CU (ARVX_EXACT_WGS_PER_CU=4)
EXP=ar_voxel_project_amd/lib/libarvx_experiments.so
for round in 1 2 3; do
  echo "== split4"; ARVX_LIB_PATH=ab_libs/split4.so python tools/config_times.py 2>&1 | grep -E "C1|C2" | cut -c1-120
  echo "== exp";    ARVX_LIB_PATH=$EXP python tools/config_times.py 2>&1 | grep -E "C1|C2" | cut -c1-120
  echo "== exp+wgs4"; ARVX_EXACT_WGS_PER_CU=4 ARVX_LIB_PATH=$EXP python tools/config_times.py 2>&1 | grep -E "C1|C2" | cut -c1-120
done
