#!/bin/bash
# usage: prof_variant.sh <tag> ; env passes through
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/pv_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --extra-grid 0 --grid 512"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1 || exit 2
