#!/bin/bash
# Collects the rocprofv3 evidence for the carve kernel (run on the GPU box via gpurun).
# usage: tools/profile.sh <tag> <grid> [extra bench args]
set -o pipefail
TAG=$1; GRID=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --no-workloads --jobs 1 --jobs-in-flight 0 --no-e2e --rounds 1 --extra-grid 0 --grid $GRID $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $BENCH > $OUT/fetch.log 2>&1 || exit 2
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $BENCH > $OUT/write.log 2>&1 || exit 3
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq -- $BENCH > $OUT/sq.log 2>&1 || exit 4
rocprofv3 --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/l2 -- $BENCH > $OUT/l2.log 2>&1 || exit 5
tail -1 $OUT/stats.log
