#!/bin/bash
# SQ counter passes over one bench configuration (GPU box).  usage: prof_variant.sh <tag> [grid]
TAG=$1; GRID=${2:-512}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pv_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --jobs 1 --extra-grid 0 --grid $GRID"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1 || exit 2
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1 || exit 3
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 --kernel-trace --output-format csv -d $OUT/sq3 -- $BENCH > $OUT/sq3.log 2>&1 || exit 4
