#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$(basename $lib .so); rm -rf $OUT
ARVX_LIB_PATH=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --no-workloads --jobs 1 --extra-grid 0 --grid 1024 > $OUT.log 2>&1
done
