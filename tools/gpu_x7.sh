#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/x7 -- python3 $GRAFT_REPO_ROOT/tools/fast_carve_time.py 1024 > $GRAFT_REPO_ROOT/gpurun_out/x7.log 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/x7 -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 $f | head -24
tail -3 $GRAFT_REPO_ROOT/gpurun_out/x7.log
