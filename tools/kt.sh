#!/bin/bash
# rocprofv3 kernel trace of a short sequential bench run; prints the arvx kernels' average us.
# usage (GPU box): tools/kt.sh <tag> [ARVX_LIB_PATH-or-""] [extra bench.py args]
TAG=$1; LIB=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/kt_$TAG; rm -rf $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$LIB" ] && export ARVX_LIB_PATH=$GRAFT_REPO_ROOT/$LIB
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --no-workloads --jobs 1 --jobs-in-flight 0 --no-e2e --rounds 1 --extra-grid 0 "$@" > $OUT.log 2>&1 || { tail -5 $OUT.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/*/*_kernel_stats.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "arvx::" in r["Name"]]
print("$TAG", {r["Name"].split("arvx::")[1][:28]: (int(r["Calls"]), round(float(r["AverageNs"])/1e3,1)) for r in rows})
PY
