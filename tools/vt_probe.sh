# kernel times of the view derivation for A/B builds of the table kernels (GPU box)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
ARVX_LIB_PATH=$GRAFT_REPO_ROOT/ab_libs/$v.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --no-workloads --jobs 1 --extra-grid 0 > $GRAFT_REPO_ROOT/gpurun_out/kt_$v.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/kt_$v/*/*_kernel_stats.csv")[0]
print("$v", {r["Name"][6:22]: round(float(r["AverageNs"])/1e3,1) for r in csv.DictReader(open(f)) if "arvx::" in r["Name"]})
PY
done
