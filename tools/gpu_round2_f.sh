set -o pipefail
mkdir -p gpurun_out/r2f
for S in 0 1; do
(ARVX_SAT_SHIFT=$S timeout -k 10 400 python -m pytest tests/test_carve_gpu.py -m gpu -q -x > gpurun_out/r2f/pytest_$S.log 2>&1; echo "shift $S pytest rc $?"; tail -2 gpurun_out/r2f/pytest_$S.log)
done
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r2f
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --extra-grid 0 > $OUT/stats.log 2>&1
tail -c 600 $OUT/stats.log
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/stats/runc/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'views' in r['Name'] or 'carve' in r['Name']: print(r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1000,1))
PY
