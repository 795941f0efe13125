set -o pipefail
cd /tmp && export TMPDIR=/tmp
for S in 0 1; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r2g_$S
mkdir -p $OUT
ARVX_SAT_SHIFT=$S timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --extra-grid 0 > $OUT/stats.log 2>&1
f=$(ls $OUT/stats/*/*_kernel_stats.csv | head -1); cut -d, -f1-4 $f | cut -c1-150 | head -12
done
