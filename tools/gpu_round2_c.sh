set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r2c
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --extra-grid 0 > $OUT/stats.log 2>&1
f=$(ls $OUT/stats/*/*_kernel_stats.csv | head -1); cut -d, -f1-6 $f | head -12
