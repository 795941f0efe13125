set -o pipefail
mkdir -p gpurun_out/r2e
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-ablation > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r2e/bench.json')); print({k:d[k] for k in ('value','ms_per_step','carve_kernel_ms','views_kernel_ms')}, d['extra'])"
cd /tmp && export TMPDIR=/tmp
for G in 512 1024; do
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r2e_$G
mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ablation --extra-grid 0 --grid $G > $OUT/stats.log 2>&1
f=$(ls $OUT/stats/*/*_kernel_stats.csv | head -1); cut -d, -f1-4 $f | head -9
done
