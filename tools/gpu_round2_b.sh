set -o pipefail
mkdir -p gpurun_out/r2b
(timeout -k 10 300 python -m pytest tests/test_carve_gpu.py tests/test_assoc_gpu.py -m gpu -q -x > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2b/pytest.log)
tail -3 gpurun_out/r2b/pytest.log
timeout -k 10 300 bash tools/ab_compare.sh "512 1024" ar_voxel_project_amd/lib/libarvx.so ab_libs/div_scalar.so 2>&1 | cut -c1-200 | tee gpurun_out/r2b/ab.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu --no-ablation > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r2b/bench.json')); print({k:d[k] for k in ('value','ms_per_step','carve_kernel_ms','views_kernel_ms')}, d['extra'])"
