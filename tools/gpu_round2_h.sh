set -o pipefail
mkdir -p gpurun_out/r2h
(timeout -k 10 600 python -m pytest tests/test_carve_gpu.py tests/test_assoc_gpu.py tests/test_fast_carve_gpu.py -m gpu -q -x > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2h/pytest.log)
timeout -k 10 300 bash tools/ab_compare.sh "512 1024" ar_voxel_project_amd/lib/libarvx.so ab_libs/prev.so 2>&1 | cut -c1-120
for i in 1 2; do
for L in ar_voxel_project_amd/lib/libarvx.so ab_libs/prev.so; do ARVX_LIB_PATH=$L python bench.py --steps 30 --warmup 5 --no-cpu --no-ablation --extra-grid 0 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', {k:round(d[k],4) for k in ('ms_per_step','carve_kernel_ms','views_kernel_ms')})"; done; done
