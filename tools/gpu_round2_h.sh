set -o pipefail
mkdir -p gpurun_out/r2h
(timeout -k 10 600 python -m pytest tests/test_carve_gpu.py tests/test_assoc_gpu.py -m gpu -q -x > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2h/pytest.log)
timeout -k 10 300 bash tools/ab_compare.sh "512 1024" ar_voxel_project_amd/lib/libarvx.so ab_libs/prev.so 2>&1 | cut -c1-120
