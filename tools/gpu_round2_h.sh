set -o pipefail
mkdir -p gpurun_out/r2h
(timeout -k 10 600 python -m pytest tests/test_carve_gpu.py tests/test_configs_gpu.py tests/test_fast_carve_gpu.py -m gpu -q -x > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2h/pytest.log)
for r in 1 2; do
echo "== block tests"; python tools/carve_stats.py 512 1024 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/' | cut -c1-110
echo "== none"; ARVX_NO_BLOCK_TESTS=1 python tools/carve_stats.py 512 1024 2>&1 | grep grid | sed 's/"stats".*"cull_ms/"cull_ms/' | cut -c1-110
done
