set -o pipefail
mkdir -p gpurun_out/r2a
(timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2a/pytest.log)
tail -30 gpurun_out/r2a/pytest.log
