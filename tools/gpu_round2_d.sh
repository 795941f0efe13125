set -o pipefail
mkdir -p gpurun_out/r2d
(python tools/dbg_views.py | tail -2; timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2d/pytest.log)
tail -25 gpurun_out/r2d/pytest.log
