set -o pipefail
mkdir -p gpurun_out/r2i
(timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r2i/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2i/pytest.log)
tail -12 gpurun_out/r2i/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2i/bench.json 2> gpurun_out/r2i/bench.err; echo "bench rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r2i/bench.json')); print({k:d[k] for k in ('value','ms_per_step','carve_kernel_ms','views_kernel_ms','parity_vs_oracle')}, d['extra'], d['roofline']['valu'], d['roofline']['physical_frac'])"
