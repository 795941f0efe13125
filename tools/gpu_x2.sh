#!/bin/bash
# scratch: fused coarse+fill and bits+sums kernels: tests, then A/B timing
set -e
mkdir -p gpurun_out/x2
timeout -k 10 900 python -m pytest tests/test_carve_gpu.py tests/test_configs_gpu.py tests/test_assoc_gpu.py -m gpu -x -q > gpurun_out/x2/tests.log 2>&1 || { tail -30 gpurun_out/x2/tests.log; exit 1; }
tail -3 gpurun_out/x2/tests.log
for v in new coarse_split views_split; do
  case $v in new) e="";; coarse_split) e="ARVX_COARSE_SPLIT=1";; views_split) e="ARVX_VIEWS_SPLIT=1";; esac
  env $e ARVX_BENCH_EVENTS=none python bench.py --steps 60 --warmup 5 --no-cpu --no-ablation > gpurun_out/x2/bench_$v.json 2> gpurun_out/x2/bench_$v.err
done
python bench.py --steps 20 --warmup 5 --no-cpu --no-ablation --extra-grid 0 > gpurun_out/x2/bench_default.json 2> gpurun_out/x2/bench_default.err
python - <<'PY'
import json
for e in ("new","coarse_split","views_split","default"):
    j=json.loads(open(f"gpurun_out/x2/bench_{e}.json").read().strip().splitlines()[-1])
    print(e, j["ms_per_step"], j["carve_kernel_ms"], j["views_kernel_ms"], j.get("extra",{}).get("ms_per_step"))
PY
