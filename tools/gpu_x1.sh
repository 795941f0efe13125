#!/bin/bash
# scratch: event cost in the bench + wave timeline of the exact kernel
set -e
mkdir -p gpurun_out/x1
for e in all carve none; do
  ARVX_BENCH_EVENTS=$e python bench.py --steps 40 --warmup 5 --no-cpu --no-ablation --extra-grid 0 > gpurun_out/x1/bench_$e.json 2> gpurun_out/x1/bench_$e.err
done
ARVX_LIB_PATH=ab_libs/timeline.so python tools/wave_timeline.py 512 > gpurun_out/x1/tl512.json 2>&1
ARVX_LIB_PATH=ab_libs/timeline.so python tools/wave_timeline.py 1024 > gpurun_out/x1/tl1024.json 2>&1
python - <<'PY'
import json
for e in ("all","carve","none"):
    j=json.loads(open(f"gpurun_out/x1/bench_{e}.json").read().strip().splitlines()[-1])
    print(e, j["ms_per_step"], j["carve_kernel_ms"], j["views_kernel_ms"])
PY
