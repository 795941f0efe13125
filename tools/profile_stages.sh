#!/bin/bash
# rocprofv3 kernel statistics of the stages around the carve (run on the GPU box):
#   the drop-in pipeline at 512^3 (tools/cpp/arvx_dropin_time) and the greedy carve at 512^3 / 1024^3
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_stages
mkdir -p $OUT
python3 - <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np
from ar_voxel_project_amd import synthetic
from tests.test_cpp_host import write_scene
sc = synthetic.sphere_scene(64, 36, with_images=True)
masks3 = np.repeat(sc.masks[..., None], 3, axis=-1)
write_scene("/tmp/scene.bin", 1, 1, 1, 1.0, sc.K, sc.Rt, masks3, sc.images, np.ones(1, np.uint8))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pipeline -- $GRAFT_REPO_ROOT/tools/cpp/arvx_dropin_time /tmp/scene.bin 512 512 512 0.001 4 > $OUT/pipeline.log 2>&1 || exit 1
for g in 512 1024; do
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fast_$g -- python3 $GRAFT_REPO_ROOT/tools/fast_carve_time.py $g > $OUT/fast_$g.log 2>&1 || exit 2
done
for d in pipeline fast_512 fast_1024; do cp $(find $OUT/$d -name "*kernel_stats.csv" | head -1) $OUT/${d}_kernel_stats.csv; done
grep "N=" $OUT/fast_512.log $OUT/fast_1024.log
