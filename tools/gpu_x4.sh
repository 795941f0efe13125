#!/bin/bash
set -e
mkdir -p gpurun_out/x4
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from ar_voxel_project_amd import synthetic
from tests.test_cpp_host import write_scene
sc = synthetic.sphere_scene(64, 36, with_images=True)
masks3 = np.repeat(sc.masks[..., None], 3, axis=-1)
write_scene("/tmp/scene.bin", 1, 1, 1, 1.0, sc.K, sc.Rt, masks3, sc.images, np.ones(1, np.uint8))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/x4/prof -o dropin -- $GRAFT_REPO_ROOT/tools/cpp/arvx_dropin_time /tmp/scene.bin 512 512 512 0.001 4 > $GRAFT_REPO_ROOT/gpurun_out/x4/run.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/x4/prof -name "*stats*" | head
