set -e
cd $GRAFT_REPO_ROOT/tools/cpp
g++ -O2 -std=c++17 -ffp-contract=off -DARVX_HOST_TRACE -I$GRAFT_REPO_ROOT/include -o /tmp/dt arvx_dropin_time.cpp -L$GRAFT_REPO_ROOT/ar_voxel_project_amd/lib -larvx -Wl,-rpath,$GRAFT_REPO_ROOT/ar_voxel_project_amd/lib
cd $GRAFT_REPO_ROOT
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
/tmp/dt /tmp/scene.bin 512 512 512 0.001 3
