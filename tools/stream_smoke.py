import os, sys, numpy as np
sys.path.insert(0, '.')
# the streaming carve lives in the experiments build only
os.environ.setdefault('ARVX_LIB_PATH', os.path.abspath('ar_voxel_project_amd/lib/libarvx_experiments.so'))
from ar_voxel_project_amd import capi, synthetic, build
build.build_oracle()
from oracle import pyoracle
for N, V in ((64, 8), (96, 36), (128, 70)):
    sc = synthetic.sphere_scene(N, V, W=320, H=240)
    with capi.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        for rep in range(3):
            ctx.reset()
            ctx.carve(capi.CARVE_STREAM)
            got = ctx.download_state()
        want = pyoracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
        print(N, V, "diff", int((got != want).sum()), flush=True)
