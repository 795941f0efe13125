#!/usr/bin/env python3
"""36 single-view carves (arvx_carve_views(i, 1)) of a fresh model, for a kernel trace:
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/single_view_probe.py 512"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ar_voxel_project_amd import capi, synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = synthetic.sphere_scene(N, 36)
with capi.Context(N, N, N, sc.voxel_size) as ctx:
    ctx.set_views(sc.M, sc.masks)
    for _ in range(4):
        ctx.reset()
        for v in range(36):
            ctx.carve_views(v, 1)
        ctx.synchronize()
