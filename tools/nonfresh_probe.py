import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ar_voxel_project_amd import capi, synthetic
for N in (512, 1024):
    sc = synthetic.sphere_scene(N, 36)
    with capi.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        for name, prep in (("fresh", lambda: ctx.reset()), ("carved again", lambda: None)):
            ts = []
            for _ in range(12):
                prep()
                ctx.synchronize()
                t0 = time.perf_counter(); ctx.carve(0); ctx.synchronize(); ts.append(time.perf_counter() - t0)
            print(N, name, "carve wall ms", round(min(ts) * 1e3, 4), flush=True)
        # view by view on a fresh model
        ts = []
        for _ in range(5):
            ctx.reset(); ctx.synchronize()
            t0 = time.perf_counter()
            for v in range(36): ctx.carve_views(v, 1)
            ctx.synchronize(); ts.append(time.perf_counter() - t0)
        print(N, "36 single-view carves wall ms", round(min(ts) * 1e3, 4), flush=True)
