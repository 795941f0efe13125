"""Time arvx_mc_cells on the carved sphere scene (GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from ar_voxel_project_amd import capi, synthetic as syn

for N in (256, 512, 1024):
    sc = syn.sphere_scene(N, 36)
    with capi.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.carve()
        n = capi.C.c_int64()
        lib = ctx._lib
        lib.arvx_mc_cells(ctx._h, capi.C.byref(n))
        ctx.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            lib.arvx_mc_cells(ctx._h, capi.C.byref(n))
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        cells = ctx.mc_cells()
        t1 = time.perf_counter() - t0
        print(f"N={N} cells={n.value} of {(N+1)**3}  classify+compact {min(ts)*1e3:.3f} ms "
              f"(incl. download {t1*1e3:.3f} ms)", flush=True)
