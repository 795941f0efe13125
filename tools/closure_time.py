"""Steady-state time of colour + closure on the carved sphere scene (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ar_voxel_project_amd import capi, synthetic as syn

for N in (256, 512, 1024):
    sc = syn.sphere_scene(N, 36, with_images=True)
    with capi.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks, campos=sc.campos)
        ctx.set_images(sc.images)
        for rep in range(3):
            ctx.reset(); ctx.carve(); ctx.synchronize()
            t0 = time.perf_counter(); ctx.color(capi.COLOR_AVERAGE); t1 = time.perf_counter()
            lib = ctx._lib
            rc = lib.arvx_closure(ctx._h, 3, 1); t2 = time.perf_counter()
            assert rc == 0
            idx, rgba = None, None
            n = capi.C.c_int64(); lib.arvx_closure_count(ctx._h, capi.C.byref(n))
            print(f"N={N} rep={rep} color {1e3*(t1-t0):.3f} ms  closure {1e3*(t2-t1):.3f} ms  filled {n.value}", flush=True)
