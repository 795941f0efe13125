"""Time arvx_fast_carve against arvx_carve on the sphere scene (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ar_voxel_project_amd import capi, synthetic as syn

for N in ([int(a) for a in sys.argv[1:]] or [256, 512, 1024]):
    sc = syn.sphere_scene(N, 36)
    with capi.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        for name, fn in (("carve", ctx.carve), ("fast_carve", ctx.fast_carve)):
            ts = []
            for _ in range(4):
                ctx.reset()
                ctx.synchronize()
                t0 = time.perf_counter()
                fn()
                ctx.synchronize()
                ts.append(time.perf_counter() - t0)
            print(f"N={N} {name}: {min(ts)*1e3:.3f} ms", flush=True)
