"""Time carve + arvx_pack_occupancy_global on one Z slab (GPU box); HIP events via torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ar_voxel_project_amd import capi, synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = syn.sphere_scene(N, 36)
stream = torch.cuda.Stream()
for rank in (0, world // 2):
    z0, z1 = rank * N // world, (rank + 1) * N // world
    with capi.Context(N, N, N, sc.voxel_size, z_range=(z0, z1)) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.set_views(sc.M, sc.masks)
        words = torch.zeros(N * N * N // 32, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        best = [1e9, 1e9]
        with torch.cuda.stream(stream):
            for _ in range(6):
                ctx.reset()
                ev[0].record(stream)
                ctx.carve()
                ev[1].record(stream)
                ctx.pack_occupancy_global(words.data_ptr())
                ev[2].record(stream)
                stream.synchronize()
                best[0] = min(best[0], ev[0].elapsed_time(ev[1]))
                best[1] = min(best[1], ev[1].elapsed_time(ev[2]))
        print(f"N={N} slab {z0}:{z1}  carve {best[0]:.3f} ms  pack {best[1]:.3f} ms", flush=True)
