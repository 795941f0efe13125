"""Time arvx_pack_occupancy (a rank's planes in local order) on one striped rank of an N-rank job.
   python tools/pack_local_time.py [grid=1024] [world=8]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ar_voxel_project_amd import capi, synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = syn.sphere_scene(N, 36)
stream = torch.cuda.Stream()
with capi.Context(N, N, N, sc.voxel_size, stripes=(world, 3)) as ctx:
    ctx.set_stream(stream.cuda_stream)
    ctx.set_views(sc.M, sc.masks)
    with torch.cuda.stream(stream):
        words = torch.zeros(N * N * (N // world) // 64, dtype=torch.int64, device="cuda")
        ctx.carve()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        best = 1e9
        for _ in range(8):
            ev[0].record(stream)
            ctx.pack_occupancy(words.data_ptr())
            ev[1].record(stream)
            stream.synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]))
    print(f"N={N} world={world}: pack_occupancy {best:.4f} ms", flush=True)
