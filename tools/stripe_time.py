"""Carve time of one rank of an N-rank job on a 1024-class grid: contiguous Z slab vs striped
(8-plane groups r, r+N, ...).   python tools/stripe_time.py [grid=1024] [world=8]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ar_voxel_project_amd import capi, synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sc = syn.sphere_scene(N, 36)
stream = torch.cuda.Stream()


def timed(ctx):
    ctx.set_stream(stream.cuda_stream)
    ctx.set_views(sc.M, sc.masks)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    best = 1e9
    with torch.cuda.stream(stream):
        for _ in range(6):
            ctx.reset()
            ev[0].record(stream)
            ctx.carve()
            ev[1].record(stream)
            stream.synchronize()
            best = min(best, ev[0].elapsed_time(ev[1]))
    return best


for rank in range(world):
    z0, z1 = rank * N // world, (rank + 1) * N // world
    with capi.Context(N, N, N, sc.voxel_size, z_range=(z0, z1)) as ctx:
        a = timed(ctx)
    with capi.Context(N, N, N, sc.voxel_size, stripes=(world, rank)) as ctx:
        b = timed(ctx)
    print(f"N={N} world={world} rank {rank}: slab {a:.3f} ms   striped {b:.3f} ms", flush=True)
