#!/usr/bin/env python3
"""Probe: how much of arvx_set_views_device hides behind arvx_carve when the two run on
different streams (two contexts, no dependency between them)?  GPU required."""
import gc
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ar_voxel_project_amd import capi, synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = synthetic.sphere_scene(N, 36)
dev = torch.device("cuda", 0)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
a = capi.Context(N, N, N, sc.voxel_size)
b = capi.Context(8, 8, 8, sc.voxel_size)
a.set_stream(sa.cuda_stream)
b.set_stream(sb.cuda_stream)
d_masks = torch.from_numpy(sc.masks).to(dev)
a.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
b.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)


def run(mode, steps=200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        a.reset()
        if mode == "serial":
            a.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
            a.carve()
        elif mode == "carve":
            a.carve()
        elif mode == "views":
            b.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
        else:  # overlap: the other context derives the same views on its own stream
            a.carve()
            b.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


gc.disable()
for _ in range(4):
    for mode in ("serial", "carve", "views", "overlap"):
        run(mode, 20)
        print(N, mode, round(run(mode), 4), "ms per step")
