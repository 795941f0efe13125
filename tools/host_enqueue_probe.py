#!/usr/bin/env python3
"""How long the HOST takes to issue a step (reset + set_views_device + carve) against how long
the GPU takes to run it, with 1 and 4 jobs in flight: is the loop bound by launches?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ar_voxel_project_amd import capi, synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
sc = synthetic.sphere_scene(N, 36)
d_masks = torch.from_numpy(sc.masks).to(dev)
for slots in (1, 4):
    streams = [torch.cuda.Stream(device=dev) for _ in range(slots)]
    ctxs = [capi.Context(N, N, N, sc.voxel_size) for _ in range(slots)]
    for c, st in zip(ctxs, streams):
        c.set_stream(st.cuda_stream)
    for rep in range(3):
        K = 400
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            c = ctxs[k % slots]
            c.reset()
            c.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1)
            c.carve(0)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"N={N} jobs {slots}: host issues a step in {(t1 - t0) / K * 1e6:.1f} us, "
              f"the GPU finishes {(t2 - t1) * 1e6:.0f} us after the last issue, "
              f"{(t2 - t0) / K * 1e6:.1f} us per step in all", flush=True)
    for c in ctxs:
        c.close()
