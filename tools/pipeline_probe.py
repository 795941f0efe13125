#!/usr/bin/env python3
"""Throughput of the step (views + carve) with 1, 2 and 3 jobs in flight: one context and one
stream per job slot, jobs dealt to the slots in turn (GPU required).
    python tools/pipeline_probe.py "512 1024" """
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ar_voxel_project_amd import capi, synthetic  # noqa: E402

grids = [int(g) for g in (sys.argv[1] if len(sys.argv) > 1 else "512").split()]
dev = torch.device("cuda", 0)
for N in grids:
    sc = synthetic.sphere_scene(N, 36)
    d_masks = torch.from_numpy(sc.masks).to(dev)
    for slots in [int(v) for v in os.environ.get("ARVX_PROBE_SLOTS", "1 2 3 1 2 3").split()]:
        streams = [torch.cuda.Stream(device=dev) for _ in range(slots)]
        ctxs = [capi.Context(N, N, N, sc.voxel_size) for _ in range(slots)]
        for c, st in zip(ctxs, streams):
            c.set_stream(st.cuda_stream)
        best = None
        for rep in range(4):
            K = 120
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                c = ctxs[k % slots]
                c.reset()
                c.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1)
                c.carve(0)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / K * 1e3
            best = ms if best is None or ms < best else best
        for c in ctxs:
            c.close()
        print(f"N={N} jobs in flight {slots}: {best:.4f} ms per step", flush=True)
