#!/usr/bin/env python3
"""Prints the carve kernel's culling statistics and cull / no-cull timings.
Usage: [ARVX_VIEWS=72] python tools/carve_stats.py [grid ...]   (GPU required)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ar_voxel_project_amd import capi, synthetic  # noqa: E402


def timed(ctx, stream, flags, reps):
    ms = []
    for _ in range(reps):
        ctx.reset()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ctx.carve(flags)
        b.record(stream)
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return float(np.median(ms)), float(np.min(ms))


def main():
    grids = [int(a) for a in sys.argv[1:] if a.isdigit()] or [512]
    V = int(os.environ.get("ARVX_VIEWS", "36"))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    for N in grids:
        sc = synthetic.sphere_scene(N, V)
        ctx = capi.Context(N, N, N, sc.voxel_size)
        ctx.set_stream(stream.cuda_stream)
        ctx.set_views(sc.M, sc.masks)
        ctx.reset()
        ctx.carve(capi.CARVE_STATS)
        st = ctx.stats()
        cull = timed(ctx, stream, 0, 7)
        fused_ms = timed(ctx, stream, getattr(capi, "CARVE_FUSED", 0), 7)
        reps = 3 if N >= 1024 else 5
        nocull = timed(ctx, stream, capi.CARVE_NO_CULL, reps)
        vv = N ** 3 * V
        print(json.dumps({
            "grid": N, "views": V, "stats": st,
            "mixed_pair_fraction": st["subtile_views_mixed"] / max(1, st["subtile_views_total"]),
            "carved_subtile_fraction": st["subtiles_carved"] / max(1, st["subtiles"]),
            "cull_ms_median_min": cull, "fused_ms_median_min": fused_ms,
            "nocull_ms_median_min": nocull,
            "cull_Mvvps": vv / cull[0] / 1e3, "nocull_Mvvps": vv / nocull[0] / 1e3}))
        ctx.close()


if __name__ == "__main__":
    main()
