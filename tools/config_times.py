#!/usr/bin/env python3
"""The BASELINE.json configurations on one GPU, as far as this repository can run them:
   C1  box data set, 128^3, the 8 silhouette views the data set has    (plumbing inputs *)
   C2  human data set, 256^3, 8 of its 24 views (what the fixture holds) (plumbing inputs *)
   C3  synthetic sphere, 512^3 x 36 views                                (the bench configuration)
   H   synthetic sphere, 1024^3 x 36 views                               (north-star target)
   C4' synthetic sphere, 1024^3 x 72 views on ONE GPU                    (C4 is the 8-GPU run)
   C5' synthetic sphere, 512^3 x 36: carve + colour vote + MC cell list  (C5 with synthetic images)
(*) PIL-decoded masks of the reference's data sets with synthetic ring cameras: real, ragged
silhouettes, but not the reference's poses -- timings, not parity claims.
Each line: carve time by HIP events (best of 5) and voxel-views per second.  (Parity of the two
data-set lines with the CPU oracle is tests/test_carve_gpu.py::test_dataset_silhouettes_plumbing;
the oracle is test infrastructure and is not used here.)   Usage: python tools/config_times.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ar_voxel_project_amd import capi, synthetic  # noqa: E402
from tests import golden_io  # noqa: E402


def carve_ms(ctx, stream, reps=5):
    best = 1e9
    for _ in range(reps):
        ctx.reset()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        ctx.carve()
        b.record(stream)
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def main():
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    rows = [("C1 box 128^3 x 8 (data set masks)", 128, golden_io.dataset_masks("box"), False),
            ("C2 human 256^3 x 24 (data set masks, re-centred)", 256, golden_io.dataset_masks("human", recentre=True), False),
            ("C3 sphere 512^3 x 36", 512, 36, False),
            ("H  sphere 1024^3 x 36", 1024, 36, False),
            ("C4' sphere 1024^3 x 72, one GPU", 1024, 72, False),
            ("C5' sphere 512^3 x 36 + colour + MC cells", 512, 36, True)]
    for name, N, views, post in rows:
        masks = None if isinstance(views, int) else views
        V = views if masks is None else masks.shape[0]
        sc = synthetic.sphere_scene(N, V, with_images=post)
        if masks is None:
            masks = sc.masks
        with capi.Context(N, N, N, sc.voxel_size) as ctx:
            ctx.set_stream(stream.cuda_stream)
            ctx.set_views(sc.M, masks, campos=sc.campos if post else None)
            ms = carve_ms(ctx, stream)
            out = {"config": name, "carve_ms": round(ms, 4),
                   "Mvoxel_views_per_s": round(N ** 3 * V / ms / 1e3, 1)}
            if post:
                ctx.set_images(sc.images)
                t0 = time.perf_counter()
                ctx.color(capi.COLOR_AVERAGE)
                ctx.synchronize()
                out["color_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
                t0 = time.perf_counter()
                cells = ctx.mc_cells()
                out["mc_cells_ms"] = round((time.perf_counter() - t0) * 1e3, 3)
                out["mc_cells"] = int(len(cells))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
