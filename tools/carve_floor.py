#!/usr/bin/env python3
"""Times the carve kernel on degenerate masks to separate its fixed costs:
all-background (every tile decided by the coarse pre-pass: pure fill), all-foreground
(nothing carved, no per-voxel work) and the sphere scene.  GPU required."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ar_voxel_project_amd import capi, synthetic  # noqa: E402
from tools.carve_stats import timed  # noqa: E402


def main():
    grids = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1024]
    V = 36
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    for N in grids:
        sc = synthetic.sphere_scene(N, V)
        for name, masks in (("all_background", np.zeros_like(sc.masks)),
                            ("all_foreground", np.full_like(sc.masks, 255)),
                            ("sphere", sc.masks)):
            ctx = capi.Context(N, N, N, sc.voxel_size)
            ctx.set_stream(stream.cuda_stream)
            ctx.set_views(sc.M, masks)
            t = timed(ctx, stream, 0, 7)
            ctx.reset()
            ctx.carve(capi.CARVE_STATS)
            print(json.dumps({"grid": N, "masks": name, "ms_median_min": t, "stats": ctx.stats()}))
            ctx.close()


if __name__ == "__main__":
    main()
