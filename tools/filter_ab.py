#!/usr/bin/env python3
"""A/B of the fp32 projection filter (ARVX_CARVE_FILTER, csrc/carve_kernels.h filtered_view_blocks)
against the default exact kernel on one box: the sphere at 512^3 / 1024^3, bench.py's three
sparse-background workloads, 8x8 and 2x2 block noise.  Per workload: best-of-7 carve times with HIP
events, interleaved, and whether the two models are the same bit for bit.
    python tools/filter_ab.py [rounds [workload substring]]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the filter lives in the experiments build only (both sides of the A/B run on that build)
os.environ.setdefault("ARVX_LIB_PATH", os.path.join(ROOT, "ar_voxel_project_amd", "lib", "libarvx_experiments.so"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from ar_voxel_project_amd import capi, synthetic  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
only = sys.argv[2] if len(sys.argv) > 2 else ""  # a substring of the workloads to run
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)


def timed(ctx, flags):
    ctx.reset()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    ctx.carve(flags)
    b.record(stream)
    torch.cuda.synchronize()
    return a.elapsed_time(b)


def entry(name, N, M, masks, s):
    if only and only not in name:
        return None
    with capi.Context(N, N, N, s) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.set_views(M, masks)
        t = {0: [], capi.CARVE_FILTER: []}
        for _ in range(rounds):
            for f in t:
                t[f].append(timed(ctx, f))
        ctx.reset()
        ctx.carve(0)
        a = ctx.download_planes()
        a = (a[0].copy(), a[1].copy())
        ctx.reset()
        ctx.carve(capi.CARVE_FILTER)
        b = ctx.download_planes()
        same = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))
    e = {"workload": name, "exact_ms": min(t[0]), "filter_ms": min(t[capi.CARVE_FILTER]),
         "exact_median_ms": float(np.median(t[0])), "filter_median_ms": float(np.median(t[capi.CARVE_FILTER])),
         "filter_over_exact": min(t[capi.CARVE_FILTER]) / min(t[0]), "same_model": same}
    print(json.dumps(e), flush=True)
    return e


sc = synthetic.sphere_scene(512, 36)
entry("sphere 512^3 x 36", 512, sc.M, sc.masks, sc.voxel_size)
for name, blk, pbg in (("2x2 noise, 10 % background", 2, 0.10), ("2x2 noise, 2 % background", 2, 0.02),
                       ("0.5 % isolated background pixels", 1, 0.005)):
    entry(name, 512, sc.M, bench.sparse_noise_masks(36, sc.H, sc.W, blk, pbg), sc.voxel_size)
entry("8x8 block noise", 512, sc.M, bench.block_noise_masks(36, sc.H, sc.W, 8), sc.voxel_size)
entry("2x2 block noise", 512, sc.M, bench.block_noise_masks(36, sc.H, sc.W, 2), sc.voxel_size)
sc = synthetic.sphere_scene(1024, 36)
entry("sphere 1024^3 x 36", 1024, sc.M, sc.masks, sc.voxel_size)
