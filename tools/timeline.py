#!/usr/bin/env python3
"""Workgroup timeline of the fused carve kernel from a DIAGNOSTIC build
(-DARVX_TIMELINE, ab_libs/timeline.so): how many workgroups run over time, how long
they live, how the 8 XCDs are loaded.  Never part of a timed run.

    ARVX_LIB_PATH=ab_libs/timeline.so python tools/timeline.py 512
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ar_voxel_project_amd import capi, synthetic  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    sc = synthetic.sphere_scene(N, 36)
    lib = capi.load_library()
    with capi.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        for _ in range(3):
            ctx.reset()
            ctx.carve()
            ctx.synchronize()
        n = C.c_int64()
        lib.arvx_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        lib.arvx_debug_timeline(ctx._h, None, C.byref(n))
        buf = np.zeros((n.value, 4), np.uint64)
        lib.arvx_debug_timeline(ctx._h, buf.ctypes.data_as(C.c_void_p), C.byref(n))
    ok = buf[:, 0] > 0
    t0 = buf[ok, 0].min()
    start = (buf[ok, 0] - t0).astype(np.float64) / 100.0  # us (100 MHz)
    end = (buf[ok, 1] - t0).astype(np.float64) / 100.0
    dur = end - start
    xcc = buf[ok, 2].astype(int)
    total = end.max()
    edges = np.linspace(0, total, 21)
    active = [(int(((start <= t) & (end > t)).sum())) for t in edges[:-1] + total / 40]
    out = {"grid": N, "workgroups": int(ok.sum()), "kernel_us": float(total),
           "wg_duration_us": {"mean": float(dur.mean()), "p50": float(np.median(dur)),
                              "p99": float(np.percentile(dur, 99)), "max": float(dur.max())},
           "sum_wg_us": float(dur.sum()),
           "active_wgs_in_20_time_bins": active,
           "per_xcd": {int(x): {"wgs": int((xcc == x).sum()),
                                "sum_us": float(dur[xcc == x].sum()),
                                "last_end_us": float(end[xcc == x].max())}
                       for x in sorted(set(xcc))},
           "long_wgs_over_20us": int((dur > 20).sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
