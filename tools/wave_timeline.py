#!/usr/bin/env python3
"""Per-wave timeline of the persistent exact kernel from a DIAGNOSTIC build
(make EXTRA=-DARVX_TIMELINE): when each wave ends, how many items / views it took.
    ARVX_LIB_PATH=ab_libs/timeline.so python tools/wave_timeline.py 512 [z0:z1]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ar_voxel_project_amd import capi, synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
zr = tuple(int(v) for v in sys.argv[2].split(":")) if len(sys.argv) > 2 else None  # slab z0:z1
sc = synthetic.sphere_scene(N, 36)
lib = capi.load_library()
with capi.Context(N, N, N, sc.voxel_size, z_range=zr) as ctx:
    ctx.set_views(sc.M, sc.masks)
    for _ in range(3):
        ctx.reset()
        ctx.carve()
        ctx.synchronize()
    n = C.c_int64()
    lib.arvx_debug_timeline.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
    lib.arvx_debug_timeline(ctx._h, None, C.byref(n))
    buf = np.zeros((n.value, 8), np.uint64)
    lib.arvx_debug_timeline(ctx._h, buf.ctypes.data_as(C.c_void_p), C.byref(n))
ok = buf[:, 0] > 0
t0 = buf[ok, 0].min()
start = (buf[ok, 0] - t0).astype(np.float64) / 100.0
end = (buf[ok, 1] - t0).astype(np.float64) / 100.0
items, views = (buf[ok, 2] & np.uint64(0xffffffff)).astype(int), buf[ok, 3].astype(int)
first_item_us = (buf[ok, 2] >> np.uint64(32)).astype(np.float64) / 100.0
first_view_us = (buf[ok, 3] >> np.uint64(32)).astype(np.float64) / 100.0
views = (buf[ok, 3] & np.uint64(0xffffffff)).astype(int)
late = end > np.percentile(end, 90)
total = end.max()
few = np.nonzero(views <= 3)[0][:12]
many = np.nonzero(views >= 30)[0][:6]
sample = [{"views": int(views[i]), "items": int(items[i]), "start": float(start[i]),
           "first_view": float(first_view_us[i]), "first_item": float(first_item_us[i]),
           "end": float(end[i])} for i in list(few) + list(many)]
ph = np.stack([(buf[ok, 4] & np.uint64(0xffffffff)), (buf[ok, 4] >> np.uint64(32)),
               (buf[ok, 5] & np.uint64(0xffffffff)), (buf[ok, 5] >> np.uint64(32))], 1).astype(np.float64) / 100.0
pct = lambda a: {k: float(np.percentile(a, q)) for k, q in (("p1", 1), ("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))}
print(json.dumps({
    "grid": N, "waves": int(ok.sum()), "kernel_us": float(total),
    "wave_start_us": pct(start), "wave_end_us": pct(end),
    "busy_fraction": float((end - start).sum() / (total * ok.sum())),
    "items_per_wave": pct(items), "views_per_wave": pct(views),
    "first_view_end_us": pct(first_view_us), "first_item_end_us": pct(first_item_us),
    "late_waves(end>p90)": {"items": pct(items[late]), "views": pct(views[late])},
    "corr_end_views": float(np.corrcoef(end, views)[0, 1]),
    "end_us_by_views": {str(lo): [int(((views >= lo) & (views < hi)).sum()),
                                  float(np.median(end[(views >= lo) & (views < hi)]))
                                  if ((views >= lo) & (views < hi)).any() else None]
                        for lo, hi in ((0, 1), (1, 4), (4, 8), (8, 12), (12, 16), (16, 24), (24, 32), (32, 99))},
    "phase_us_per_item(pull+read, set-up, views, write-back)": [float(ph[:, k].sum() / max(1, items.sum())) for k in range(4)],
    "sample": sample,
    "items": int(items.sum()), "views": int(views.sum()),
    "us_per_view_mean": float((end - start).sum() / max(1, views.sum()))}))
