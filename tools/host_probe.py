#!/usr/bin/env python3
"""Probe: host enqueue time and total time per bench-like step over many back-to-back
repetitions (is the step GPU-bound all the time? where do host stalls come from?).
    python tools/host_probe.py <grid> <steps per rep> <reps> <mode: both|carve|views>
GPU required."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ar_voxel_project_amd import capi, synthetic  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
R = int(sys.argv[3]) if len(sys.argv) > 3 else 14
mode = sys.argv[4] if len(sys.argv) > 4 else "both"
sc = synthetic.sphere_scene(N, 36)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(s)
a = capi.Context(N, N, N, sc.voxel_size)
a.set_stream(s.cuda_stream)
d = torch.from_numpy(sc.masks).to(dev)
a.set_views_device(sc.M, d.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
steps_done = 0
tstart = time.perf_counter()
for rep in range(R):
    torch.cuda.synchronize()
    per = np.zeros(K)
    t0 = time.perf_counter()
    for i in range(K):
        t = time.perf_counter()
        a.reset()
        if mode != "carve":
            a.set_views_device(sc.M, d.data_ptr(), sc.W, sc.H, 1, campos=sc.campos)
        if mode != "views":
            a.carve()
        per[i] = time.perf_counter() - t
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    slow = np.flatnonzero(per > 1e-3)
    note = "".join(" | stall %.1f ms at step %d (t=%.0f ms)" % (per[j] * 1e3, steps_done + j,
                                                               (t0 - tstart) * 1e3) for j in slow)
    steps_done += K
    print("%s rep %2d: enqueue %.1f us/step (p50 %.1f max %.1f), total %.1f us/step%s" % (
        mode, rep, (t1 - t0) / K * 1e6, np.percentile(per, 50) * 1e6, per.max() * 1e6,
        (t2 - t0) / K * 1e6, note), flush=True)
