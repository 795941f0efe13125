#!/usr/bin/env python3
"""Carve time of the current library per grid and row-sum grouping, and the same for other builds
on the same box, interleaved: python tools/carve_ab.py "512 1024" [libA.so libB.so+VAR=1 ...]
(`+VAR=value` after a library sets that environment variable for its runs: the switches of an
-DARVX_EXPERIMENTS build)
(GPU required; A/B within one process run, device-to-device spread is ~10 %)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(grids):
    import numpy as np
    import torch
    from ar_voxel_project_amd import capi, synthetic
    V = int(os.environ.get("ARVX_VIEWS", "36"))
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    for N in grids:
        sc = synthetic.sphere_scene(N, V)
        out = []
        for assoc in (1, 0):
            ctx = capi.Context(N, N, N, sc.voxel_size)
            if hasattr(ctx._lib, "arvx_ctx_set_projection_assoc"):
                ctx.set_assoc(assoc)
            elif assoc == 1:
                ctx.close()
                continue
            ctx.set_stream(stream.cuda_stream)
            ctx.set_views(sc.M, sc.masks)
            ms = []
            for _ in range(25):
                ctx.reset()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                ctx.carve(0)
                b.record(stream)
                torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            ctx.close()
            out.append(f"{'LEFT' if assoc else 'RIGHT'} median {np.median(ms[5:]):.4f} min {np.min(ms):.4f} ms")
        # the bench's step: derive the views from the resident masks, then carve
        import time
        ctx = capi.Context(N, N, N, sc.voxel_size)
        ctx.set_stream(stream.cuda_stream)
        d_masks = torch.from_numpy(sc.masks).to(dev)
        steps = []
        for rep in range(4):
            K = 100
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                ctx.reset()
                ctx.set_views_device(sc.M, d_masks.data_ptr(), sc.W, sc.H, 1)
                ctx.carve(0)
            torch.cuda.synchronize()
            steps.append((time.perf_counter() - t0) / K * 1e3)
        ctx.close()
        out.append(f"step (views + carve) {min(steps[1:]):.4f} ms")
        print(f"  N={N}: " + " | ".join(out), flush=True)


def main():
    if sys.argv[1] == "--child":
        child([int(g) for g in sys.argv[2].split()])
        return
    grids = sys.argv[1]
    libs = sys.argv[2:] or [""]
    for rnd in range(2):
        for lib in libs:
            env = dict(os.environ)
            path, *sets = lib.split("+")
            if path:
                env["ARVX_LIB_PATH"] = os.path.abspath(path)
            for kv in sets:
                k, _, v = kv.partition("=")
                env[k] = v
            print(f"== {lib or 'current build'} (round {rnd})", flush=True)
            subprocess.run([sys.executable, __file__, "--child", grids], env=env)


if __name__ == "__main__":
    main()
