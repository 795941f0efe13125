#!/usr/bin/env python3
"""Fuzz: the split carve (rectangle tests, work lists, shared items, atomic merges) against the
brute-force kernel (ARVX_CARVE_NO_CULL: every voxel projected in every view) on random ragged
grids, random cameras (some inside the grid) and blocky noise masks -- two different code paths
over the same arithmetic; any difference is a bug in one of them.  Also from a pre-carved model
(second half of the views on top of the first half).
    python tools/fuzz_cull.py [cases=200] [seed=0]        (GPU required)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ar_voxel_project_amd import capi  # noqa: E402
from tests import scenes  # noqa: E402


def one_case(rng, i):
    big = rng.random() < 0.25
    hi = 320 if big else 96
    X, Y, Z = (int(rng.integers(4, hi)) for _ in range(3))
    if rng.random() < 0.5:
        X = max(4, X // 4 * 4)  # the 4-byte aligned kernels
    V = int(rng.integers(1, 41))
    W, H = int(rng.integers(16, 200)), int(rng.integers(16, 160))
    extent = 0.512
    s = np.float32(extent / max(X, Y, Z))
    _, _, M = scenes.random_cameras(V, extent, seed=int(rng.integers(1 << 30)), W=W, H=H,
                                    inside=rng.random() < 0.3)
    masks = scenes.noise_masks(V, H, W, C=int(rng.choice([1, 3])), p_bg=float(rng.uniform(0.2, 0.8)),
                               block=int(rng.choice([1, 4, 16, 48])), seed=int(rng.integers(1 << 30)))
    with capi.Context(X, Y, Z, s) as ctx:
        ctx.set_views(M, masks)
        ctx.carve(capi.CARVE_NO_CULL)
        want = ctx.download_state()
        ctx.reset()
        ctx.carve()
        got = ctx.download_state()
        h = V // 2
        ctx.reset()
        if h:
            ctx.carve_views(0, h)
        ctx.carve_views(h, V - h)
        got2 = ctx.download_state()
    ok = np.array_equal(got, want) and np.array_equal(got2, want)
    if not ok:
        print(f"case {i}: MISMATCH grid {X}x{Y}x{Z} V={V} image {W}x{H}: "
              f"{int((got != want).sum())} / {int((got2 != want).sum())} voxels differ", flush=True)
    return ok


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = sum(0 if one_case(rng, i) else 1 for i in range(cases))
    print(f"{cases} cases, {bad} mismatches", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
