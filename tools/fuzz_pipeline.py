#!/usr/bin/env python3
"""A long random run of the stage sequence of src/main.cpp:262-303 against the oracle (GPU box;
tests/test_pipeline_gpu.py holds a short one): carve or greedy carve, colour vote (either mode),
the sample lists behind it, handleUnseen, closure, marching-cubes cells, on ragged grids up to
~100^3 and on long thin ones.
    python tools/fuzz_pipeline.py <seconds> [seed]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ar_voxel_project_amd import capi  # noqa: E402
from oracle import pyoracle as oracle  # noqa: E402
from tests import scenes  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
n = 0


def check(ok, what):
    if not ok:
        print("MISMATCH", what, flush=True)
        sys.exit(1)


while time.time() - t0 < budget:
    X, Y, Z = (int(v) for v in rng.integers(2, 100, 3))
    if rng.random() < 0.1:
        X = int(rng.integers(500, 2200))
        Y, Z = int(rng.integers(2, 12)), int(rng.integers(2, 12))
    Vn = int(rng.integers(1, 9))
    W, H = int(rng.integers(16, 200)), int(rng.integers(16, 150))
    s = np.float32(0.512 / max(X, Y, Z))
    _, Rt, M = scenes.random_cameras(Vn, 0.512, seed=int(rng.integers(1 << 30)), W=W, H=H)
    campos = np.ascontiguousarray(Rt[:, :, 3], dtype=np.float32)  # (F11)
    masks = scenes.noise_masks(Vn, H, W, block=int(rng.choice([2, 6, 20, 50])),
                               p_bg=float(rng.uniform(0.2, 0.8)), seed=int(rng.integers(1 << 30)))
    images = rng.integers(0, 256, size=(Vn, H, W, 3), dtype=np.uint8)
    mode = int(rng.integers(0, 2))
    greedy = rng.random() < 0.3
    ksize = 3  # (the oracle restates the reference's 3 x 3 x 3 closure)
    st = (oracle.fast_carve if greedy else oracle.carve)(X, Y, Z, s, M, masks)
    model = oracle.color(X, Y, Z, s, M, campos, images, mode, oracle.model_from_state(st))
    unseen = oracle.handle_unseen(st, model)
    closed = oracle.closure(X, Y, Z, unseen)
    what = f"case {n}: {X}x{Y}x{Z} V={Vn} {W}x{H} mode={mode} greedy={greedy} k={ksize}"
    with capi.Context(X, Y, Z, s) as ctx:
        ctx.set_views(M, masks, campos=campos)
        ctx.set_images(images)
        ctx.fast_carve() if greedy else ctx.carve()
        check(np.array_equal(ctx.download_state(), st), what + " state")
        ctx.color(mode)
        check(np.array_equal(ctx.export_model(False), model), what + " colours")
        idx, rgb = ctx.surface()
        if len(idx):  # the lists behind the vote, for a few of the coloured voxels
            pick = idx[rng.integers(0, len(idx), min(8, len(idx)))]
            smp = ctx.color_samples(pick)
            ok = smp["valid"].astype(bool)
            check(ok.any(axis=1).all(), what + " a coloured voxel without samples")
            if mode == 0:
                k = np.argmin(np.where(ok, smp["depth"], np.inf), axis=1)
                got = np.stack([smp["r"], smp["g"], smp["b"]], 2)[np.arange(len(pick)), k].astype(np.float32)
                check(np.array_equal(got, model.reshape(-1, 4)[pick, :3]), what + " closest sample")
        check(np.array_equal(ctx.export_model(True), unseen), what + " handleUnseen")
        ctx.closure(ksize, True)
        check(np.array_equal(ctx.export_model(True), closed), what + " closure")
        check(np.array_equal(ctx.mc_cells(), oracle.mc_cells(X, Y, Z, closed)), what + " cells")
    n += 1
    if n % 20 == 0:
        print(f"{n} cases ok ({time.time() - t0:.0f} s)", flush=True)
print(f"pipeline fuzz ok: {n} cases in {budget:.0f} s")
