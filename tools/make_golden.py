#!/usr/bin/env python3
"""Writes the small golden fixtures under tests/golden/.

The reference cannot be built or run in this image (OpenCV + Eigen absent) and
holds no unit-level golden vectors, so these fixtures are produced by the repo's
own CPU oracle (oracle/arvx_oracle.c) and pin *regressions* of oracle and
kernels, not the reference itself ("parity unpinned").  Each file
holds the complete inputs and the expected outputs, so nothing has to be
regenerated to use it.

    python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ar_voxel_project_amd import build, synthetic as syn  # noqa: E402
from tests import scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def pack_case(name, X, Y, Z, s, M, Rt, masks, images, po):
    M = np.ascontiguousarray(M, np.float32).reshape(-1, 3, 4)
    campos = syn.campos_from_rt(Rt)
    state = po.carve(X, Y, Z, s, M, masks, threads=1)
    fast = po.fast_carve(X, Y, Z, s, M, masks)
    model = po.model_from_state(state)
    closest = po.color(X, Y, Z, s, M, campos, images, 0, model)
    average = po.color(X, Y, Z, s, M, campos, images, 1, model)
    unseen = po.handle_unseen(state, average)
    closed = po.closure(X, Y, Z, unseen)
    m4 = masks if masks.ndim == 4 else masks[..., None]
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        dims=np.array([X, Y, Z], np.int32), voxel_size=np.float32(s), M=M,
        Rt=np.asarray(Rt, np.float32), campos=campos,
        mask_shape=np.array(m4.shape, np.int32),
        mask_nonzero_bits=np.packbits(m4 != 0),  # only ==0 / !=0 matters to the path
        images=images, state=state, fast_state=fast,
        closest_rgb=closest[:, :3].astype(np.uint8), average_rgb=average[:, :3].astype(np.uint8),
        closest_w=closest[:, 3].astype(np.uint8),
        final_rgba_after_unseen=unseen.astype(np.float32),
        closed_rgba=closed.astype(np.float32))
    print(name, "occupied", int((state & 1).sum()), "of", state.size,
          "fast carved", int(((fast & 1) == 0).sum()))


def main():
    build.build_oracle()
    from oracle import pyoracle as po
    os.makedirs(OUT, exist_ok=True)
    W, H = 96, 72
    # 1. sphere, ring cameras (SURVEY 8d shape, small)
    sc = syn.sphere_scene(32, 6, W=W, H=H, with_images=True)
    pack_case("sphere32", 32, 32, 32, sc.voxel_size, sc.M, sc.Rt, sc.masks, sc.images, po)
    # 2. ragged grid (reference benchmark "Medium": 50x50x25), box silhouettes
    sb = syn.box_scene((50, 50, 25), 5, W=W, H=H, with_images=True)
    pack_case("box50x50x25", 50, 50, 25, sb.voxel_size, sb.M, sb.Rt, sb.masks, sb.images, po)
    # 3. noise masks (3 channels) + random cameras, some inside the grid
    K32, Rt, M = scenes.random_cameras(5, 0.512, seed=11, W=W, H=H, inside=True)
    masks = scenes.noise_masks(5, H, W, C=3, block=4, seed=12)
    images = syn.pattern_images(5, W, H, seed=13)
    pack_case("noise24", 24, 24, 24, np.float32(0.512 / 24), M, Rt, masks, images, po)


if __name__ == "__main__":
    main()
