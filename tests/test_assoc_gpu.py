"""The grouping of the M*world row sums (csrc/arvx_device.h row_sum) is visible in
the kernels: on the KAT of tests/scenes.py::assoc_kat the default build must follow the
default oracle and the -DARVX_ASSOC_LEFT build the -DARVX_ORACLE_ASSOC_LEFT oracle, in every
kernel that projects voxels (split, brute force, fused, colour vote)."""
import os

import numpy as np
import pytest

from tests import scenes

pytestmark = pytest.mark.gpu


def _carve(arvx, lib_path, X, Y, Z, s, M, masks, flags):
    with arvx.Context(X, Y, Z, s, lib_path=lib_path) as ctx:
        ctx.set_views(M, masks)
        ctx.carve(flags)
        return ctx.download_state()


@pytest.mark.parametrize("N", [2, 8, 64])
@pytest.mark.parametrize("flags", [0, 1, 8])
def test_kernel_follows_its_grouping(arvx, oracle, N, flags):
    X, Y, Z, s, M, masks, (tx, ty, tz), st_default, st_left = scenes.assoc_kat(N)
    want = oracle.carve(X, Y, Z, s, M, masks)
    with oracle.variant("assoc_left"):
        want_left = oracle.carve(X, Y, Z, s, M, masks)
    assert want[tz, ty, tx] == st_default and want_left[tz, ty, tx] == st_left
    assert arvx.load_library().arvx_projection_assoc() == 0
    got = _carve(arvx, None, X, Y, Z, s, M, masks, flags)
    assert np.array_equal(got, want)
    if not os.path.exists(arvx.ASSOC_LEFT_LIB_PATH):
        pytest.fail("libarvx_assoc_left.so missing: run __graft_entry__.build()")
    assert arvx.load_library(arvx.ASSOC_LEFT_LIB_PATH).arvx_projection_assoc() == 1
    got_left = _carve(arvx, arvx.ASSOC_LEFT_LIB_PATH, X, Y, Z, s, M, masks, flags)
    assert np.array_equal(got_left, want_left)
    assert got[tz, ty, tx] != got_left[tz, ty, tx]


def test_colour_vote_follows_its_grouping(arvx, oracle):
    """The colour pass projects with the same row sums (src/ColorReconstruction.cpp:17-20)."""
    X, Y, Z, s, M, masks, (tx, ty, tz), _, _ = scenes.assoc_kat(4)
    H, W = masks.shape[1:]
    images = np.zeros((1, H, W, 3), np.uint8)
    images[0, :, :, 2] = np.arange(W)[None, :] * 10  # R encodes the pixel column
    campos = np.zeros((1, 3), np.float32)
    state = np.full((Z, Y, X), 3, np.uint8)
    state[tz, ty, tx + 1] = 2  # a carved neighbour: the KAT voxel is on the surface
    model = oracle.model_from_state(state)
    flat = tx + X * (ty + Y * tz)
    for lib_path, var in ((None, ""), (arvx.ASSOC_LEFT_LIB_PATH, "assoc_left")):
        with oracle.variant(var):
            want = oracle.color(X, Y, Z, s, M, campos, images, 1, model)
        with arvx.Context(X, Y, Z, s, lib_path=lib_path) as ctx:
            ctx.set_views(M, masks, campos)
            ctx.set_images(images)
            ctx.upload_state(state)
            ctx.color(1)
            idx, rgb = ctx.surface()
        got = model.copy()
        got[idx, :3] = rgb
        assert np.array_equal(got, want)
        assert got[flat, 0] == (40.0 if var == "" else 30.0)
