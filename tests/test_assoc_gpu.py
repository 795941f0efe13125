"""The grouping of the M*world row sums (include/arvx/arvx.h, ARVX_ASSOC_*; csrc/arvx_device.h
row_sum) is a run-time property of a context and visible in the kernels: on the KAT of
tests/scenes.py::assoc_kat a context must follow the oracle under the same grouping, in every
kernel that projects voxels (split, brute force, fused, colour vote, the self-test hook)."""
import numpy as np
import pytest

from tests import scenes

pytestmark = pytest.mark.gpu

GROUPINGS = (("assoc_left", 1), ("assoc_right", 0))


def _carve(arvx, assoc, X, Y, Z, s, M, masks, flags):
    with arvx.Context(X, Y, Z, s, assoc=assoc) as ctx:
        assert ctx.assoc == assoc
        ctx.set_views(M, masks)
        ctx.carve(flags)
        return ctx.download_state()


def test_default_grouping_is_left(arvx, oracle):
    """Kernels, C oracle and numpy twin start with the same grouping (LEFT: the A*Bt branch of
    GEMMSingleMul sums `(s0 + s1 + s2 + s3) * alpha`)."""
    assert arvx.load_library().arvx_projection_assoc() == arvx.ASSOC_LEFT == oracle.assoc()
    with arvx.Context(4, 4, 4, 1.0) as ctx:
        assert ctx.assoc == arvx.ASSOC_LEFT


@pytest.mark.parametrize("N", [2, 8, 64])
@pytest.mark.parametrize("flags", [0, 1, 8])
def test_kernel_follows_its_grouping(arvx, oracle, N, flags):
    X, Y, Z, s, M, masks, (tx, ty, tz), st_right, st_left = scenes.assoc_kat(N)
    got = {}
    for name, assoc in GROUPINGS:
        with oracle.variant(name):
            want = oracle.carve(X, Y, Z, s, M, masks)
        assert want[tz, ty, tx] == (st_left if assoc else st_right)
        got[assoc] = _carve(arvx, assoc, X, Y, Z, s, M, masks, flags)
        assert np.array_equal(got[assoc], want), name
    assert got[0][tz, ty, tx] != got[1][tz, ty, tx]
    # one context, switched between carves
    with arvx.Context(X, Y, Z, s) as ctx:
        ctx.set_views(M, masks)
        for assoc in (0, 1, 0):
            ctx.set_assoc(assoc)
            ctx.reset()
            ctx.carve(flags)
            assert np.array_equal(ctx.download_state(), got[assoc])


def test_selftest_hooks_follow_the_oracle(arvx, oracle):
    """arvx_selftest_project / _depth: the raw rows, quotients and depths of the kernels equal
    the oracle's, on the KAT voxel (where the grouping shows) and on random voxels."""
    X, Y, Z, s, M, masks, (tx, ty, tz), _, _ = scenes.assoc_kat(8)
    rng = np.random.default_rng(5)
    xyz = np.vstack([[tx, ty, tz], rng.integers(0, 512, size=(200, 3))]).astype(np.int32)
    K32, Rt32, Ms = scenes.random_cameras(3, 0.512, seed=4)
    for name, assoc in GROUPINGS:
        with arvx.Context(X, Y, Z, s, assoc=assoc) as ctx, oracle.variant(name):
            rows, uv = ctx.selftest_project(M[0], s, xyz)
            for i, (x, y, z) in enumerate(xyz):
                raw = oracle.project_raw(M[0], s, int(x), int(y), int(z))
                assert np.array_equal(np.concatenate([rows[i], uv[i]]).view(np.uint32),
                                      np.asarray(raw, np.float32).view(np.uint32)), (name, i)
            assert rows[0, 0] == (np.float32(1.0) if assoc else np.float32(1.0) + np.float32(2.0 ** -23))
            for v in range(3):
                rows, uv = ctx.selftest_project(Ms[v], 0.001, xyz)
                for i, (x, y, z) in enumerate(xyz[:50]):
                    raw = oracle.project_raw(Ms[v], 0.001, int(x), int(y), int(z))
                    assert np.array_equal(np.concatenate([rows[i], uv[i]]).view(np.uint32),
                                          np.asarray(raw, np.float32).view(np.uint32))
    cam = np.array([0.3, -0.2, 0.7], np.float32)
    with arvx.Context(4, 4, 4, 1.0) as ctx:
        d = ctx.selftest_depth(cam, 0.001, xyz)
    for i, (x, y, z) in enumerate(xyz):
        assert d[i] == np.float32(oracle.depth(cam, 0.001, int(x), int(y), int(z)))


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
    for name, assoc in GROUPINGS:
        with oracle.variant(name):
            want = oracle.color(X, Y, Z, s, M, campos, images, 1, model)
        with arvx.Context(X, Y, Z, s, assoc=assoc) as ctx:
            ctx.set_views(M, masks, campos)
            ctx.set_images(images)
            ctx.upload_state(state)
            ctx.color(1)
            idx, rgb = ctx.surface()
        got = model.copy()
        got[idx, :3] = rgb
        assert np.array_equal(got, want)
        assert got[flat, 0] == (30.0 if assoc else 40.0)
