"""Greedy carve (reference fastCarve, src/VoxelCarving.cpp:74-167) on the GPU vs
the oracle's literal queue-based restatement: state plane bit-exact."""
import numpy as np
import pytest

from tests import golden_io, scenes
from tests.test_carve_gpu import assert_same

pytestmark = pytest.mark.gpu


def gpu_fast(arvx, X, Y, Z, s, M, masks, state=None):
    with arvx.Context(X, Y, Z, s) as ctx:
        ctx.set_views(M, masks)
        if state is not None:
            ctx.upload_state(state)
        ctx.fast_carve()
        return ctx.download_state()


@pytest.mark.parametrize("dims,V", [((32, 32, 32), 6), ((100, 100, 50), 5), ((70, 9, 33), 4),
                                    ((130, 20, 20), 4), ((1, 1, 1), 3), ((10, 10, 5), 5)])
def test_fast_carve_sphere(arvx, oracle, dims, V):
    X, Y, Z = dims
    sc = scenes.small_sphere(32, V)
    s = np.float32(0.512 / max(dims))
    want = oracle.fast_carve(X, Y, Z, s, sc.M, sc.masks)
    got = gpu_fast(arvx, X, Y, Z, s, sc.M, sc.masks)
    assert_same(got, want, f"fast carve {dims}")


@pytest.mark.parametrize("block,p_bg", [(1, 0.6), (4, 0.5), (8, 0.45), (16, 0.7)])
def test_fast_carve_mazes(arvx, oracle, block, p_bg):
    """Noise masks carve maze-like regions: long winding fronts across many tiles."""
    N, V, W, H = 72, 3, 160, 120
    s = np.float32(0.512 / N)
    _, _, M = scenes.random_cameras(V, 0.512, seed=block, W=W, H=H)
    masks = scenes.noise_masks(V, H, W, block=block, p_bg=p_bg, seed=block + 1)
    want = oracle.fast_carve(N, N, N, s, M, masks)
    got = gpu_fast(arvx, N, N, N, s, M, masks)
    assert_same(got, want, f"maze block={block}")


def test_fast_carve_respects_previsited_walls(arvx, oracle):
    """Voxels already seen are 'visited' to the reference's BFS and block it."""
    N, V = 40, 5
    sc = scenes.small_sphere(N, V)
    st0 = np.full((N, N, N), 1, np.uint8)
    st0[:, :, 20] |= 2  # a seen wall across x = 20
    st0[5, 5, 20] = 1   # with one hole
    want = oracle.fast_carve(N, N, N, sc.voxel_size, sc.M, sc.masks, state=st0)
    got = gpu_fast(arvx, N, N, N, sc.voxel_size, sc.M, sc.masks, state=st0)
    assert_same(got, want, "walls")
    st1 = st0.copy()
    st1[0, 0, 0] |= 2  # origin already visited: the queue dies at once
    want = oracle.fast_carve(N, N, N, sc.voxel_size, sc.M, sc.masks, state=st1)
    assert np.array_equal(want, st1)
    assert_same(gpu_fast(arvx, N, N, N, sc.voxel_size, sc.M, sc.masks, state=st1), want, "dead")


@pytest.mark.parametrize("hole", [False, True])
def test_fast_carve_whole_tile_prepass_is_blocked_by_walls(arvx, oracle, hole):
    """Many 64x16x16 tiles are completely open here, on both sides of a seen wall: the
    whole-tile pre-pass may only take the side of the origin; the far side is reached
    through the hole (a partly open tile) or not at all."""
    X, Y, Z, V, W, H = 192, 48, 96, 4, 160, 120
    sc = scenes.small_sphere(32, V, W=W, H=H)
    masks = sc.masks.copy()
    masks[:, :, :] = 0  # background everywhere: every voxel some view sees is carvable
    s = np.float32(0.512 / X)
    st0 = np.full((Z, Y, X), 1, np.uint8)
    st0[40, :, :] |= 2  # wall: plane z = 40 already visited
    if hole:
        st0[40, 30, 100] = 1
    want = oracle.fast_carve(X, Y, Z, s, sc.M, masks, state=st0)
    got = gpu_fast(arvx, X, Y, Z, s, sc.M, masks, state=st0)
    assert_same(got, want, f"wall hole={hole}")
    beyond = (want[41:] & 1) == 0
    assert beyond.any() == hole  # the far side is carved only through the hole


def test_fast_carve_origin_not_carvable(arvx, oracle):
    N, V, W, H = 24, 3, 96, 72
    sc = scenes.small_sphere(N, V, W=W, H=H)
    masks = np.full((V, H, W), 255, np.uint8)  # everything foreground: only the seed is visited
    want = oracle.fast_carve(N, N, N, sc.voxel_size, sc.M, masks)
    assert (want & 2).sum() // 2 == 1
    assert_same(gpu_fast(arvx, N, N, N, sc.voxel_size, sc.M, masks), want, "seed only")


@pytest.mark.parametrize("name", golden_io.names())
def test_fast_carve_golden(arvx, name):
    g = golden_io.load(name)
    got = gpu_fast(arvx, g["X"], g["Y"], g["Z"], g["s"], g["M"], g["masks"])
    assert np.array_equal(got, g["fast_state"])


@pytest.mark.parametrize("dims", [(2112, 8, 9), (4160, 3, 5), (2050, 17, 8)])
def test_fast_carve_wide_grids(arvx, oracle, dims):
    """More than 32 tiles along x: the records <-> bit plane conversions take the tiles of a row
    in chunks; more than 64: no whole-tile pre-pass (its rows of tiles are 64-bit words)."""
    X, Y, Z = dims
    V, W, H = 3, 200, 60
    s = np.float32(0.512 / X)
    _, _, M = scenes.random_cameras(V, 0.512, seed=X, W=W, H=H)
    masks = scenes.noise_masks(V, H, W, block=6, p_bg=0.7, seed=X + 1)
    rng = np.random.default_rng(X)
    for state in (None, np.where(rng.random((Z, Y, X)) < 0.03, 3, 1).astype(np.uint8)):
        want = oracle.fast_carve(X, Y, Z, s, M, masks, state=state)
        assert_same(gpu_fast(arvx, X, Y, Z, s, M, masks, state=state), want, f"{dims}")


def test_dense_carve_after_fast_carve(arvx, oracle):
    """fastCarve leaves the model as records; a dense carve, a byte download and a second
    greedy carve on top of it start from that state."""
    N, V = 48, 4
    sc = scenes.small_sphere(N, V)
    rng = np.random.default_rng(5)
    state = np.where(rng.random((N, N, N)) < 0.04, 3, 1).astype(np.uint8)
    with arvx.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.upload_state(state)
        ctx.fast_carve()
        st = oracle.fast_carve(N, N, N, sc.voxel_size, sc.M, sc.masks, state=state)
        ctx.carve_views(0, 2)
        st = oracle.carve(N, N, N, sc.voxel_size, sc.M[:2], sc.masks[:2], state=st)
        assert_same(ctx.download_state(), st, "dense after greedy")
        ctx.fast_carve()
        st = oracle.fast_carve(N, N, N, sc.voxel_size, sc.M, sc.masks, state=st)
        assert_same(ctx.download_state(), st, "greedy after dense")


def test_fast_carve_rejects_slabs(arvx):
    sc = scenes.small_sphere(16, 3)
    with arvx.Context(16, 16, 16, sc.voxel_size, z_range=(0, 8)) as ctx:
        ctx.set_views(sc.M, sc.masks)
        with pytest.raises(arvx.ArvxError):
            ctx.fast_carve()


def test_fast_carve_fuzz(arvx, oracle):
    """Random small ragged grids, cameras and noise masks, fresh models and models with
    random pre-seen voxels (walls to the reference's BFS): GPU plane == oracle plane."""
    rng = np.random.default_rng(77)
    for i in range(120):
        X, Y, Z = (int(rng.integers(1, 72)) for _ in range(3))
        if rng.random() < 0.5:
            X = max(8, X // 8 * 8)  # the 8- and 16-voxel-wide apply / pack kernels
        V = int(rng.integers(1, 6))
        W, H = int(rng.integers(16, 160)), int(rng.integers(16, 120))
        s = np.float32(0.512 / max(X, Y, Z))
        _, _, M = scenes.random_cameras(V, 0.512, seed=int(rng.integers(1 << 30)), W=W, H=H)
        masks = scenes.noise_masks(V, H, W, block=int(rng.choice([1, 4, 12, 40])),
                                   p_bg=float(rng.uniform(0.4, 0.9)), seed=int(rng.integers(1 << 30)))
        state = None
        if rng.random() < 0.4:  # some voxels already seen (and a few already carved)
            state = np.ones((Z, Y, X), np.uint8)
            state[rng.random((Z, Y, X)) < 0.05] = 3
            state[rng.random((Z, Y, X)) < 0.01] = 2
        want = oracle.fast_carve(X, Y, Z, s, M, masks, state=state)
        got = gpu_fast(arvx, X, Y, Z, s, M, masks, state=state)
        assert_same(got, want, f"case {i}: {X}x{Y}x{Z} V={V} pre={'yes' if state is not None else 'no'}")
