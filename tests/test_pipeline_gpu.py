"""The whole device-side sequence of src/main.cpp:262-303 at 256^3 -- carve, colour,
handleUnseen, closure, marching-cubes cells -- against the oracle, and the greedy
carve at the same size: the multi-word, multi-tile, multi-block paths of every
secondary kernel on one realistic model."""
import numpy as np
import pytest

from tests import scenes

pytestmark = pytest.mark.gpu

N, V = 256, 8


@pytest.fixture(scope="module")
def scene():
    return scenes.syn.sphere_scene(N, V, with_images=True)


def test_pipeline_256(arvx, oracle, scene):
    sc = scene
    st = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
    model = oracle.color(N, N, N, sc.voxel_size, sc.M, sc.campos, sc.images, 1,
                         oracle.model_from_state(st))
    model = oracle.handle_unseen(st, model)
    with arvx.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks, campos=sc.campos)
        ctx.set_images(sc.images)
        ctx.carve()
        assert np.array_equal(ctx.download_state(), st)
        cells0 = ctx.mc_cells()
        ctx.color(arvx.COLOR_AVERAGE)
        idx, rgb = ctx.surface()
        assert np.array_equal(ctx.export_model(True), model)
        fidx, frgba = ctx.closure(3, True)
        cells1 = ctx.mc_cells()
        closed = ctx.export_model(True)
        # every list total reached the host through its page-locked word
        assert ctx.stats()["host_total_fallbacks"] == 0
    assert len(idx) > 50000 and np.all(np.diff(idx) > 0)
    assert np.array_equal(cells0, oracle.mc_cells(N, N, N, oracle.model_from_state(st)))
    want = oracle.closure(N, N, N, model)
    assert np.array_equal(closed, want)
    filled = np.nonzero((want[:, 3] != 0) & (model[:, 3] == 0))[0]
    assert np.array_equal(fidx, filled) and np.array_equal(frgba, want[filled])
    assert np.array_equal(cells1, oracle.mc_cells(N, N, N, want))


def test_fast_carve_256(arvx, oracle, scene):
    sc = scene
    want = oracle.fast_carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
    with arvx.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.fast_carve()
        got = ctx.download_state()
        ctx.reset()
        ctx.fast_carve()  # the cached work buffer is reused
        again = ctx.download_state()
    assert np.array_equal(got, want) and np.array_equal(again, want)


def test_pipeline_fuzz(arvx, oracle):
    """The same sequence on random small ragged grids with random cameras, noise masks and
    noise images: carve (or greedy carve), colour (either mode), handleUnseen, closure,
    marching-cubes cells -- every intermediate result equal to the oracle's."""
    rng = np.random.default_rng(99)
    wide = {57: (2112, 5, 9), 58: (6, 2100, 7), 59: (7, 5, 2090)}  # thousands of words / rows
    for i in range(60):
        X, Y, Z = (int(rng.integers(2, 48)) for _ in range(3))
        X, Y, Z = wide.get(i, (X, Y, Z))
        Vn = int(rng.integers(1, 7))
        W, H = int(rng.integers(16, 120)), int(rng.integers(16, 90))
        s = np.float32(0.512 / max(X, Y, Z))
        _, Rt, M = scenes.random_cameras(Vn, 0.512, seed=int(rng.integers(1 << 30)), W=W, H=H)
        # F11: "camera position" = translation column of the world->camera matrix
        campos = np.ascontiguousarray(Rt[:, :, 3], dtype=np.float32)
        masks = scenes.noise_masks(Vn, H, W, block=int(rng.choice([2, 6, 20])),
                                   p_bg=float(rng.uniform(0.3, 0.7)), seed=int(rng.integers(1 << 30)))
        images = rng.integers(0, 256, size=(Vn, H, W, 3), dtype=np.uint8)
        mode = int(rng.integers(0, 2))
        greedy = rng.random() < 0.3
        st = (oracle.fast_carve if greedy else oracle.carve)(X, Y, Z, s, M, masks)
        model = oracle.color(X, Y, Z, s, M, campos, images, mode, oracle.model_from_state(st))
        unseen = oracle.handle_unseen(st, model)
        closed = oracle.closure(X, Y, Z, unseen)
        with arvx.Context(X, Y, Z, s) as ctx:
            ctx.set_views(M, masks, campos=campos)
            ctx.set_images(images)
            if greedy:
                ctx.fast_carve()
            else:
                ctx.carve()
            what = f"case {i}: {X}x{Y}x{Z} V={Vn} mode={mode} greedy={greedy}"
            assert np.array_equal(ctx.download_state(), st), what
            ctx.color(mode)
            assert np.array_equal(ctx.export_model(False), model), what + " colours"
            assert np.array_equal(ctx.export_model(True), unseen), what + " handleUnseen"
            ctx.closure(3, True)
            assert np.array_equal(ctx.export_model(True), closed), what + " closure"
            assert np.array_equal(ctx.mc_cells(), oracle.mc_cells(X, Y, Z, closed)), what + " cells"
            assert ctx.stats()["host_total_fallbacks"] == 0, what
