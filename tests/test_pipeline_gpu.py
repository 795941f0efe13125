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
