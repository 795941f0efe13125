"""Colour vote + model export on the GPU (through the C-ABI) against the oracle.
north_star asks for colour within 1 LSB; the bar used here is exact equality."""
import numpy as np
import pytest

from tests import golden_io, np_restate as npr, scenes

pytestmark = pytest.mark.gpu


def gpu_color(arvx, X, Y, Z, s, M, campos, masks, images, mode, z_range=None, state=None):
    with arvx.Context(X, Y, Z, s, z_range=z_range) as ctx:
        ctx.set_views(M, masks, campos=campos)
        ctx.set_images(images)
        if state is None:
            ctx.carve()
        else:
            ctx.upload_state(state)
        ctx.color(mode)
        idx, rgb = ctx.surface()
        depth = ctx.surface_depth()
        plain = ctx.export_model(False)
        unseen = ctx.export_model(True)
        st = ctx.download_state()
    return idx, rgb, depth, plain, unseen, st


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("dims,V", [((32, 32, 32), 6), ((50, 50, 25), 5), ((33, 17, 9), 4)])
def test_color_parity(arvx, oracle, dims, V, mode):
    X, Y, Z = dims
    sc = scenes.syn.sphere_scene(32, V, W=160, H=120, with_images=True)
    s = np.float32(0.512 / max(dims))
    st = oracle.carve(X, Y, Z, s, sc.M, sc.masks)
    model = oracle.model_from_state(st)
    want = oracle.color(X, Y, Z, s, sc.M, sc.campos, sc.images, mode, model)
    idx, rgb, depth, plain, unseen, gst = gpu_color(arvx, X, Y, Z, s, sc.M, sc.campos, sc.masks,
                                                    sc.images, mode)
    assert np.array_equal(gst, st)
    changed = np.nonzero((want != model).any(axis=1))[0]
    # a voxel whose voted colour equals MODEL_COLOR would not show up in `changed`
    assert set(changed) <= set(idx)
    assert np.array_equal(want[idx, :3], rgb), "surface colours"
    assert np.all(np.diff(idx) > 0), "ascending index order"
    assert np.array_equal(plain, want), "exported Model::voxels"
    assert np.array_equal(unseen, oracle.handle_unseen(st, want)), "after handleUnseen"
    assert len(idx) > 0


def test_min_depth_is_bit_exact(arvx, oracle):
    N, V = 32, 6
    sc = scenes.syn.sphere_scene(N, V, W=160, H=120, with_images=True)
    idx, _, depth, *_ = gpu_color(arvx, N, N, N, sc.voxel_size, sc.M, sc.campos, sc.masks,
                                  sc.images, 0)
    x, y, z = idx % N, (idx // N) % N, idx // (N * N)
    best = np.full(len(idx), np.inf, np.float32)
    for v in range(V):
        inside, _, _ = npr.project(sc.M[v], sc.voxel_size, x, y, z, 160, 120)
        d = npr.depth(sc.campos[v], sc.voxel_size, x, y, z)
        best = np.where(inside & (d < best), d, best)
    assert np.array_equal(depth, best)
    for k in range(0, len(idx), 97):  # and the C oracle agrees with numpy on samples
        dd = [oracle.depth(sc.campos[v], sc.voxel_size, x[k], y[k], z[k]) for v in range(V)
              if oracle.project(sc.M[v], sc.voxel_size, x[k], y[k], z[k], 160, 120)]
        assert np.float32(min(dd)) == depth[k]


@pytest.mark.parametrize("z_range", [None, (5, 21)])
def test_color_samples_are_the_lists_behind_the_vote(arvx, z_range):
    """arvx_color_samples: the per-view samples Model::addColor would collect (reference
    src/ColorReconstruction.h:44-60): validity, pixel and depth against the numpy restatement,
    sample for sample; and both votes (closest: strict <, first view wins; average: rounded
    mean) recomputed from the samples give the colours arvx_color voted."""
    N, V, W, H = 32, 7, 160, 120
    sc = scenes.syn.sphere_scene(N, V, W=W, H=H, with_images=True)
    z0 = z_range[0] if z_range else 0
    with arvx.Context(N, N, N, sc.voxel_size, z_range=z_range) as ctx:
        ctx.set_views(sc.M, sc.masks, campos=sc.campos)
        ctx.set_images(sc.images)
        ctx.carve()
        voted = {}
        for mode in (0, 1):
            ctx.color(mode)
            voted[mode] = ctx.surface()
        idx = voted[0][0]
        assert len(idx) > 50 and np.array_equal(idx, voted[1][0])
        # every coloured voxel, and a few that are not on the list (inside and far outside)
        extra = np.array([0, N * N * 3 + 17, (N // 2) * (1 + N) + N * N * 4], np.int64)
        ask = np.concatenate([idx, extra])
        smp = ctx.color_samples(ask)
        with pytest.raises(arvx.ArvxError):
            ctx.color_samples(np.array([N * N * N], np.int64))
    assert smp.shape == (len(ask), V)
    x, y, z = ask % N, (ask // N) % N, ask // (N * N) + z0
    for v in range(V):
        inside, px, py = npr.project(sc.M[v], sc.voxel_size, x, y, z, W, H)
        assert np.array_equal(smp["valid"][:, v].astype(bool), inside), f"view {v}: validity"
        bgr = sc.images[v][py, px]
        for name, ch in (("r", 2), ("g", 1), ("b", 0)):
            assert np.array_equal(smp[name][:, v][inside], bgr[:, ch][inside]), f"view {v}: {name}"
        d = npr.depth(sc.campos[v], sc.voxel_size, x, y, z)
        assert np.array_equal(smp["depth"][:, v][inside], d[inside]), f"view {v}: depth"
        assert not smp["depth"][:, v][~inside].any()
    s = smp[:len(idx)]
    ok = s["valid"].astype(bool)
    assert ok.any(axis=1).all()  # a coloured voxel has at least one sample
    rgb = np.stack([s["r"], s["g"], s["b"]], axis=2).astype(np.float32)
    first_min = np.argmin(np.where(ok, s["depth"], np.inf), axis=1)  # first of equal minima
    assert np.array_equal(rgb[np.arange(len(idx)), first_min], voted[0][1]), "closest colour"
    n = ok.sum(axis=1).astype(np.float32)
    mean = (np.where(ok[..., None], rgb, 0).sum(axis=1, dtype=np.float32) / n[:, None]).astype(np.float32)
    assert np.array_equal(npr.round_half_away(mean).astype(np.float32), voted[1][1]), "average colour"


@pytest.mark.parametrize("mode", [0, 1])
def test_color_on_z_slabs_uses_halo(arvx, oracle, mode):
    """isInner needs the plane above and below a slab: recomputed locally as halo."""
    N, V = 40, 6
    sc = scenes.syn.sphere_scene(N, V, W=160, H=120, with_images=True)
    st = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
    want = oracle.color(N, N, N, sc.voxel_size, sc.M, sc.campos, sc.images, mode,
                        oracle.model_from_state(st))
    parts = []
    for r in [(0, 9), (9, 10), (10, 31), (31, 40)]:
        _, _, _, plain, _, _ = gpu_color(arvx, N, N, N, sc.voxel_size, sc.M, sc.campos, sc.masks,
                                         sc.images, mode, z_range=r)
        parts.append(plain)
    assert np.array_equal(np.concatenate(parts), want)


@pytest.mark.parametrize("dims", [(21, 13, 11), (136, 12, 9), (130, 10, 7), (64, 8, 8), (8, 8, 8)])
def test_color_of_uploaded_model_with_random_occupancy(arvx, oracle, dims):
    """Random occupancy makes almost every voxel a surface voxel with odd neighbours;
    rows of one, two and three 64-bit words of the bit plane, with and without the
    8-voxels-per-thread pack (X % 8 == 0)."""
    X, Y, Z = dims
    V = 4
    sc = scenes.syn.sphere_scene(32, V, W=96, H=72, with_images=True)
    s = np.float32(0.512 / X)
    rng = np.random.default_rng(9)
    st0 = rng.choice(np.array([0, 1, 2, 3], np.uint8), p=[0.1, 0.3, 0.2, 0.4], size=(Z, Y, X))
    model = oracle.model_from_state(st0)
    for mode in (0, 1):
        want = oracle.color(X, Y, Z, s, sc.M, sc.campos, sc.images, mode, model)
        _, _, _, plain, unseen, _ = gpu_color(arvx, X, Y, Z, s, sc.M, sc.campos, sc.masks,
                                              sc.images, mode, state=st0)
        assert np.array_equal(plain, want)
        assert np.array_equal(unseen, oracle.handle_unseen(st0, want))


@pytest.mark.parametrize("name", golden_io.names())
def test_color_matches_golden(arvx, name):
    g = golden_io.load(name)
    for mode, key in ((0, "closest_rgb"), (1, "average_rgb")):
        _, _, _, plain, unseen, _ = gpu_color(arvx, g["X"], g["Y"], g["Z"], g["s"], g["M"],
                                              g["campos"], g["masks"], g["images"], mode)
        assert np.array_equal(plain[:, :3], g[key].astype(np.float32))
        if mode == 1:
            assert np.array_equal(unseen, g["final_rgba_after_unseen"])


def test_color_call_order_errors(arvx):
    sc = scenes.syn.sphere_scene(16, 3, W=96, H=72, with_images=True)
    with arvx.Context(16, 16, 16, sc.voxel_size) as ctx:
        one = np.array([5], np.int64)
        with pytest.raises(arvx.ArvxError):
            ctx.selftest_view_tables(0, 96, 72)  # no views yet
        ctx.set_views(sc.M, sc.masks)  # no campos
        with pytest.raises(arvx.ArvxError):
            ctx.selftest_view_tables(3, 96, 72)  # view 3 of 3
        with pytest.raises(arvx.ArvxError):
            ctx.color(0)  # no images
        with pytest.raises(arvx.ArvxError):
            ctx.color_samples(one)  # no images
        ctx.set_images(sc.images)
        with pytest.raises(arvx.ArvxError):
            ctx.color(0)  # no campos
        with pytest.raises(arvx.ArvxError):
            ctx.color_samples(one)  # no campos
        ctx.set_views(sc.M, sc.masks, campos=sc.campos)
        ctx.set_images(sc.images)
        ctx.carve()
        with pytest.raises(arvx.ArvxError):
            ctx.color(7)
        ctx.color(1)
        assert len(ctx.surface()[0]) > 0
        ctx.carve()  # state changed: colour result is stale
        with pytest.raises(arvx.ArvxError):
            ctx.surface()
        assert ctx.color_samples(one).shape == (1, 3)  # (the samples do not depend on the state)
        assert ctx.color_samples(np.zeros(0, np.int64)).shape == (0, 3)
        ctx.set_views(sc.M, sc.masks, campos=sc.campos)  # new views drop the images
        with pytest.raises(arvx.ArvxError):
            ctx.color_samples(one)


@pytest.mark.parametrize("mode", [0, 1])
def test_views_scaled_by_powers_of_two(arvx, oracle, mode):
    """Rows of M * world far outside the range in which two quotients may share one reciprocal
    (divide2_shared_rcp, csrc/arvx_device.h: 2^-60 <= |a2| <= 2^60): every second view's matrix is
    scaled by 2^-70 or 2^+70.  The scaling is exact and cancels in a0 / a2, a1 / a2, so the pixels are
    those of the unscaled views -- through the IEEE divide the carve and the colour vote fall back
    to; the oracle divides in the reference's way throughout."""
    X, Y, Z, V = 40, 36, 28, 6
    sc = scenes.syn.sphere_scene(40, V, W=160, H=120, with_images=True)
    s = np.float32(0.512 / 40)
    M = sc.M.copy()
    M[1] *= np.float32(2.0 ** -70)
    M[3] *= np.float32(2.0 ** 70)
    M[5] *= np.float32(2.0 ** -70)
    st = oracle.carve(X, Y, Z, s, M, sc.masks)
    assert np.array_equal(st, oracle.carve(X, Y, Z, s, sc.M, sc.masks)), "the scaling cancels"
    model = oracle.model_from_state(st)
    want = oracle.color(X, Y, Z, s, M, sc.campos, sc.images, mode, model)
    idx, rgb, depth, plain, unseen, gst = gpu_color(arvx, X, Y, Z, s, M, sc.campos, sc.masks,
                                                    sc.images, mode)
    assert np.array_equal(gst, st), "carve"
    assert np.array_equal(want[idx, :3], rgb), "surface colours"
    assert np.array_equal(plain, want), "exported Model::voxels"
