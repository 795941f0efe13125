"""Morphological closure (reference applyClosure, src/Postprocessing3d.cpp:4-100)
on the GPU vs the oracle's literal restatement: the whole RGBA model, exactly."""
import numpy as np
import pytest

from tests import golden_io, scenes

pytestmark = pytest.mark.gpu


def pipeline(arvx, X, Y, Z, s, M, campos, masks, images, mode, unseen, ksize=3):
    with arvx.Context(X, Y, Z, s) as ctx:
        ctx.set_views(M, masks, campos=campos)
        ctx.set_images(images)
        ctx.carve()
        if mode is not None:
            ctx.color(mode)
        idx, rgba = ctx.closure(ksize, unseen)
        out = ctx.export_model(unseen)
        st = ctx.download_state()
        with pytest.raises(arvx.ArvxError):
            ctx.closure(ksize, unseen)  # already applied
        with pytest.raises(arvx.ArvxError):
            ctx.export_model(not unseen)  # inconsistent flag
    return idx, rgba, out, st


@pytest.mark.parametrize("mode,unseen", [(1, True), (0, True), (1, False), (None, True)])
@pytest.mark.parametrize("dims", [(32, 32, 32), (50, 50, 25), (21, 13, 11)])
def test_closure_parity(arvx, oracle, dims, mode, unseen):
    X, Y, Z = dims
    V = 5
    sc = scenes.syn.sphere_scene(32, V, W=160, H=120, with_images=True)
    s = np.float32(0.512 / max(dims))
    st = oracle.carve(X, Y, Z, s, sc.M, sc.masks)
    model = oracle.model_from_state(st)
    if mode is not None:
        model = oracle.color(X, Y, Z, s, sc.M, sc.campos, sc.images, mode, model)
    if unseen:
        model = oracle.handle_unseen(st, model)
    want = oracle.closure(X, Y, Z, model)
    idx, rgba, out, gst = pipeline(arvx, X, Y, Z, s, sc.M, sc.campos, sc.masks, sc.images, mode,
                                   unseen)
    assert np.array_equal(out, want)
    filled = np.nonzero((want[:, 3] != 0) & (model[:, 3] == 0))[0]
    assert np.array_equal(idx, filled) and len(idx) > 0
    assert np.array_equal(rgba, want[idx])
    occ_want = want[:, 3] != 0
    occ_got = ((gst.reshape(-1) & 1) == 1) | (unseen & ((gst.reshape(-1) & 2) == 0))
    assert np.array_equal(occ_got, occ_want)


@pytest.mark.parametrize("dims", [(136, 12, 9), (130, 10, 7), (64, 8, 8), (200, 6, 5)])
@pytest.mark.parametrize("ksize", [3, 5, 9])
def test_closure_multiword_rows_random_occupancy(arvx, oracle, dims, ksize):
    """Sparse random occupancy on rows of one to four 64-bit words: the box dilation has
    to carry across word borders and stop at the row ends, for every kernel size."""
    X, Y, Z = dims
    rng = np.random.default_rng(X + ksize)
    st0 = np.where(rng.random((Z, Y, X)) < 0.01, 3, 2).astype(np.uint8)
    st0[Z // 2, Y // 2, 63 if X > 64 else X - 1] = 3  # one voxel right at a word / row border
    model = oracle.model_from_state(st0)
    want = closure_any_kernel(model, X, Y, Z, ksize)
    with arvx.Context(X, Y, Z, 0.01) as ctx:
        ctx.upload_state(st0)
        idx, rgba = ctx.closure(ksize, False)
        out = ctx.export_model(False)
    assert np.array_equal(out, want)
    filled = np.nonzero((want[:, 3] != 0) & (model[:, 3] == 0))[0]
    assert np.array_equal(idx, filled) and len(idx) > 0


def closure_any_kernel(model, X, Y, Z, ksize):
    """numpy restatement of applyClosure for any odd kernel size (the oracle's C
    version is the literal 3x3x3 one): mean RGBA of the occupied neighbours, summed in
    the reference's x, y, z offset order in fp32."""
    r = (ksize - 1) // 2
    m = model.reshape(Z, Y, X, 4)
    occ = m[..., 3] != 0
    out = m.copy()
    pad = np.zeros((Z + 2 * r, Y + 2 * r, X + 2 * r, 4), np.float32)
    pad[r:r + Z, r:r + Y, r:r + X] = np.where(occ[..., None], m, 0)
    pocc = np.zeros((Z + 2 * r, Y + 2 * r, X + 2 * r), np.float32)
    pocc[r:r + Z, r:r + Y, r:r + X] = occ
    acc = np.zeros((Z, Y, X, 4), np.float32)
    cnt = np.zeros((Z, Y, X), np.float32)
    for a in range(-r, r + 1):          # x offset outermost (src/Postprocessing3d.cpp:31-48)
        for b in range(-r, r + 1):
            for c in range(-r, r + 1):
                sl = (slice(r + c, r + c + Z), slice(r + b, r + b + Y), slice(r + a, r + a + X))
                acc = (acc + pad[sl]).astype(np.float32)
                cnt += pocc[sl]
    fill = (~occ) & (cnt > 0)
    out[fill] = (acc[fill] / cnt[fill][:, None]).astype(np.float32)
    return out.reshape(-1, 4)


def test_closure_kernel_5_and_uploaded_colors(arvx, oracle):
    """Bigger box, and a model whose colours come from the caller (arvx_colors_upload)."""
    X, Y, Z = 18, 15, 12
    rng = np.random.default_rng(4)
    st0 = np.where(rng.random((Z, Y, X)) < 0.08, 3, 2).astype(np.uint8)
    model = oracle.model_from_state(st0)
    occ = np.nonzero(model[:, 3] != 0)[0]
    pick = occ[rng.random(len(occ)) < 0.5]
    model[pick, :3] = rng.integers(0, 256, size=(len(pick), 3)).astype(np.float32)
    for k, radius_fn in ((5, None), (3, None), (1, None)):
        want = model.copy()
        if k == 3:
            want = oracle.closure(X, Y, Z, model)
        with arvx.Context(X, Y, Z, 0.01) as ctx:
            ctx.upload_state(st0)
            ctx.upload_colors(pick, model[pick, :3])
            idx, rgba = ctx.closure(k, False)
            out = ctx.export_model(False)
        if k == 3:
            assert np.array_equal(out, want)
        elif k == 1:
            assert len(idx) == 0 and np.array_equal(out, model)
        else:
            assert len(idx) > 0 and np.all(out[idx, 3] == 1.0)
            assert np.array_equal(out[occ], model[occ])


@pytest.mark.parametrize("name", golden_io.names())
def test_closure_matches_golden(arvx, name):
    g = golden_io.load(name)
    _, _, out, _ = pipeline(arvx, g["X"], g["Y"], g["Z"], g["s"], g["M"], g["campos"], g["masks"],
                            g["images"], 1, True)
    assert np.array_equal(out, g["closed_rgba"])


@pytest.mark.parametrize("seed", range(8))
def test_closure_fills_into_tiles_that_exist_only_as_a_code(arvx, oracle, seed):
    """Grids of whole tiles (Y, Z multiples of 8), fresh carve: most coarse tiles exist only as their
    code when the closure runs, and some of the voxels it fills lie in tiles coded "carved".  Those
    tiles -- and only those -- are written out by rec_or_bitgrid_lazy_kernel; everything that reads
    the state afterwards (export, planes, packets, the cell list, a second carve of the same model,
    handleUnseen) must see records and codes agree."""
    rng = np.random.default_rng(700 + seed)
    X = int(rng.choice([64, 96, 128, 200]))
    Y = 8 * int(rng.integers(4, 20))
    Z = 8 * int(rng.integers(4, 20))
    V = int(rng.integers(2, 7))
    W, H = 160, 120
    s = np.float32(0.512 / max(X, Y, Z))
    _, Rt, M = scenes.random_cameras(V, 0.512, seed=seed + 31, W=W, H=H)
    campos = np.ascontiguousarray(Rt[:, :, 3], dtype=np.float32)
    masks = scenes.noise_masks(V, H, W, block=int(rng.choice([12, 24, 48])), p_bg=float(rng.uniform(0.3, 0.6)),
                               seed=seed + 5)
    images = rng.integers(0, 256, size=(V, H, W, 3), dtype=np.uint8)
    unseen = bool(seed % 2)
    st = oracle.carve(X, Y, Z, s, M, masks)
    model = oracle.color(X, Y, Z, s, M, campos, images, 1, oracle.model_from_state(st))
    if unseen:
        model = oracle.handle_unseen(st, model)
    want = oracle.closure(X, Y, Z, model)
    with arvx.Context(X, Y, Z, s) as ctx:
        ctx.set_views(M, masks, campos=campos)
        ctx.set_images(images)
        ctx.carve()
        ctx.color(arvx.COLOR_AVERAGE)
        if unseen and seed % 4 == 1:
            ctx.handle_unseen()  # (the closure then starts from the colour pass's planes)
        ctx.closure(3, unseen)
        assert np.array_equal(ctx.mc_cells(), oracle.mc_cells(X, Y, Z, want))
        occ_want = (want[:, 3] != 0).reshape(Z, Y, X)
        if X % 32 == 0:
            po, ps, no, ns = ctx.download_packets()
        occ, seen = ctx.download_planes()
        got = np.unpackbits(occ.view(np.uint8), bitorder="little").reshape(Z, Y, -1)[:, :, :X].astype(bool)
        handled = unseen and seed % 4 == 1  # handleUnseen ran on the device: unseen voxels ARE occupied
        st_after = ctx.download_state()
        occ_state = ((st_after & 1) == 1) | ((not handled) & unseen & ((st_after & 2) == 0))
        assert np.array_equal(occ_state, occ_want)
        assert np.array_equal(got, (st_after & 1) == 1)
        assert np.array_equal(ctx.export_model(unseen), want)
        # the same model carved again by other views (every record must exist and be right)
        _, _, M2 = scenes.random_cameras(3, 0.512, seed=seed + 77, W=W, H=H)
        masks2 = scenes.noise_masks(3, H, W, block=16, p_bg=0.4, seed=seed + 9)
        ctx.set_views(M2, masks2)
        ctx.carve()
        assert np.array_equal(ctx.download_state(), oracle.carve(X, Y, Z, s, M2, masks2, state=st_after))
