"""Parity of the HIP dense carve (through the C-ABI) with the CPU oracle.
Bar: the state plane is bit-exact (occupancy AND seen), every voxel."""
import numpy as np
import pytest

from tests import scenes

pytestmark = pytest.mark.gpu


def run_gpu(arvx, X, Y, Z, s, M, masks, flags=0, state=None, z_range=None, views=None):
    with arvx.Context(X, Y, Z, s, z_range=z_range) as ctx:
        ctx.set_views(M, masks)
        if state is not None:
            ctx.upload_state(state)
        if views is None:
            ctx.carve(flags)
        else:
            ctx.carve_views(views[0], views[1], flags)
        return ctx.download_state()


def assert_same(got, want, what=""):
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        z, y, x = bad[0]
        raise AssertionError(
            f"{what}: {len(bad)} of {got.size} voxels differ; first (x={x},y={y},z={z}) "
            f"gpu={got[z, y, x]} oracle={want[z, y, x]}")


@pytest.mark.parametrize("flags", [0, 1, 8])  # split kernels, brute force, fused kernel
@pytest.mark.parametrize("N,V", [(32, 6), (64, 8)])
def test_sphere_parity(arvx, oracle, N, V, flags):
    sc = scenes.small_sphere(N, V)
    want = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
    got = run_gpu(arvx, N, N, N, sc.voxel_size, sc.M, sc.masks, flags)
    assert_same(got, want, f"sphere {N}^3 x {V} flags={flags}")
    assert (want == 3).sum() > 0 and (want == 2).sum() > 0


@pytest.mark.parametrize("flags", [0, 1, 8])
@pytest.mark.parametrize("dims", [(10, 10, 5), (50, 50, 25), (100, 100, 50), (33, 17, 9),
                                  (64, 8, 8), (1, 1, 1), (130, 7, 19)])
def test_ragged_grids(arvx, oracle, dims, flags):
    """The reference's own benchmark sizes (src/main.cpp:306-440) and odd shapes."""
    X, Y, Z = dims
    sc = scenes.small_sphere(32, 5)
    s = np.float32(0.512 / max(dims))
    want = oracle.carve(X, Y, Z, s, sc.M, sc.masks)
    got = run_gpu(arvx, X, Y, Z, s, sc.M, sc.masks, flags)
    assert_same(got, want, f"grid {dims} flags={flags}")


@pytest.mark.parametrize("flags", [0, 1, 8])
@pytest.mark.parametrize("C,block", [(1, 1), (3, 1), (1, 8), (3, 16)])
def test_noise_masks_random_cameras(arvx, oracle, C, block, flags):
    """Noise masks make every rounding decision visible; random cameras put
    voxels behind the camera, on the image border and at tiny depth."""
    N, V, W, H = 48, 7, 160, 120
    s = np.float32(0.512 / N)
    _, _, M = scenes.random_cameras(V, 0.512, seed=3 + C + block, W=W, H=H, inside=True)
    masks = scenes.noise_masks(V, H, W, C=C, block=block, seed=C * 10 + block)
    want = oracle.carve(N, N, N, s, M, masks)
    got = run_gpu(arvx, N, N, N, s, M, masks, flags)
    assert_same(got, want, f"noise C={C} block={block} flags={flags}")


def test_cull_matches_no_cull_large(arvx):
    """Size-independent property at a size the oracle would take minutes for:
    the culled kernel and the brute-force kernel agree voxel for voxel."""
    sc = scenes.syn.sphere_scene(256, 12)
    a = run_gpu(arvx, 256, 256, 256, sc.voxel_size, sc.M, sc.masks, 0)
    b = run_gpu(arvx, 256, 256, 256, sc.voxel_size, sc.M, sc.masks, 1)
    assert_same(a, b, "cull vs no-cull 256^3")
    occ = (a & 1).mean()
    assert 0.1 < occ < 0.4  # visual hull of the 0.35E sphere
    assert not np.any(a == 0)  # carved implies seen


def test_precarved_state_and_view_ranges(arvx, oracle):
    """carve ANDs into whatever the model holds (voxels set to w=0 by the caller,
    unseen) and a view-by-view run equals the all-views run."""
    N, V = 40, 6
    sc = scenes.small_sphere(N, V)
    rng = np.random.default_rng(5)
    st0 = rng.choice(np.array([0, 1, 2, 3], np.uint8), size=(N, N, N))
    want = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks, state=st0)
    got = run_gpu(arvx, N, N, N, sc.voxel_size, sc.M, sc.masks, 0, state=st0)
    assert_same(got, want, "pre-carved state")
    with arvx.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.upload_state(st0)
        cur = st0
        for i in range(V):
            ctx.carve_views(i, 1)
            cur = oracle.carve_view(N, N, N, sc.voxel_size, sc.M[i], sc.masks[i], cur)
            assert_same(ctx.download_state(), cur, f"after view {i}")
    assert_same(cur, want, "view-by-view == all views")


def test_z_slabs_concatenate(arvx, oracle):
    N, V = 48, 6
    sc = scenes.small_sphere(N, V)
    want = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
    parts = [run_gpu(arvx, N, N, N, sc.voxel_size, sc.M, sc.masks, 0, z_range=r)
             for r in [(0, 13), (13, 14), (14, 40), (40, 48)]]
    assert_same(np.concatenate(parts, axis=0), want, "z slabs")


def test_many_views_chunks(arvx, oracle):
    """More than 64 views: the per-lane view classification runs in chunks."""
    N, V = 24, 70
    sc = scenes.small_sphere(N, V, W=96, H=72)
    want = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
    for flags in (0, 1, 8):
        got = run_gpu(arvx, N, N, N, sc.voxel_size, sc.M, sc.masks, flags)
        assert_same(got, want, f"70 views flags={flags}")


def test_all_background_and_all_foreground(arvx, oracle):
    N, V, W, H = 32, 3, 96, 72
    sc = scenes.small_sphere(N, V, W=W, H=H)
    for fill in (0, 255):
        masks = np.full((V, H, W), fill, np.uint8)
        want = oracle.carve(N, N, N, sc.voxel_size, sc.M, masks)
        got = run_gpu(arvx, N, N, N, sc.voxel_size, sc.M, masks)
        assert_same(got, want, f"fill={fill}")


def test_stats_and_errors(arvx):
    sc = scenes.small_sphere(64, 8)
    with arvx.Context(64, 64, 64, sc.voxel_size) as ctx:
        with pytest.raises(arvx.ArvxError):
            ctx.carve()  # no views yet
        ctx.set_views(sc.M, sc.masks)
        ctx.carve(arvx.CARVE_STATS)
        st = ctx.stats()
        assert st["subtiles"] == 4 * 8 * 8
        assert 0 < st["subtiles_carved"] < st["subtiles"]
        with pytest.raises(arvx.ArvxError):
            ctx.carve_views(5, 10)
    with pytest.raises(arvx.ArvxError):
        arvx.Context(0, 4, 4, 0.1)
    with pytest.raises(arvx.ArvxError):
        arvx.Context(4, 4, 4, -1.0)


def test_pack_occupancy_layout(arvx, oracle):
    """voxel i -> bit i%32 of word i//32, slab-local: what the RCCL exchange ships."""
    import torch
    X, Y, Z, V = 24, 8, 12, 4
    sc = scenes.small_sphere(32, V, W=96, H=72)
    s = np.float32(0.512 / 24)
    want = oracle.carve(X, Y, Z, s, sc.M, sc.masks)
    for zr in [(0, 12), (4, 8)]:
        with arvx.Context(X, Y, Z, s, z_range=zr) as ctx:
            ctx.set_views(sc.M, sc.masks)
            ctx.carve()
            n = X * Y * (zr[1] - zr[0])
            words = torch.full(((n + 31) // 32 + 2,), -1, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()  # the fill runs on torch's stream, the pack on the context's
            ctx.pack_occupancy(words.data_ptr())
            ctx.synchronize()
            got = words.cpu().numpy()
        occ = (want[zr[0]:zr[1]].reshape(-1) & 1).astype(np.uint8)
        ref = np.packbits(occ, bitorder="little").view(np.int32)
        assert np.array_equal(got[:len(ref)], ref)
        assert np.all(got[len(ref):] == -1), "wrote past the slab's words"


@pytest.mark.parametrize("seed,block,inside", [(1, 8, False), (2, 32, True), (3, 64, False),
                                               (4, 16, True), (5, 128, True)])
def test_cull_matches_no_cull_block_noise(arvx, seed, block, inside):
    """Blocky random masks give large all-background / all-foreground rectangles next
    to edges everywhere: the rectangle classification (coarse pre-pass, inheritance,
    fp32 margins) is exercised on every outcome, with cameras inside and outside the
    grid.  The brute-force kernel (NO_CULL) is the reference; it is itself checked
    against the oracle at small sizes."""
    N, V, W, H = 192, 9, 640, 480
    s = np.float32(0.512 / N)
    _, _, M = scenes.random_cameras(V, 0.512, seed=seed, W=W, H=H, inside=inside)
    masks = scenes.noise_masks(V, H, W, block=block, p_bg=0.55, seed=seed + 100)
    a = run_gpu(arvx, N, N, N, s, M, masks, 0)
    b = run_gpu(arvx, N, N, N, s, M, masks, 1)
    assert_same(a, b, f"cull vs no-cull, block noise seed={seed}")
    assert_same(run_gpu(arvx, N, N, N, s, M, masks, 8), b, "fused kernel")
    # slabs and view sub-ranges go through the same coarse tables
    c = np.concatenate([run_gpu(arvx, N, N, N, s, M, masks, 0, z_range=r)
                        for r in [(0, 70), (70, 71), (71, 192)]], axis=0)
    assert_same(c, b, "slabs")
    with arvx.Context(N, N, N, s) as ctx:
        ctx.set_views(M, masks)
        ctx.carve_views(0, 4)
        ctx.carve_views(4, 5)
        assert_same(ctx.download_state(), b, "two view ranges")


@pytest.mark.parametrize("seed,block,p_bg", [(11, 2, 0.02), (12, 2, 0.10), (13, 1, 0.005)])
def test_cull_matches_no_cull_sparse_background(arvx, seed, block, p_bg):
    """Mostly-foreground masks with a few small background squares (bench.py's `workloads`: the
    regime where no rectangle test settles anything and nearly every voxel stays occupied through
    all views -- no early-out ends an item): the split carve and the fused kernel against the
    brute-force kernel, voxel for voxel; the reference has no early-out at all
    (src/VoxelCarving.cpp:39-55)."""
    N, V, W, H = 192, 9, 640, 480
    s = np.float32(0.512 / N)
    _, _, M = scenes.random_cameras(V, 0.512, seed=seed, W=W, H=H, inside=False)
    masks = scenes.noise_masks(V, H, W, block=block, p_bg=p_bg, seed=seed + 100)
    b = run_gpu(arvx, N, N, N, s, M, masks, arvx.CARVE_NO_CULL)
    assert (b & 1).mean() > 0.3  # (voxels do stay alive)
    assert_same(run_gpu(arvx, N, N, N, s, M, masks, 0), b, f"cull vs no-cull, sparse background seed={seed}")
    assert_same(run_gpu(arvx, N, N, N, s, M, masks, arvx.CARVE_FUSED), b, "fused kernel")


def test_lazy_reset_and_reuse(arvx, oracle):
    """reset() is lazy (the carve writes every voxel); every reader must still see a
    fresh plane, and a context can be carved, reset and carved again."""
    N, V = 32, 5
    sc = scenes.small_sphere(N, V)
    want = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
    with arvx.Context(N, N, N, sc.voxel_size) as ctx:
        assert np.all(ctx.download_state() == 1)  # fresh at creation
        ctx.set_views(sc.M, sc.masks)
        ctx.carve()
        assert_same(ctx.download_state(), want, "first carve")
        ctx.reset()
        assert np.all(ctx.download_state() == 1)
        ctx.reset()
        ctx.carve(arvx.CARVE_NO_CULL)
        assert_same(ctx.download_state(), want, "after reset, no-cull")
        ctx.reset()
        ctx.carve_views(0, 2)
        ctx.carve_views(2, 3)
        assert_same(ctx.download_state(), want, "after reset, two ranges")


@pytest.mark.parametrize("world", [2, 3, 8])
def test_striped_slabs(arvx, oracle, world):
    """Striped (load-balanced) split: rank r holds 8-plane groups r, r+world, ...; the
    union over ranks is the full carve, and pack_occupancy_global puts every plane's
    bits at its place in the whole grid's word plane."""
    import torch
    X, Y, Z, V = 48, 40, 64, 6
    sc = scenes.small_sphere(64, V)
    s = np.float32(0.512 / 64)
    want = oracle.carve(X, Y, Z, s, sc.M, sc.masks)
    total_words = X * Y * Z // 32
    merged = torch.zeros(total_words, dtype=torch.int32, device="cuda")
    got = np.zeros_like(want)
    for rank in range(world):
        with arvx.Context(X, Y, Z, s, stripes=(world, rank)) as ctx:
            ctx.set_views(sc.M, sc.masks)
            ctx.carve()
            st = ctx.download_state()
            assert st.shape[0] == len(ctx.planes)
            got[ctx.planes] = st
            words = torch.zeros(total_words, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()  # the fill runs on torch's stream, the pack on the context's
            ctx.pack_occupancy_global(words.data_ptr())
            ctx.synchronize()
            merged += words  # what the SUM all-reduce does
            with pytest.raises(arvx.ArvxError):
                ctx.fast_carve()
    assert_same(got, want, f"striped world={world}")
    ref = np.packbits((want.reshape(-1) & 1).astype(np.uint8), bitorder="little").view(np.int32)
    assert np.array_equal(merged.cpu().numpy(), ref)
    # contiguous slabs through the same global pack
    merged.zero_()
    for zr in [(0, 24), (24, 64)]:
        with arvx.Context(X, Y, Z, s, z_range=zr) as ctx:
            ctx.set_views(sc.M, sc.masks)
            ctx.carve()
            words = torch.zeros(total_words, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()  # the fill runs on torch's stream, the pack on the context's
            ctx.pack_occupancy_global(words.data_ptr())
            ctx.synchronize()
            merged += words
    assert np.array_equal(merged.cpu().numpy(), ref)


@pytest.mark.parametrize("W,H,C,V", [(33, 17, 1, 3), (97, 61, 3, 1), (64, 64, 4, 64), (40, 30, 2, 65),
                                     (1, 1, 1, 2)])
def test_odd_image_sizes_and_view_counts(arvx, oracle, W, H, C, V):
    N = 28
    s = np.float32(0.512 / N)
    _, _, M = scenes.random_cameras(V, 0.512, seed=W + V, W=W, H=H, inside=(V % 2 == 0))
    masks = scenes.noise_masks(V, H, W, C=C, block=max(1, W // 8), seed=H)
    want = oracle.carve(N, N, N, s, M, masks)
    for flags in (0, 1):
        assert_same(run_gpu(arvx, N, N, N, s, M, masks, flags), want,
                    f"{W}x{H}x{C} V={V} flags={flags}")


def test_camera_centre_on_a_voxel(arvx, oracle):
    """proj2 == 0 exactly for the voxel the camera sits on: 0/0 and x/0 -> outside."""
    N, W, H = 24, 96, 72
    s = np.float32(0.5 / 16)  # power of two: the voxel position is exact
    K = scenes.syn.K_DATASET.copy()
    K[0] *= W / 640
    K[1] *= H / 480
    K32 = K.astype(np.float32)
    Rts = []
    for (vx, vy, vz) in [(5, 6, 7), (0, 0, 0), (23, 23, 23)]:
        cam = np.array([vy * float(s), vx * float(s), -vz * float(s)])
        Rts.append(scenes.syn.look_at_rt(cam, cam + np.array([0.3, 0.2, -0.5])))
    Rt = np.array(Rts)
    Rt[:, :, :3] = np.round(Rt[:, :, :3] * 64) / 64  # few mantissa bits: R*cam is exact
    Rt[:, :, 3] = -np.einsum("vij,vj->vi", Rt[:, :, :3],
                             np.array([[6 * s, 5 * s, -7 * s], [0, 0, 0], [23 * s, 23 * s, -23 * s]]))
    Rt = Rt.astype(np.float32)
    M = scenes.syn.compose_m(K32, Rt)
    raw = oracle.project_raw(M[0], s, 5, 6, 7)
    assert raw[2] == 0.0 and oracle.project(M[0], s, 5, 6, 7, W, H) is None
    masks = scenes.noise_masks(3, H, W, block=3, seed=5)
    want = oracle.carve(N, N, N, s, M, masks)
    for flags in (0, 1):
        assert_same(run_gpu(arvx, N, N, N, s, M, masks, flags), want, f"on-voxel camera {flags}")


def test_row_stride_padding(arvx, oracle):
    """Masks with padded rows (cv::Mat step > cols*channels) through arvx_set_views."""
    import ctypes as C
    N, V, W, H, Cn = 20, 3, 50, 30, 3
    s = np.float32(0.512 / N)
    _, _, M = scenes.random_cameras(V, 0.512, seed=1, W=W, H=H)
    tight = scenes.noise_masks(V, H, W, C=Cn, block=4, seed=2)
    stride = W * Cn + 13
    padded = np.full((V, H, stride), 0xAB, np.uint8)  # garbage in the padding
    padded[:, :, :W * Cn] = tight.reshape(V, H, W * Cn)
    want = oracle.carve(N, N, N, s, M, tight)
    lib = arvx.load_library()
    with arvx.Context(N, N, N, s) as ctx:
        ptrs = (C.c_void_p * V)(*[padded[i].ctypes.data for i in range(V)])
        Mf = np.ascontiguousarray(M, np.float32).reshape(-1)
        rc = lib.arvx_set_views(ctx._h, V, Mf.ctypes.data_as(C.POINTER(C.c_float)), None, ptrs,
                                W, H, Cn, stride)
        assert rc == 0, lib.arvx_last_error()
        ctx.carve()
        assert_same(ctx.download_state(), want, "padded rows")
        assert lib.arvx_set_views(ctx._h, V, Mf.ctypes.data_as(C.POINTER(C.c_float)), None, ptrs,
                                  W, H, Cn, W * Cn - 1) == 1  # stride too small


def test_shared_reciprocal_division_is_ieee(arvx):
    """divide2_shared_rcp (one v_rcp for both quotients) is the hardware's own division sequence
    without v_div_scale (csrc/arvx_device.h): inside the range the kernel uses it in (|b| in
    [2^-59, 2^59], |a| <= 2^59) it must equal the IEEE `/` BIT FOR BIT on every operand pair
    that v_div_scale would leave alone; the pairs it would rescale -- numerators below 2^-103,
    exponent differences >= 96 or <= -126 -- may differ, and then both quotients are so small
    (pixel 0) or so large (outside any image) that the pixel decision is the same."""
    rng = np.random.default_rng(0)
    n = 1 << 21
    parts = []
    # projection-like magnitudes
    parts.append((rng.normal(scale=300, size=n), rng.normal(scale=300, size=n),
                  rng.uniform(0.05, 4.0, size=n) * rng.choice([-1, 1], size=n)))
    # whole exponent range the guard admits
    ea, eb = rng.uniform(-59, 59, size=n), rng.uniform(-59, 59, size=n)
    parts.append((np.ldexp(rng.uniform(1, 2, n), ea.astype(int)) * rng.choice([-1, 1], n),
                  np.ldexp(rng.uniform(1, 2, n), rng.uniform(-140, 59, n).astype(int)),
                  np.ldexp(rng.uniform(1, 2, n), eb.astype(int)) * rng.choice([-1, 1], n)))
    # quotients engineered to sit on and next to rounding ties k + 0.5
    b = rng.uniform(0.1, 3.0, size=n).astype(np.float32)
    k = rng.integers(0, 700, size=n).astype(np.float32) + 0.5
    a = (k * b).astype(np.float32)
    a = np.nextafter(a, np.float32(np.inf) * rng.choice([-1, 1], n).astype(np.float32))
    parts.append((a, (k * b).astype(np.float32), b))
    # zeros and tiny numerators
    parts.append((np.zeros(n), np.ldexp(rng.uniform(1, 2, n), -140), rng.uniform(0.5, 2, n)))
    a0 = np.concatenate([p[0] for p in parts]).astype(np.float32)
    a1 = np.concatenate([p[1] for p in parts]).astype(np.float32)
    bb = np.concatenate([p[2] for p in parts]).astype(np.float32)
    with arvx.Context(4, 4, 4, 0.1) as ctx:
        out = ctx.selftest_divide(a0, a1, bb)
    with np.errstate(all="ignore"):
        ref0, ref1 = (a0 / bb).astype(np.float32), (a1 / bb).astype(np.float32)
    assert np.array_equal(out[:, 2].view(np.uint32), ref0.view(np.uint32)), "GPU `/` is IEEE"
    assert np.array_equal(out[:, 3].view(np.uint32), ref1.view(np.uint32))
    eb = np.frexp(bb)[1]
    for fast, ref, a in ((out[:, 0], ref0, a0), (out[:, 1], ref1, a1)):
        same = fast.view(np.uint32) == ref.view(np.uint32)
        ediff = np.frexp(a)[1].astype(np.int64) - eb
        rescaled = ((a != 0) & (np.abs(a) < 2.0 ** -103)) | (ediff >= 96) | (ediff <= -125)
        bad = ~same & ~rescaled
        assert not bad.any(), (f"{bad.sum()} quotients differ where v_div_scale does not rescale, "
                               f"e.g. {a[bad][:3]} / {bb[bad][:3]}: {fast[bad][:3]} vs {ref[bad][:3]}")
        # where it would: the same pixel decision
        tiny = (np.abs(ref) < 2.0 ** -40) & (np.abs(fast) < 2.0 ** -40)
        huge = (np.abs(ref) > 2.0 ** 90) & (np.abs(fast) > 2.0 ** 90)
        assert ((same | tiny | huge)[rescaled]).all()
        assert rescaled.mean() < 0.5 and same.mean() > 0.7  # (a quarter of the data is tiny numerators)


def test_cull_matches_no_cull_bench_scene_512(arvx):
    """The bench workload itself (512^3 x 36, sphere): culled == brute force, every voxel."""
    sc = scenes.syn.sphere_scene(512, 36)
    with arvx.Context(512, 512, 512, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.carve(0)
        a = ctx.download_state()
        ctx.reset()
        ctx.carve(arvx.CARVE_NO_CULL)
        b = ctx.download_state()
    assert_same(a, b, "512^3 x 36 sphere")
    assert abs(float((a & 1).mean()) - 0.1806) < 0.001


def _flat_views(X, Y, s):
    """One view that looks along z: pixel (u, v) = voxel (x, y) (Model::toWord swaps x and y)."""
    M = np.zeros((1, 3, 4), np.float32)
    M[0, 0, 1] = 1.0 / s  # u = world[1] / s = x
    M[0, 1, 0] = 1.0 / s  # v = world[0] / s = y
    M[0, 2, 3] = 1.0
    return M


def test_tile_summary_is_dropped_when_something_else_writes_the_state(arvx, oracle):
    """A carve remembers which coarse tiles (64 x 32 x 32 voxels) it emptied as a whole and a
    later carve of the same model skips them (CarveParams::cstate).  Carving cannot undo that --
    an upload or the closure can: the summary must not survive them.  A view along z with
    background in the rows y < 32 empties the coarse tiles of those rows; then
      * an upload of an all-occupied model and the same carve: emptied again, and
      * the closure (a dilation: reference F10) puts voxels of row y = 31 back: carved again."""
    X, Y, Z = 64, 96, 64
    s = np.float32(0.01)
    M = _flat_views(X, Y, s)
    masks = np.full((1, Y, X), 255, np.uint8)
    masks[0, :32, :] = 0
    want1 = oracle.carve(X, Y, Z, s, M, masks)
    assert (want1[:, :32, :] == 2).all() and (want1[:, 32:, :] == 3).all()
    with arvx.Context(X, Y, Z, s) as ctx:
        ctx.set_views(M, masks)
        ctx.carve()
        assert_same(ctx.download_state(), want1, "first carve")
        ctx.carve()  # (nothing left to do in the emptied tiles)
        assert_same(ctx.download_state(), want1, "second carve")
        full = np.full((Z, Y, X), 1, np.uint8)
        ctx.upload_state(full)
        ctx.carve()
        assert_same(ctx.download_state(), want1, "after an upload")
        ctx.upload_planes(*planes_of(full))
        ctx.carve()
        assert_same(ctx.download_state(), want1, "after an upload of planes")
        idx, _ = ctx.closure(3, True)
        after = ctx.download_state()
        filled = np.zeros(X * Y * Z, bool)
        filled[idx] = True
        filled = filled.reshape(Z, Y, X)
        assert filled[:, 31, :].all() and not filled[:, :31, :].any()
        assert ((after & 1) == 1)[filled].all()
        ctx.carve()  # the voxels of row 31 project onto background: carved again
        assert_same(ctx.download_state(), oracle.carve(X, Y, Z, s, M, masks, state=after), "after the closure")
        assert ((ctx.download_state() & 1) == 0)[filled].all()


def test_views_one_by_one_on_emptied_and_seen_tiles(arvx, oracle):
    """The same summary, view by view on a grid with many coarse tiles: tiles emptied or seen as a
    whole by earlier views are not revisited, the result is that of the oracle after every view."""
    N, V = 160, 9
    sc = scenes.syn.sphere_scene(N, V, W=320, H=240)
    cur = np.full((N, N, N), 1, np.uint8)
    with arvx.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        for i in list(range(V)) + [2, 0]:
            ctx.carve_views(i, 1)
            cur = oracle.carve_view(N, N, N, sc.voxel_size, sc.M[i], sc.masks[i], cur)
            assert_same(ctx.download_state(), cur, f"view {i}")


def test_size_independent_properties_at_the_target_size_1024(arvx):
    """1024^3 x 36 views: what carving guarantees whatever the size (reference
    src/VoxelCarving.cpp:38-60: a voxel is carved iff SOME view shows background at its pixel, seen
    iff SOME view has it inside the image) -- on the paths a fresh carve alone does not take:
      * idempotence: carving the carved model again changes nothing (a model that is not fresh:
        no lazy codes, the coarse fill kernel, records read before they are written);
      * the views in two halves, first the second half then the first: the same model (order of
        the views, state carried between calls);
      * a Z slab is the same planes of the whole grid."""
    N, V = 1024, 36
    sc = scenes.syn.sphere_scene(N, V)
    with arvx.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.carve(0)
        occ, seen = ctx.download_planes()
        ctx.carve(0)  # again, on the carved model
        occ2, seen2 = ctx.download_planes()
        assert np.array_equal(occ, occ2) and np.array_equal(seen, seen2), "idempotence"
        ctx.reset()
        ctx.carve_views(V // 2, V - V // 2)
        ctx.carve_views(0, V // 2)
        occ3, seen3 = ctx.download_planes()
        assert np.array_equal(occ, occ3) and np.array_equal(seen, seen3), "views in two halves"
    frac = np.bitwise_count(occ).sum() / N ** 3
    assert abs(float(frac) - 0.1806) < 0.001
    z0, z1 = 384, 648  # (not a multiple of the coarse tile's 32 planes at the upper end)
    wpp = occ.size // N  # words per plane
    with arvx.Context(N, N, N, sc.voxel_size, z_range=(z0, z1)) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.carve(0)
        so, ss = ctx.download_planes()
    assert np.array_equal(so, occ[z0 * wpp:z1 * wpp]) and np.array_equal(ss, seen[z0 * wpp:z1 * wpp]), "slab"


def test_cull_matches_no_cull_target_config_1024(arvx):
    """The north-star target configuration, 1024^3 x 36 views: culled == brute force."""
    sc = scenes.syn.sphere_scene(1024, 36)
    with arvx.Context(1024, 1024, 1024, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.carve(0)
        a = ctx.download_state()
        ctx.reset()
        ctx.carve(arvx.CARVE_NO_CULL)
        b = ctx.download_state()
    assert np.array_equal(a, b)
    assert abs(float((a & 1).mean()) - 0.1806) < 0.001 and not np.any(a == 0)


@pytest.mark.parametrize("which,N", [("box", 128), ("human", 96)])
def test_dataset_silhouettes_plumbing(arvx, oracle, which, N):
    """BASELINE config 0/1 shaped run on the reference data set's own (PIL-decoded)
    masks with synthetic ring poses: plumbing, not a parity claim about the reference's
    run.  Ragged real silhouettes with JPEG speckle: oracle == culled == brute force."""
    from tests import golden_io
    masks = golden_io.dataset_masks(which, recentre=(which == "human"))
    V = masks.shape[0]
    sc = scenes.syn.sphere_scene(N, V)  # ring cameras + dataset intrinsics, 640x480
    want = oracle.carve(N, N, N, sc.voxel_size, sc.M, masks)
    for flags in (0, 1):
        assert_same(run_gpu(arvx, N, N, N, sc.voxel_size, sc.M, masks, flags), want,
                    f"{which} masks flags={flags}")
    assert 0.0 < (want & 1).mean() < 1.0


@pytest.mark.parametrize("case", ["tiny_voxels", "huge_voxels", "long_focal", "short_focal",
                                  "far_camera", "offcentre_principal", "grazing"])
def test_extreme_geometry(arvx, oracle, case):
    """Margins of the rectangle tests scale with |M|, |w| and 1/depth: push each."""
    N, V, W, H = 40, 5, 200, 150
    rng = np.random.default_rng(sum(ord(c) for c in case))  # fixed per case
    s = {"tiny_voxels": 1e-4, "huge_voxels": 7.5}.get(case, 0.512 / N)
    E = s * N
    f = {"long_focal": 6000.0, "short_focal": 25.0}.get(case, 160.0)
    cxp, cyp = (W / 2, H / 2) if case != "offcentre_principal" else (-300.0, 900.0)
    K = np.array([[f, 0, cxp], [0, f * 1.01, cyp], [0, 0, 1]], np.float64)
    centre = np.array([E / 2, E / 2, -E / 2])
    Rts = []
    for i in range(V):
        d = {"far_camera": 200.0, "grazing": 0.75}.get(case, 2.0) * E
        dirv = rng.normal(size=3)
        dirv /= np.linalg.norm(dirv)
        cam = centre + d * dirv
        target = centre + rng.normal(scale=0.2 * E, size=3)
        if case == "grazing":  # look along a face of the grid: many voxels near depth 0
            target = cam + np.array([1.0, 0.02, 0.01]) * E
        Rts.append(scenes.syn.look_at_rt(cam, target, rng.normal(size=3)))
    Rt = np.array(Rts).astype(np.float32)
    M = scenes.syn.compose_m(K.astype(np.float32), Rt)
    for block in (1, 12):
        masks = scenes.noise_masks(V, H, W, block=block, p_bg=0.5, seed=block)
        want = oracle.carve(N, N, N, np.float32(s), M, masks)
        for flags in (0, 1, 8):
            assert_same(run_gpu(arvx, N, N, N, np.float32(s), M, masks, flags), want,
                        f"{case} block={block} flags={flags}")


def test_more_than_256_views_uses_fused_kernel(arvx, oracle):
    """The split launch carries up to 4 chunks of 64 views; beyond that arvx_carve falls
    back to the fused kernel.  Same answer either way."""
    N, V, W, H = 16, 260, 48, 36
    s = np.float32(0.512 / N)
    _, _, M = scenes.random_cameras(V, 0.512, seed=9, W=W, H=H, inside=False)
    masks = scenes.noise_masks(V, H, W, block=6, p_bg=0.25, seed=10)
    want = oracle.carve(N, N, N, s, M, masks)
    for flags in (0, 1, 8):
        assert_same(run_gpu(arvx, N, N, N, s, M, masks, flags), want, f"260 views flags={flags}")
    # 256 views: the largest count the split launch takes
    want = oracle.carve(N, N, N, s, M[:256], masks[:256])
    assert_same(run_gpu(arvx, N, N, N, s, M[:256], masks[:256], 0), want, "256 views")


@pytest.mark.parametrize("N,V", [(96, 36), (160, 24), (224, 12), (288, 8), (352, 6), (416, 6)])
def test_work_distribution_regimes(arvx, oracle, N, V):
    """The exact kernel hands its sub-tiles out differently depending on how many there are
    per wave: items shared by 8 / 4 / 2 waves that merge with atomics (small grids), whole
    items with no pool, a small pool (look before draw), a large pool (walk).  The sphere
    scene puts these grid sizes into the different regimes; every one must give the oracle's
    plane, from a fresh model and from a pre-carved one (parts then start from loaded state),
    and the same with the sharing switched off is checked through the brute-force kernel."""
    sc = scenes.small_sphere(N, V, W=320, H=240)
    want = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
    got = run_gpu(arvx, N, N, N, sc.voxel_size, sc.M, sc.masks)
    assert_same(got, want, f"{N}^3 x {V} fresh")
    assert_same(run_gpu(arvx, N, N, N, sc.voxel_size, sc.M, sc.masks, 1), want, f"{N}^3 brute force")
    # second half of the views on top of the first half's result
    h = V // 2
    first = oracle.carve(N, N, N, sc.voxel_size, sc.M[:h], sc.masks[:h])
    with arvx.Context(N, N, N, sc.voxel_size) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.carve_views(0, h)
        assert_same(ctx.download_state(), first, f"{N}^3 first {h} views")
        ctx.carve_views(h, V - h)
        assert_same(ctx.download_state(), want, f"{N}^3 remaining views on a carved model")


def test_pixel_rounding_instruction_is_std_round(arvx):
    """The kernels round a quotient to its pixel with one v_cvt_rpi_i32_f32; on every float
    in (-0.5, 2^24] -- every quotient that can land inside an image -- it must equal
    (int)std::round (and the floor/fract form it replaced)."""
    with arvx.Context(4, 4, 4, 0.1) as ctx:
        assert ctx.selftest_round() == 0


def test_fuzz_split_against_brute_force(arvx):
    """tools/fuzz_cull.py in small: random ragged grids, cameras and noise masks; the split
    carve (from a fresh and from a pre-carved model) must equal the brute-force kernel."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fuzz_cull", os.path.join(root, "tools", "fuzz_cull.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(2026)
    assert all(fz.one_case(rng, i) for i in range(120))


def test_fuzz_against_oracle(arvx, oracle):
    """Random small ragged grids, cameras (some inside the grid), image sizes, channel counts
    and noise masks: GPU plane == oracle plane, every voxel, 200 cases."""
    rng = np.random.default_rng(4242)
    for i in range(200):
        X, Y, Z = (int(rng.integers(1, 56)) for _ in range(3))
        V = int(rng.integers(1, 20))
        W, H = int(rng.integers(8, 120)), int(rng.integers(8, 90))
        s = np.float32(0.512 / max(X, Y, Z))
        _, _, M = scenes.random_cameras(V, 0.512, seed=int(rng.integers(1 << 30)), W=W, H=H,
                                        inside=rng.random() < 0.4)
        masks = scenes.noise_masks(V, H, W, C=int(rng.choice([1, 3])),
                                   p_bg=float(rng.uniform(0.2, 0.8)),
                                   block=int(rng.choice([1, 3, 8, 32])),
                                   seed=int(rng.integers(1 << 30)))
        want = oracle.carve(X, Y, Z, s, M, masks)
        assert_same(run_gpu(arvx, X, Y, Z, s, M, masks), want,
                    f"case {i}: {X}x{Y}x{Z} V={V} {W}x{H}")


@pytest.mark.parametrize("W,H,C", [(64, 3, 1), (128, 50, 3), (192, 117, 1), (640, 480, 3),
                                   (96, 64, 3), (65, 64, 1), (64, 600, 1), (130, 129, 4),
                                   (96, 70, 1), (70, 33, 1), (63, 40, 1), (127, 70, 3),
                                   (2000, 1500, 1), (4100, 70, 3)])
def test_view_preprocessing_paths(arvx, oracle, W, H, C):
    """Bit planes and summed-area tables (csrc/views_kernels.h) for image widths that are and
    are not multiples of 64 (tile columns of the table kernels), heights above and below one
    tile row, 1 and 3 channels (one channel with W x H % 32 == 0: views_bits16_kernel; else
    views_bits_kernel): checked through the carve they drive, from host masks and from
    masks already on the device (re-derived twice).  W = 63, 64, 127: the last table column
    (X = W) alone in a tile, in the first tile, at a tile's end."""
    import torch
    N, V = 40, 5
    s = np.float32(0.512 / N)
    _, _, M = scenes.random_cameras(V, 0.512, seed=W + H + C, W=W, H=H, inside=False)
    masks = scenes.noise_masks(V, H, W, C=C, block=5, p_bg=0.4, seed=W * 7 + C)
    want = oracle.carve(N, N, N, s, M, masks)
    assert_same(run_gpu(arvx, N, N, N, s, M, masks), want, f"{W}x{H}x{C} host masks")
    d_masks = torch.from_numpy(np.ascontiguousarray(masks)).cuda()
    with arvx.Context(N, N, N, s) as ctx:
        for _ in range(2):
            ctx.reset()
            ctx.set_views_device(M, d_masks.data_ptr(), W, H, C)
            ctx.carve()
            assert_same(ctx.download_state(), want, f"{W}x{H}x{C} device masks")


@pytest.mark.parametrize("W,H,C", [(64, 3, 1), (640, 480, 1), (640, 480, 3), (65, 64, 1), (63, 40, 1),
                                   (127, 70, 3), (130, 129, 4), (192, 117, 1), (96, 64, 3),
                                   (1000, 700, 1), (2000, 150, 1), (1024, 100, 1), (128, 65, 1),
                                   (128, 65, 3)])
def test_view_tables_against_numpy(arvx, W, H, C):
    """arvx_selftest_view_tables: the background bit plane and the summed-area table derived by
    arvx_set_views[_device] (csrc/views_kernels.h), entry for entry: bit = all channel bytes zero
    (reference src/VoxelCarving.cpp:49-50); table[Y][X] = foreground pixels in rows < Y and columns
    < X, modulo 2^16 (1000 x 700 all foreground runs far past 2^16).  Through the carve alone a
    wrong entry could hide behind a conservative "mixed" answer."""
    import torch
    V = 3
    _, _, M = scenes.random_cameras(V, 0.512, seed=W + H, W=W, H=H, inside=False)
    masks = scenes.noise_masks(V, H, W, C=C, block=7, p_bg=0.45, seed=W * 3 + H + C)
    masks[1] = 255  # every pixel foreground
    masks[2][:, : W // 2] = 0  # left half background
    d_masks = torch.from_numpy(np.ascontiguousarray(masks)).cuda()
    with arvx.Context(8, 8, 8, np.float32(0.064)) as ctx:
        for src in ("host", "device"):
            if src == "host":
                ctx.set_views(M, masks)
            else:
                ctx.set_views_device(M, d_masks.data_ptr(), W, H, C)
            for v in range(V):
                bg, table = ctx.selftest_view_tables(v, W, H)
                want_bg = (masks[v].reshape(H, W, -1) == 0).all(axis=2)
                assert np.array_equal(bg, want_bg), f"{src} view {v}: bit plane"
                full = np.zeros((H + 1, W + 1), np.int64)
                full[1:, 1:] = np.cumsum(np.cumsum(~want_bg, axis=0, dtype=np.int64), axis=1)
                bad = np.argwhere(table != (full & 0xffff).astype(np.uint16))
                assert bad.size == 0, f"{src} view {v}: {len(bad)} entries differ, first at {bad[0]}"


@pytest.mark.parametrize("N,V,W,H", [(64, 6, 160, 120), (256, 36, 640, 480)])
def test_jobs_in_flight_on_several_contexts(arvx, N, V, W, H):
    """Several jobs in flight, as bench.py --jobs runs them: job k on context and stream k % 3,
    scenes alternating, nothing between the jobs but stream order inside a slot.  Every job's
    packed occupancy must be that of its scene carved alone (contexts share nothing: each has its
    own tables, lists and work buffers).  At 256^3 x 36 views of 640 x 480 the view derivation of
    one job really runs beside the exact kernel of another."""
    import torch
    scs = [scenes.syn.sphere_scene(N, V, W=W, H=H), scenes.syn.box_scene((N, N, N), V, W=W, H=H)]
    n = N * N * N // 64
    want = []
    with arvx.Context(N, N, N, scs[0].voxel_size) as c:  # one job at a time
        for sc in scs:
            c.set_views(sc.M, sc.masks)
            c.reset()
            c.carve()
            d = torch.zeros(n, dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            c.pack_occupancy(d.data_ptr())
            c.synchronize()
            want.append(d.clone())
    assert not torch.equal(want[0], want[1])
    d_masks = [torch.from_numpy(np.ascontiguousarray(sc.masks)).cuda() for sc in scs]
    slots, jobs = 3, 14
    streams = [torch.cuda.Stream() for _ in range(slots)]
    ctxs = [arvx.Context(N, N, N, scs[0].voxel_size) for _ in range(slots)]
    got = [torch.zeros(n, dtype=torch.int64, device="cuda") for _ in range(jobs)]
    torch.cuda.synchronize()
    try:
        for c, st in zip(ctxs, streams):
            c.set_stream(st.cuda_stream)
        for k in range(jobs):
            c, sc = ctxs[k % slots], scs[(k * 5 // 3) % 2]
            c.reset()
            c.set_views_device(sc.M, d_masks[(k * 5 // 3) % 2].data_ptr(), sc.W, sc.H, 1)
            c.carve()
            c.pack_occupancy(got[k].data_ptr())  # (same stream: ordered behind the carve)
        torch.cuda.synchronize()
    finally:
        for c in ctxs:
            c.close()
    for k in range(jobs):
        assert torch.equal(got[k], want[(k * 5 // 3) % 2]), f"job {k}"


@pytest.mark.parametrize("dims", [(2112, 8, 9), (8, 2112, 9), (9, 8, 2112), (4160, 3, 5)])
def test_long_thin_grids(arvx, oracle, dims):
    """One extent far beyond the others: thousands of tiles / coarse tiles along a single axis."""
    X, Y, Z = dims
    V, W, H = 4, 200, 60
    s = np.float32(0.512 / max(dims))
    _, _, M = scenes.random_cameras(V, 0.512, seed=sum(dims), W=W, H=H)
    masks = scenes.noise_masks(V, H, W, block=6, p_bg=0.6, seed=X + 2)
    want = oracle.carve(X, Y, Z, s, M, masks)
    for flags in (0, arvx.CARVE_NO_CULL):
        assert_same(run_gpu(arvx, X, Y, Z, s, M, masks, flags=flags), want, f"{dims} flags {flags}")


def test_view_by_view_on_random_states_many_seeds(arvx, oracle):
    """One view at a time on random pre-carved states (every combination of occupied / seen),
    re-synchronised with the oracle after every view: single-view launches have only a handful
    of sub-tiles with exact work -- the regime in which items are shared between waves -- and a
    ragged grid puts partial sub-tiles and padded tiles on the work lists.  (This is the test
    that found work-list overflow when sub-tiles were numbered over the list instead of over
    the grid.)"""
    N, V = 40, 6
    sc = scenes.small_sphere(N, V)
    for seed in range(12):
        rng = np.random.default_rng(seed)
        cur = rng.choice(np.array([0, 1, 2, 3], np.uint8), size=(N, N, N))
        with arvx.Context(N, N, N, sc.voxel_size) as ctx:
            ctx.set_views(sc.M, sc.masks)
            for i in range(V):
                ctx.upload_state(cur)
                ctx.carve_views(i, 1)
                cur = oracle.carve_view(N, N, N, sc.voxel_size, sc.M[i], sc.masks[i], cur)
                assert_same(ctx.download_state(), cur, f"seed {seed} view {i}")


def planes_of(state):
    """(occ, seen) uint32 bit planes of a (Z, Y, X) state array, rows padded to 32-bit words."""
    Z, Y, X = state.shape
    wpr = (X + 31) // 32
    out = []
    for bit in (1, 2):
        b = np.zeros((Z, Y, wpr * 32), np.uint8)
        b[:, :, :X] = (state & bit) != 0
        out.append(np.packbits(b, axis=2, bitorder="little").view(np.uint32).reshape(-1))
    return out


@pytest.mark.parametrize("dims,zr", [((64, 8, 8), None), ((100, 100, 50), None), ((33, 17, 9), None),
                                     ((40, 24, 40), (9, 31)), ((128, 64, 32), (8, 16))])
def test_bit_plane_transfers(arvx, oracle, dims, zr):
    """arvx_state_download_planes / _upload_planes (the 2-bit form a host Model keeps) against
    the byte plane: after a carve, and for random planes, on whole grids and slabs."""
    X, Y, Z = dims
    sc = scenes.small_sphere(32, 5)
    s = np.float32(0.512 / max(dims))
    want = oracle.carve(X, Y, Z, s, sc.M, sc.masks)
    rng = np.random.default_rng(X + Y)
    with arvx.Context(X, Y, Z, s, z_range=zr) as ctx:
        ctx.set_views(sc.M, sc.masks)
        ctx.carve()
        mine = want if zr is None else want[zr[0]:zr[1]]
        occ, seen = ctx.download_planes()
        wocc, wseen = planes_of(mine)
        assert np.array_equal(occ, wocc) and np.array_equal(seen, wseen)
        assert_same(ctx.download_state(), mine, "bytes after planes")
        st = rng.choice(np.array([0, 1, 2, 3], np.uint8), size=mine.shape)
        ctx.upload_planes(*planes_of(st))
        assert_same(ctx.download_state(), st, "uploaded planes")
        o2, s2 = ctx.download_planes()
        assert np.array_equal(o2, planes_of(st)[0]) and np.array_equal(s2, planes_of(st)[1])
        ctx.carve()  # and the carve continues from them
        again = oracle.carve_planes(X, Y, s, sc.M, sc.masks, ctx.planes, state=st)
        assert_same(ctx.download_state(), again, "carve after uploaded planes")


@pytest.mark.parametrize("dims,zr", [((64, 8, 8), None), ((33, 17, 9), None), ((100, 40, 24), (5, 17))])
def test_byte_plane_is_an_exchange_form(arvx, oracle, dims, zr):
    """The device keeps records (2 bits per voxel) + a paint bit plane; the byte plane of
    arvx_state_upload / _download / _device_ptr is converted inside those calls.  Bits 0..2
    survive an upload -> download round trip (bit2 = painted by a host Model); the paint is
    dropped by the next carve, as before; arvx_state_device_ptr is a SNAPSHOT: its content is
    that of the moment of the call, and querying again after a carve gives the new state."""
    X, Y, Z = dims
    sc = scenes.small_sphere(32, 5)
    s = np.float32(0.512 / max(dims))
    rng = np.random.default_rng(X * 3 + Y)
    with arvx.Context(X, Y, Z, s, z_range=zr) as ctx:
        nz = Z if zr is None else zr[1] - zr[0]
        st = rng.integers(0, 8, size=(nz, Y, X), dtype=np.uint8)
        ctx.upload_state(st)
        assert_same(ctx.download_state(), st, "round trip with bit2")
        st2 = st & 3
        ctx.upload_state(st2)
        assert_same(ctx.download_state(), st2, "second upload replaces the paint")
        ctx.set_views(sc.M, sc.masks)
        n = ctx.nvox
        ptr = ctx.state_device_ptr()
        ctx.synchronize()
        snap0 = _device_bytes(ptr, n).reshape(st2.shape)
        assert_same(snap0, st2, "snapshot before the carve")
        ctx.upload_state(st)  # painted again
        ctx.carve()
        want = oracle.carve_planes(X, Y, s, sc.M, sc.masks, ctx.planes, state=st2)
        assert_same(ctx.download_state(), want, "carve drops the paint, keeps bits 0..1")
        ptr2 = ctx.state_device_ptr()  # re-queried: the state after the carve
        ctx.synchronize()
        assert_same(_device_bytes(ptr2, n).reshape(st2.shape), want, "snapshot after the carve")


def _device_bytes(ptr, n):
    """n bytes at device address ptr -> numpy (hipMemcpy through torch's HIP runtime)."""
    import ctypes
    out = np.empty(n, np.uint8)
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    rc = hip.hipMemcpy(out.ctypes.data, ctypes.c_void_p(ptr), n, 2)  # hipMemcpyDeviceToHost
    assert rc == 0, rc
    return out


@pytest.mark.parametrize("dims,kw", [((128, 96, 72), {}), ((72, 64, 80), {}),
                                     ((128, 128, 128), {"z_range": (40, 104)}),
                                     ((64, 64, 128), {"stripes": (2, 1)})])
def test_lazy_state_readers_agree_with_the_written_records(arvx, oracle, dims, kw):
    """After the carve of a fresh model the coarse tiles it settled as a whole exist only as
    their code (csrc/arvx_device.h, lazy state).  Every reader that takes the codes -- byte
    download, bit-plane download, occupancy packing (both row widths), the colour pass's surface
    -- must say what the same state says once it has been written out (arvx_export_model forces
    that), and what the oracle says."""
    import torch
    X, Y, Z = dims
    sc = scenes.syn.sphere_scene(max(dims), 6, W=320, H=240, with_images=True)
    with arvx.Context(X, Y, Z, sc.voxel_size, **kw) as ctx:
        want = oracle.carve_planes(X, Y, sc.voxel_size, sc.M, sc.masks, ctx.planes)
        ctx.set_views(sc.M, sc.masks, campos=sc.campos)
        ctx.carve()
        lazy = dict(state=ctx.download_state(), planes=ctx.download_planes())
        nw = (X * Y * len(ctx.planes) + 31) // 32
        words = torch.zeros(nw, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.pack_occupancy(words.data_ptr())
        ctx.synchronize()
        lazy["packed"] = words.cpu().numpy().copy()
        if "stripes" not in kw:
            ctx.set_images(sc.images)
            ctx.color(arvx.COLOR_AVERAGE)
            lazy["surface"] = ctx.surface()
        ctx.export_model(False)  # a stage that wants records: the lazy tiles are written out
        full = dict(state=ctx.download_state(), planes=ctx.download_planes())
        words.zero_()
        torch.cuda.synchronize()
        ctx.pack_occupancy(words.data_ptr())
        ctx.synchronize()
        full["packed"] = words.cpu().numpy().copy()
        if "stripes" not in kw:
            ctx.color(arvx.COLOR_AVERAGE)
            full["surface"] = ctx.surface()
    assert_same(lazy["state"], want, "lazy download vs oracle")
    assert_same(full["state"], want, "written-out download vs oracle")
    for k in ("planes", "surface"):
        if k in lazy:
            assert all(np.array_equal(a, b) for a, b in zip(lazy[k], full[k])), k
    assert np.array_equal(lazy["packed"], full["packed"])
    bits = np.packbits((want.reshape(-1) & 1).astype(np.uint8), bitorder="little")
    got = lazy["packed"].view(np.uint8)[:len(bits)]
    assert np.array_equal(got, bits)
    assert 0.0 < (want & 1).mean() < 1.0 and (want == 2).mean() > 0.3  # big carved regions: lazy tiles
