"""CPU suite: the oracle against hand-computed answers, against a second
independent numpy restatement, against size-independent properties and against
the committed golden fixtures.  No GPU involved."""
import numpy as np
import pytest

from ar_voxel_project_amd import synthetic as syn
from tests import golden_io, np_restate as npr, scenes


# ---- known answers computed by hand -------------------------------------------

def kat_matrix(f=100.0, cx=50.0, cy=40.0):
    # a2 = -w2 = z*s ; a0 = f*w0 - cx*w2 ; a1 = f*w1 - cy*w2  with w=(y s, x s, -z s, 1)
    # => u = f*y/z + cx , v = f*x/z + cy   (exact for power-of-two s and small ints)
    return np.array([[f, 0, -cx, 0], [0, f, -cy, 0], [0, 0, -1, 0]], np.float32)


@pytest.mark.parametrize("xyz,expect", [
    ((3, 2, 4), (100, 115)),      # u = 100*2/4+50 = 100, v = 100*3/4+40 = 115
    ((0, 1, 8), (63, 40)),        # u = 62.5 -> 63 (half away from zero), v = 40
    ((1, 0, 8), (50, 53)),        # v = 52.5 -> 53
    ((0, 0, 1), (50, 40)),        # on the optical axis
    ((0, 0, 0), None),            # 0/0 -> NaN -> outside
    ((0, 3, 0), None),            # +inf -> outside
    ((0, 12, 8), None),           # u = 200 == W -> outside
    ((0, 11, 8), (188, 40)),      # u = 187.5 -> 188 < W
])
def test_projection_known_answers(oracle, xyz, expect):
    M = kat_matrix()
    x, y, z = xyz
    assert oracle.project(M, 0.25, x, y, z, 200, 150) == expect
    inside, px, py = npr.project(M, 0.25, np.array([x]), np.array([y]), np.array([z]), 200, 150)
    assert (bool(inside[0]), (int(px[0]), int(py[0])) if inside[0] else None) == \
        (expect is not None, expect)


def test_projection_negative_half_is_outside(oracle):
    # u = f*y/z + cx with cx = -0.5 - 12.5 -> y/z = 1/8: u = 12.5 - 13 = -0.5 -> round = -1
    M = kat_matrix(cx=-13.0)
    assert oracle.project(M, 0.25, 0, 1, 8, 200, 150) is None
    # u = 12.5 - 12.75 = -0.25 -> round = -0 -> pixel 0 (inside)
    M = kat_matrix(cx=-12.75)
    assert oracle.project(M, 0.25, 0, 1, 8, 200, 150) == (0, 40)


def test_world_axes_swap_and_sign(oracle):
    # row 0 picks world[0] = y*s, row 1 picks world[1] = x*s, row 2 = 1: (u,v) = (y s, x s)
    M = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    assert oracle.project(M, 1.0, 3, 7, 5, 20, 20) == (7, 3)
    # world[2] = -z*s
    M = np.array([[0, 0, -1, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    assert oracle.project(M, 1.0, 3, 7, 5, 20, 20) == (5, 3)


def test_tiny_carve_by_hand(oracle):
    """2x2x2 grid, one view that maps (x,y,z) -> pixel (y, x) of a 2x2 mask."""
    M = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 0, 1]], np.float32)
    mask = np.array([[0, 255], [255, 0]], np.uint8)  # mask[row=py=x][col=px=y]
    st = oracle.carve(2, 2, 2, 1.0, M[None], mask[None], threads=1)
    # voxel (x,y,z): background iff mask[x][y]==0 iff x==y; every voxel is seen
    want = np.empty((2, 2, 2), np.uint8)
    for z in range(2):
        for y in range(2):
            for x in range(2):
                want[z, y, x] = 2 if x == y else 3
    assert np.array_equal(st, want)


def test_compose_is_unfused_fp32(oracle):
    rng = np.random.default_rng(0)
    K = rng.normal(size=(3, 3)).astype(np.float32) * 300
    Rt = rng.normal(size=(50, 3, 4)).astype(np.float32)
    got = oracle.compose(K, Rt)
    want = syn.compose_m(K, Rt)
    assert np.array_equal(got, want)
    # and it is NOT what an fp64 product rounded once would give everywhere
    once = (K.astype(np.float64) @ Rt.astype(np.float64)).astype(np.float32)
    assert (once != got).any()


# ---- second restatement ------------------------------------------------------------

@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_matches_numpy_restatement_projection(oracle, seed):
    rng = np.random.default_rng(seed)
    _, _, Ms = scenes.random_cameras(4, 0.512, seed=seed, W=160, H=120, inside=True)
    s = np.float32(0.512 / 64)
    x, y, z = (rng.integers(0, 64, size=4000) for _ in range(3))
    for M in Ms:
        inside, px, py = npr.project(M, s, x, y, z, 160, 120)
        raw, u, v = npr.project_raw(M, s, x, y, z)
        for i in range(0, 4000, 7):
            got = oracle.project(M, s, x[i], y[i], z[i], 160, 120)
            want = (int(px[i]), int(py[i])) if inside[i] else None
            assert got == want
            o = oracle.project_raw(M, s, x[i], y[i], z[i])
            assert o[0] == raw[0][i] and o[1] == raw[1][i] and o[2] == raw[2][i]
            assert (o[3] == u[i] or (np.isnan(o[3]) and np.isnan(u[i])))


def test_oracle_matches_numpy_restatement_carve(oracle):
    N, V, W, H = 20, 5, 96, 72
    _, _, M = scenes.random_cameras(V, 0.512, seed=7, W=W, H=H, inside=True)
    for C in (1, 3):
        masks = scenes.noise_masks(V, H, W, C=C, block=2, seed=C)
        s = np.float32(0.512 / N)
        a = oracle.carve(N, N, N, s, M, masks, threads=1)
        b = npr.carve(N, N, N, s, M, masks)
        assert np.array_equal(a, b)


# ---- properties ----------------------------------------------------------------------

def test_carve_properties(oracle):
    sc = scenes.small_sphere(24, 6, W=96, H=72)
    N = 24
    st = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks, threads=1)
    assert not np.any(st == 0), "carved implies seen (reference src/VoxelCarving.cpp:50-54)"
    assert np.array_equal(st, oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks, threads=4))
    perm = np.random.default_rng(0).permutation(6)
    assert np.array_equal(st, oracle.carve(N, N, N, sc.voxel_size, sc.M[perm], sc.masks[perm],
                                           threads=1)), "view order must not matter"
    again = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks, state=st, threads=1)
    assert np.array_equal(again, st), "idempotent"
    # the sphere is inside its visual hull
    E, c, r = 0.512, np.array([0.256, 0.256, -0.256]), 0.35 * 0.512
    z, y, x = np.meshgrid(*(np.arange(N),) * 3, indexing="ij")
    s = float(sc.voxel_size)
    wpos = np.stack([y * s, x * s, -z * s], -1)
    inside_sphere = np.linalg.norm(wpos - c, axis=-1) < r - 2 * s
    assert np.all((st & 1)[inside_sphere] == 1)


def test_reference_shaped_port_equals_state_plane_carve(oracle):
    sc = scenes.small_sphere(16, 4, W=96, H=72)
    N = 16
    st = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks, threads=1)
    rgba, seen = oracle.carve_ref(N, N, N, sc.voxel_size, sc.K, sc.Rt, sc.masks)
    occ = rgba[:, 3] != 0
    seen_b = np.unpackbits(seen.view(np.uint8), bitorder="little")[:N ** 3].astype(bool)
    assert np.array_equal(occ, (st.ravel() & 1) == 1)
    assert np.array_equal(seen_b, (st.ravel() & 2) == 2)
    assert np.array_equal(rgba[occ], np.tile(np.float32([50, 168, 141, 1]), (occ.sum(), 1)))
    # a sampled slab equals the same planes of the full run
    r2, _ = oracle.carve_ref(N, N, N, sc.voxel_size, sc.K, sc.Rt, sc.masks, 5, 9)
    assert np.array_equal(r2, rgba.reshape(N, N * N, 4)[5:9].reshape(-1, 4))


def test_fast_carve_closed_form(oracle):
    """fastCarve == 6-connected component of carvable voxels containing the origin;
    seen == origin + carved + their 6-neighbours."""
    from scipy import ndimage
    sc = scenes.small_sphere(20, 5, W=96, H=72)
    N = 20
    dense = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks, threads=1)
    carvable = (dense & 1) == 0
    fast = oracle.fast_carve(N, N, N, sc.voxel_size, sc.M, sc.masks)
    lab, _ = ndimage.label(carvable)  # default structure = 6-connectivity
    comp = (lab == lab[0, 0, 0]) & carvable if carvable[0, 0, 0] else np.zeros_like(carvable)
    assert np.array_equal((fast & 1) == 0, comp)
    grown = ndimage.binary_dilation(comp) | comp
    grown[0, 0, 0] = True
    assert np.array_equal((fast & 2) == 2, grown)


def test_color_matches_numpy_restatement(oracle):
    sc = scenes.small_sphere(20, 5, W=96, H=72, with_images=True)
    N = 20
    st = oracle.carve(N, N, N, sc.voxel_size, sc.M, sc.masks, threads=1)
    model = oracle.model_from_state(st)
    for mode in (0, 1):
        a = oracle.color(N, N, N, sc.voxel_size, sc.M, sc.campos, sc.images, mode, model)
        b = npr.color(N, N, N, sc.voxel_size, sc.M, sc.campos, sc.images, mode, model)
        assert np.array_equal(a, b)
        changed = (a != model).any(axis=1)
        assert changed.sum() > 0
        assert np.all(a[changed, 3] == 1) and np.all(a[changed, :3] == np.round(a[changed, :3]))


def test_depth_uses_translation_column(oracle):
    # cameras[i] = translation of the world->camera matrix
    d = oracle.depth(np.float32([3, 4, 0]), 1.0, 0, 0, 0)
    assert d == 5.0
    d = oracle.depth(np.float32([0, 0, 0]), 0.5, 2, 4, 6)  # world = (2, 1, -3)
    assert d == np.float32(np.sqrt(np.float64(14.0)))


def test_closure_is_one_dilation(oracle):
    rng = np.random.default_rng(3)
    X, Y, Z = 9, 7, 6
    rgba = np.zeros((Z * Y * X, 4), np.float32)
    occ = rng.random(Z * Y * X) < 0.15
    rgba[occ, :3] = rng.integers(0, 256, size=(occ.sum(), 3))
    rgba[occ, 3] = 1
    a = oracle.closure(X, Y, Z, rgba)
    b = npr.closure(X, Y, Z, rgba)
    assert np.array_equal(a, b)
    assert np.array_equal(a[occ], rgba[occ])
    assert (a[:, 3] != 0).sum() > occ.sum()


def test_handle_unseen(oracle):
    st = np.array([0, 1, 2, 3], np.uint8)
    model = oracle.model_from_state(st)
    out = oracle.handle_unseen(st, model)
    assert np.array_equal(out, np.float32([[204, 0, 0, 1], [204, 0, 0, 1], [0, 0, 0, 0],
                                           [50, 168, 141, 1]]))


# ---- golden fixtures ---------------------------------------------------------------------

@pytest.mark.parametrize("name", golden_io.names())
def test_oracle_reproduces_golden(oracle, name):
    g = golden_io.load(name)
    X, Y, Z, s = g["X"], g["Y"], g["Z"], g["s"]
    st = oracle.carve(X, Y, Z, s, g["M"], g["masks"], threads=1)
    assert np.array_equal(st, g["state"])
    assert np.array_equal(oracle.fast_carve(X, Y, Z, s, g["M"], g["masks"]), g["fast_state"])
    model = oracle.model_from_state(st)
    avg = oracle.color(X, Y, Z, s, g["M"], g["campos"], g["images"], 1, model)
    clo = oracle.color(X, Y, Z, s, g["M"], g["campos"], g["images"], 0, model)
    assert np.array_equal(avg[:, :3], g["average_rgb"].astype(np.float32))
    assert np.array_equal(clo[:, :3], g["closest_rgb"].astype(np.float32))
    unseen = oracle.handle_unseen(st, avg)
    assert np.array_equal(unseen, g["final_rgba_after_unseen"])
    assert np.array_equal(oracle.closure(X, Y, Z, unseen), g["closed_rgba"])
    # the independent numpy restatement agrees with the fixture as well
    assert np.array_equal(npr.carve(X, Y, Z, s, g["M"], g["masks"]), g["state"])


def test_golden_fixtures_exist():
    assert set(golden_io.names()) >= {"sphere32", "box50x50x25", "noise24"}


# ---- marching-cubes cell walk (src/MarchingCubes.cpp:12-18, MarchingCubes.h:479-488,537-552)

def np_mc_cells(st):
    """Independent numpy restatement: pad with empty, combine the 8 corner planes."""
    Z, Y, X = st.shape
    occ = np.zeros((Z + 2, Y + 2, X + 2), bool)
    occ[1:-1, 1:-1, 1:-1] = (st & 1) == 1
    corner = [(1, 0, 0), (0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 1), (0, 0, 1), (0, 1, 1), (1, 1, 1)]
    idx = np.zeros((Z + 1, Y + 1, X + 1), np.int32)
    for i, (dx, dy, dz) in enumerate(corner):
        idx |= (~occ[dz:dz + Z + 1, dy:dy + Y + 1, dx:dx + X + 1]).astype(np.int32) << i
    zz, yy, xx = np.nonzero((idx != 0) & (idx != 255))
    order = np.lexsort((zz, yy, xx))  # x outermost, z innermost
    zz, yy, xx = zz[order], yy[order], xx[order]
    return np.stack([xx - 1, yy - 1, zz - 1, idx[zz, yy, xx]], axis=1).astype(np.int32)


def test_mc_cells_oracle_matches_numpy(oracle):
    rng = np.random.default_rng(11)
    for dims, fill in [((5, 4, 3), 0.4), ((16, 9, 12), 0.1), ((7, 7, 7), 1.0), ((6, 5, 4), 0.0)]:
        X, Y, Z = dims
        st = np.where(rng.random((Z, Y, X)) < fill, 3, 2).astype(np.uint8)
        got = oracle.mc_cells(X, Y, Z, oracle.model_from_state(st))
        assert np.array_equal(got, np_mc_cells(st))


def test_mc_cells_oracle_threshold_and_colours(oracle):
    """The walk looks at w only: colours do not matter, and any threshold in (0,1]
    gives the same cells as the 0.5 the reference passes."""
    rng = np.random.default_rng(12)
    X, Y, Z = 9, 8, 7
    st = np.where(rng.random((Z, Y, X)) < 0.3, 3, 2).astype(np.uint8)
    model = oracle.model_from_state(st)
    base = oracle.mc_cells(X, Y, Z, model)
    painted = model.copy()
    painted[:, :3] = rng.integers(0, 256, size=(X * Y * Z, 3))
    assert np.array_equal(oracle.mc_cells(X, Y, Z, painted), base)
    for thr in (0.01, 0.5, 1.0):
        assert np.array_equal(oracle.mc_cells(X, Y, Z, model, thr), base)


def test_assoc_kat_tells_the_groupings_apart(oracle):
    """proj[r] = float(((p0 + p1) + p2) + p3) by default (LEFT), float(p0 + ((p1 + p2) + p3))
    under "assoc_right" (what SURVEY 8(c)(3) recalls).  The KAT voxel lands on pixel 3
    (background: carved) under LEFT and on pixel 4 (foreground: seen, kept) under RIGHT; C
    oracle (both settings) and the numpy twin agree on which is which."""
    from tests import np_restate as npr, scenes
    X, Y, Z, s, M, masks, (tx, ty, tz), st_right, st_left = scenes.assoc_kat()
    assert oracle.assoc() == 1
    raw = oracle.project_raw(M[0], s, tx, ty, tz)
    assert raw[0] == np.float32(1.0) and raw[3] < 3.5
    assert oracle.project(M[0], s, tx, ty, tz, 8, 4) == (3, 1)
    left = oracle.carve(X, Y, Z, s, M, masks, threads=1)
    assert left[tz, ty, tx] == st_left
    assert np.array_equal(left, npr.carve(X, Y, Z, s, M, masks))
    with oracle.variant("assoc_right"):
        assert oracle.assoc() == 0
        raw = oracle.project_raw(M[0], s, tx, ty, tz)
        assert raw[0] == np.float32(1.0) + np.float32(2.0 ** -23) and raw[3] > 3.5
        assert oracle.project(M[0], s, tx, ty, tz, 8, 4) == (4, 1)
        right = oracle.carve(X, Y, Z, s, M, masks, threads=1)
    assert oracle.assoc() == 1
    assert right[tz, ty, tx] == st_right
    diff = np.argwhere(left != right)
    assert diff.tolist() == [[tz, ty, tx]]  # the only voxel the grouping decides
    a, _, _ = npr.project_raw(M[0], s, np.array(tx), np.array(ty), np.array(tz), assoc_left=False)
    assert a[0] == np.float32(1.0) + np.float32(2.0 ** -23)


# ---- the OpenCV pin (tools/pin_with_opencv.py --emit tests/golden/opencv_pin.json) ------------------


def check_pin(doc, oracle):
    """Holds the C oracle and the numpy twin to what an OpenCV computed on the probes of
    tests/pin_probes.py.  Returns the grouping the file records."""
    from tests import np_restate as npr, pin_probes
    assert doc["grouping"] in ("LEFT", "RIGHT"), f"cv::gemm sums its products in a way the library does not know: {doc}"
    left = doc["grouping"] == "LEFT"
    M, s, xyz = pin_probes.projection_probes()
    want = np.array(doc["projection_rows_bits"], np.uint32).reshape(-1, 3)
    with oracle.variant("assoc_left" if left else "assoc_right"):
        # the voxel the grouping decides (random probes differ only once in ~10^8)
        kat = oracle.project_raw(pin_probes.kat_matrix(), np.float32(1.0), 1, 1, 1)[:1].view(np.uint32)[0]
        assert int(kat) == doc["kat_row0_bits"], "the known-answer voxel contradicts the recorded grouping"
        for i, (x, y, z) in enumerate(xyz):
            got = oracle.project_raw(M, s, x, y, z)[:3].view(np.uint32)
            assert np.array_equal(got, want[i]), f"oracle rows of probe {i} differ from cv::gemm"
    a, _, _ = npr.project_raw(M, s, xyz[:, 0], xyz[:, 1], xyz[:, 2], assoc_left=left)
    assert np.array_equal(np.stack(a, 1).astype(np.float32).view(np.uint32), want), "numpy twin rows"
    campos, s, xyz = pin_probes.depth_probes()
    wantd = np.array(doc["depth_bits"], np.uint32)
    gotd = np.array([oracle.depth(campos, s, x, y, z) for x, y, z in xyz], np.float32).view(np.uint32)
    assert np.array_equal(gotd, wantd), "oracle depths differ from cv::norm"
    src, K, dist = pin_probes.undistort_probe()
    assert np.array_equal(npr.undistort(src, K, dist).reshape(-1),
                          np.array(doc["undistort_u8"], np.uint8)), "restated undistort differs from cv::undistort"
    return doc["grouping"]


def test_opencv_pin_file(oracle):
    """When an OpenCV host has committed its answers, the oracle is PINNED to them;
    a recorded grouping that is not the library's default fails here and says what to change."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "opencv_pin.json")
    if not os.path.exists(path):
        pytest.skip("no tests/golden/opencv_pin.json yet: run tools/pin_with_opencv.py --emit on an OpenCV host")
    grouping = check_pin(json.load(open(path)), oracle)
    default_left = oracle.assoc() == 1
    assert (grouping == "LEFT") == default_left, (
        f"OpenCV {grouping} but the oracle / library default is {'LEFT' if default_left else 'RIGHT'}: flip "
        "g_default_assoc (csrc/arvx_capi.hip), ARVX_ORACLE default (oracle/arvx_oracle.c) and np_restate")


@pytest.mark.parametrize("left", [True, False])
def test_opencv_pin_loop_with_a_stand_in(oracle, left):
    """The emit -> check loop itself, with a stand-in for cv2 that computes the probes with the numpy
    twin under either grouping (test plumbing: it says nothing about OpenCV, it shows that a pin file
    of either grouping is read, compared bit for bit and told apart)."""
    from tests import np_restate as npr, pin_probes

    class FakeCv2:
        __version__ = "stand-in"

        @staticmethod
        def gemm(A, B, alpha, C, beta):
            A64, B64 = A.astype(np.float64), B.astype(np.float64).reshape(4)
            out = []
            for r in range(A.shape[0]):
                p = A64[r] * B64
                out.append(np.float32(((p[0] + p[1]) + p[2]) + p[3] if left else p[0] + ((p[1] + p[2]) + p[3])))
            return np.array(out, np.float32).reshape(-1, 1)

        @staticmethod
        def norm(d):
            d = d.reshape(-1).astype(np.float64)
            return np.float32(np.sqrt(((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3]))

        @staticmethod
        def undistort(src, K, dist):
            return npr.undistort(src, K, dist)

    doc = pin_probes.emit(FakeCv2)
    assert doc["grouping"] == ("LEFT" if left else "RIGHT")
    assert check_pin(doc, oracle) == doc["grouping"]
    # a file of the OTHER grouping must not pass as this one
    doc2 = dict(doc, grouping="RIGHT" if left else "LEFT")
    with pytest.raises(AssertionError):
        check_pin(doc2, oracle)
