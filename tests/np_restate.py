"""Second, independent restatement of the path's arithmetic in vectorised numpy
(IEEE float32/float64 element ops), used to cross-check the C oracle.
Follows the same reference lines as oracle/arvx_oracle.c."""
import numpy as np

F32, F64 = np.float32, np.float64


def to_word(s, x, y, z):
    """Model::toWord, reference src/Model.h:134-140."""
    s = F32(s)
    wx = (np.asarray(y).astype(F32) * s).astype(F32)  # note the x/y swap
    wy = (np.asarray(x).astype(F32) * s).astype(F32)
    wz = ((-np.asarray(z)).astype(F32) * s).astype(F32)
    return wx, wy, wz


def project_raw(M, s, x, y, z, assoc_left=True):
    """proj = M * world via the cv::gemm generic path: exact f64 products, summed
    ((p0+p1)+p2)+p3 (`(s0+s1+s2+s3)*alpha`: the default; assoc_left=False: p0+((p1+p2)+p3)),
    rounded to f32; then the two f32 divides."""
    M = np.asarray(M, F32).reshape(3, 4).astype(F64)
    w0, w1, w2 = (a.astype(F64) for a in to_word(s, x, y, z))
    a = []
    for r in range(3):
        p0, p1, p2, p3 = M[r, 0] * w0, M[r, 1] * w1, M[r, 2] * w2, M[r, 3] * 1.0
        acc = ((p0 + p1) + p2) + p3 if assoc_left else p0 + ((p1 + p2) + p3)
        a.append(acc.astype(F32))
    with np.errstate(divide="ignore", invalid="ignore"):
        u = (a[0] / a[2]).astype(F32)
        v = (a[1] / a[2]).astype(F32)
    return a, u, v


def round_half_away(t):
    """std::round for float32 inputs, computed exactly in float64."""
    t = np.asarray(t, F64)
    return np.where(np.isfinite(t), np.copysign(np.floor(np.abs(t) + 0.5), t), np.nan)


def project(M, s, x, y, z, W, H):
    """-> (inside bool array, px, py)."""
    _, u, v = project_raw(M, s, x, y, z)
    ru, rv = round_half_away(u), round_half_away(v)
    inside = (ru >= 0) & (ru < W) & (rv >= 0) & (rv < H)  # NaN compares false
    px = np.where(inside, ru, 0).astype(np.int64)
    py = np.where(inside, rv, 0).astype(np.int64)
    return inside, px, py


def is_background(mask, px, py):
    m = np.asarray(mask)
    if m.ndim == 2:
        return m[py, px] == 0
    return (m[py, px, :] == 0).all(axis=-1)


def carve(X, Y, Z, s, Ms, masks, state=None):
    """Dense carve, all views, on a (Z,Y,X) state plane (bit0 occ, bit1 seen)."""
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    st = np.full((Z, Y, X), 1, np.uint8) if state is None else np.array(state, np.uint8)
    Ms = np.asarray(Ms, F32).reshape(-1, 3, 4)
    for i in range(Ms.shape[0]):
        inside, px, py = project(Ms[i], s, x, y, z, masks[i].shape[1], masks[i].shape[0])
        bg = is_background(masks[i], px, py) & inside
        st = np.where(inside, st | 2, st)
        st = np.where(bg, st & 0xFE, st).astype(np.uint8)
    return st


def depth(campos, s, x, y, z):
    """cv::norm(cameras[i] - word_coord): f32 differences, f64 sum of squares."""
    w0, w1, w2 = to_word(s, x, y, z)
    c = np.asarray(campos, F32)
    d0 = (c[0] - w0).astype(F32).astype(F64)
    d1 = (c[1] - w1).astype(F32).astype(F64)
    d2 = (c[2] - w2).astype(F32).astype(F64)
    d3 = F64(0.0)
    acc = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3
    return np.sqrt(acc).astype(F32)


def surface_mask(occ):
    """occupied and not isInner (reference src/Model.h:126-132; outside = empty)."""
    p = np.pad(occ, 1, constant_values=False)
    inner = (p[1:-1, 1:-1, :-2] & p[1:-1, 1:-1, 2:] & p[1:-1, :-2, 1:-1] & p[1:-1, 2:, 1:-1] &
             p[:-2, 1:-1, 1:-1] & p[2:, 1:-1, 1:-1])
    return occ & ~inner


def color(X, Y, Z, s, Ms, campos, images, mode, rgba):
    """Colour vote (reference src/ColorReconstruction.h:34-74, .cpp:22-70)."""
    rgba = np.array(rgba, F32).reshape(Z, Y, X, 4)
    occ = rgba[..., 3] != 0
    surf = surface_mask(occ)
    zs, ys, xs = np.nonzero(surf)
    Ms = np.asarray(Ms, F32).reshape(-1, 3, 4)
    n = np.zeros(len(zs), np.int64)
    ssum = np.zeros((len(zs), 3), F32)
    best = np.zeros((len(zs), 3), F32)
    bestd = np.full(len(zs), np.inf, F32)
    for i in range(Ms.shape[0]):
        H, W = images[i].shape[:2]
        inside, px, py = project(Ms[i], s, xs, ys, zs, W, H)
        bgr = images[i][py, px].astype(F32)
        rgb = bgr[:, ::-1]
        d = depth(campos[i], s, xs, ys, zs)
        take = inside & ((n == 0) | (d < bestd))
        best[take] = rgb[take]
        bestd[take] = d[take]
        ssum[inside] = (ssum[inside] + rgb[inside]).astype(F32)
        n += inside
    has = n > 0
    out = rgba.copy()
    if mode == 0:
        col = best
    else:
        with np.errstate(divide="ignore", invalid="ignore"):
            col = round_half_away((ssum / n[:, None].astype(F32)).astype(F32)).astype(F32)
    sel = (zs[has], ys[has], xs[has])
    out[sel + (slice(0, 3),)] = col[has]
    out[sel + (3,)] = 1.0
    return out.reshape(-1, 4)


def closure(X, Y, Z, rgba):
    """applyClosure(kernel 3) == one 3x3x3 dilation with colour mean."""
    a = np.array(rgba, F32).reshape(Z, Y, X, 4)
    occ = a[..., 3] > 0
    p = np.pad(np.where(occ[..., None], a, 0).astype(F32), ((1, 1), (1, 1), (1, 1), (0, 0)))
    c = np.pad(occ.astype(np.int32), 1)
    ssum = np.zeros_like(a)
    cnt = np.zeros((Z, Y, X), np.int32)
    # reference order: x offset outermost, then y, then z (sums of small ints: exact)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                sl = (slice(1 + dz, 1 + dz + Z), slice(1 + dy, 1 + dy + Y),
                      slice(1 + dx, 1 + dx + X))
                ssum = (ssum + p[sl]).astype(F32)
                cnt += c[sl]
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = (ssum / cnt[..., None].astype(F32)).astype(F32)
    out = np.where(occ[..., None], a, np.where(cnt[..., None] > 0, mean, F32(0)))
    return out.reshape(-1, 4).astype(F32)


# ---- marching cubes (src/MarchingCubes.cpp:8-31, src/MarchingCubes.h:414-578) ---------------

_MC_SECOND = [1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7]
_MC_CORNER = [(1, 0, 0), (0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 1), (0, 0, 1), (0, 1, 1), (1, 1, 1)]
_MODEL_RGB, _UNSEEN_RGB = (50.0, 168.0, 141.0), (204.0, 0.0, 0.0)


def mc_triangle_rows():
    """Bourke's triangle table from the shared data file (include/arvx/mc_triangles.inc)."""
    import os
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include",
                        "arvx", "mc_triangles.inc")
    src = open(path).read()
    rows = re.findall(r'"([0-9a-b]*)"', src[src.index("*/"):])
    assert len(rows) == 256
    return [[int(c, 16) for c in r] for r in rows]


def _vertex_interp(thr, p0, v0, p1, v1):
    """VertexInterp, src/MarchingCubes.h:428-468, in float32 element ops."""
    if v0[3] == 0 and v1[3] != 0:
        return p1.copy(), v1[:3].copy()
    if v0[3] != 0 and v1[3] == 0:
        return p0.copy(), v0[:3].copy()
    f = F32(0.5) if v0[3] == v1[3] else F32(F32(thr - v0[3]) / F32(v1[3] - v0[3]))
    g = F32(F32(1) - f)
    coord = (g * p0).astype(F32) + (f * p1).astype(F32)
    c0, c1 = v0[:3], v1[:3]
    if tuple(c0) in (_MODEL_RGB, _UNSEEN_RGB):
        col = c1.copy()
    elif tuple(c1) in (_MODEL_RGB, _UNSEEN_RGB):
        col = c0.copy()
    else:
        col = (g * c0).astype(F32) + (f * c1).astype(F32)
    return coord.astype(F32), col.astype(F32)


def mc_mesh(X, Y, Z, rgba, threshold=0.5):
    """-> (verts (3T,3) float32, face_rgb (T,3) int): the reference's cell walk, one Python
    loop per cell -- small grids only."""
    rows = mc_triangle_rows()
    thr = F32(threshold)
    vox = np.asarray(rgba, F32).reshape(Z, Y, X, 4)
    zero = np.zeros(4, F32)

    def get(x, y, z):  # Model::get: zero outside the grid
        if x < 0 or x >= X or y < 0 or y >= Y or z < 0 or z >= Z:
            return zero
        return vox[z, y, x]

    verts, faces = [], []
    for x in range(-1, X):
        for y in range(-1, Y):
            for z in range(-1, Z):
                val = [get(x + a, y + b, z + c) for a, b, c in _MC_CORNER]
                idx = sum(1 << i for i in range(8) if val[i][3] < thr)
                if idx in (0, 255):
                    continue
                pts = [np.array([x + a, y + b, z + c], F32) for a, b, c in _MC_CORNER]
                vl = {}
                for e in range(12):
                    a, b = e % 8, _MC_SECOND[e]
                    if ((idx >> a) & 1) != ((idx >> b) & 1):
                        vl[e] = _vertex_interp(thr, pts[a], val[a], pts[b], val[b])
                row = rows[idx]
                for t in range(0, len(row), 3):
                    e0, e1, e2 = row[t:t + 3]
                    verts += [vl[e0][0], vl[e1][0], vl[e2][0]]
                    c0, c1, c2 = vl[e0][1], vl[e1][1], vl[e1][1]  # the `i + 1` of :506
                    s = ((c0 + c1).astype(F32) + c2).astype(F32) / F32(3)
                    faces.append(np.floor(np.abs(s.astype(F64)) + 0.5).astype(np.int64))
    if not faces:
        return np.zeros((0, 3), F32), np.zeros((0, 3), np.int64)
    return np.array(verts, F32), np.array(faces, np.int64)


# ---- cv::undistort (OpenCV 4.x calib3d/imgproc, restated; see csrc/undistort_kernels.h) -------

def undistort(img, K, dist):
    """img (H,W[,C]) uint8 -> same shape: initUndistortRectifyMap (double) + remap with 5-bit
    fixed-point bilinear weights and a constant-0 border."""
    im = np.asarray(img, np.uint8)
    squeeze = im.ndim == 2
    if squeeze:
        im = im[..., None]
    H, W, C = im.shape
    K = np.asarray(K, F64).reshape(3, 3)
    k = np.zeros(8, F64)
    d = np.asarray(dist, F64).reshape(-1)
    k[:len(d)] = d
    k1, k2, p1, p2, k3, k4, k5, k6 = k
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    u = np.arange(W, dtype=F64)[None, :]
    v = np.arange(H, dtype=F64)[:, None]
    x = u * (1.0 / fx) + (-cx / fx) + 0 * v
    y = v * (1.0 / fy) + (-cy / fy) + 0 * u
    x2, y2 = x * x, y * y
    r2 = x2 + y2
    _2xy = 2 * x * y
    kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    xd = x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)
    yd = y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy
    iu = np.rint((fx * xd + cx) * 32.0).astype(np.int64)  # cvRound: half to even
    iv = np.rint((fy * yd + cy) * 32.0).astype(np.int64)
    sx = np.clip(iu >> 5, -32768, 32767)
    sy = np.clip(iv >> 5, -32768, 32767)
    a, b = iu & 31, iv & 31
    w = [(32 - a) * (32 - b) * 32, a * (32 - b) * 32, (32 - a) * b * 32, a * b * 32]

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
        t = im[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.int64)
        return np.where(ok[..., None], t, 0)

    acc = (w[0][..., None] * tap(sy, sx) + w[1][..., None] * tap(sy, sx + 1) +
           w[2][..., None] * tap(sy + 1, sx) + w[3][..., None] * tap(sy + 1, sx + 1))
    out = ((acc + (1 << 14)) >> 15).astype(np.uint8)
    return out[..., 0] if squeeze else out
