"""ctypes access to oracle/libarvx_oracle.so -- the CPU restatement used as the
CHECKER by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Never import this from ar_voxel_project_amd/."""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libarvx_oracle.so")
OCC, SEEN = 1, 2
_lib = None
ASSOC_RIGHT, ASSOC_LEFT = 0, 1  # grouping of the M*world row sums (arvx_oracle.c, G2)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} missing: run `make -C oracle`")
        L = C.CDLL(LIB_PATH)
        L.arvx_oracle_project.restype = C.c_int
        L.arvx_oracle_depth.restype = C.c_float
        L.arvx_oracle_assoc.restype = C.c_int
        _lib = L
    return _lib


@contextlib.contextmanager
def variant(name: str):
    """Run the enclosed calls with another grouping of the M*world row sums:
    "assoc_right" = p0+((p1+p2)+p3), "assoc_left" = ((p0+p1)+p2)+p3 (the default), "" = as is."""
    L = lib()
    old = int(L.arvx_oracle_assoc())
    if name:
        L.arvx_oracle_set_assoc({"assoc_left": 1, "assoc_right": 0}[name])
    try:
        yield L
    finally:
        L.arvx_oracle_set_assoc(old)


def assoc() -> int:
    return int(lib().arvx_oracle_assoc())


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _u8(a):
    return np.ascontiguousarray(a, np.uint8)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def compose(K, Rt):
    K = _f32(K).reshape(9)
    Rt = _f32(Rt).reshape(-1, 12)
    M = np.empty_like(Rt)
    for i in range(Rt.shape[0]):
        lib().arvx_oracle_compose(_p(K), _p(Rt[i]), _p(M[i]))
    return M.reshape(-1, 3, 4)


def project(M, s, x, y, z, W, H):
    M = _f32(M).reshape(12)
    px, py = C.c_int(), C.c_int()
    ok = lib().arvx_oracle_project(_p(M), C.c_float(s), int(x), int(y), int(z), W, H,
                                   C.byref(px), C.byref(py))
    return (px.value, py.value) if ok else None


def project_raw(M, s, x, y, z):
    M = _f32(M).reshape(12)
    out = np.empty(5, np.float32)
    lib().arvx_oracle_project_raw(_p(M), C.c_float(s), int(x), int(y), int(z), _p(out))
    return out


def _masks4(masks):
    m = _u8(masks)
    if m.ndim == 3:
        m = m[..., None]
    return m


def fresh_state(X, Y, Z):
    return np.full((Z, Y, X), OCC, np.uint8)


def carve(X, Y, Z, s, M, masks, state=None, threads=0):
    """Dense carve over all views; returns the (Z,Y,X) state plane."""
    M = _f32(M).reshape(-1, 12)
    m = _masks4(masks)
    V, H, W, Cn = m.shape
    st = fresh_state(X, Y, Z) if state is None else _u8(state).copy().reshape(Z, Y, X)
    args = [X, Y, Z, C.c_float(s), V, _p(M), _p(m), W, H, Cn, C.c_long(W * Cn), _p(st)]
    if threads == 1:
        lib().arvx_oracle_carve(*args)
    else:
        lib().arvx_oracle_carve_mt(*args, int(threads))
    return st


def carve_planes(X, Y, s, M, masks, planes, state=None, threads=0):
    """Dense carve of the global z planes `planes` only (a slab / one rank's stripes);
    returns (len(planes), Y, X)."""
    M = _f32(M).reshape(-1, 12)
    m = _masks4(masks)
    V, H, W, Cn = m.shape
    zl = np.ascontiguousarray(planes, np.int32)
    st = (np.full((len(zl), Y, X), OCC, np.uint8) if state is None
          else _u8(state).copy().reshape(len(zl), Y, X))
    lib().arvx_oracle_carve_planes_mt(X, Y, C.c_float(s), V, _p(M), _p(m), W, H, Cn,
                                      C.c_long(W * Cn), _p(zl), len(zl), _p(st), int(threads))
    return st


def carve_view(X, Y, Z, s, M, mask, state):
    M = _f32(M).reshape(12)
    m = _u8(mask)
    if m.ndim == 2:
        m = m[..., None]
    H, W, Cn = m.shape
    st = _u8(state).copy().reshape(Z, Y, X)
    lib().arvx_oracle_carve_view(X, Y, Z, C.c_float(s), _p(M), _p(m), W, H, Cn,
                                 C.c_long(W * Cn), _p(st))
    return st


def carve_ref(X, Y, Z, s, K, Rt, masks, zlo=0, zhi=None):
    """Reference-shaped baseline (AoS floats, x->y->z). Returns (rgba, seen_bits)."""
    K = _f32(K).reshape(9)
    Rt = _f32(Rt).reshape(-1, 12)
    m = _masks4(masks)
    V, H, W, Cn = m.shape
    zhi = Z if zhi is None else zhi
    N = X * Y * (zhi - zlo)  # arrays hold the sampled planes only
    rgba = np.empty((N, 4), np.float32)
    lib().arvx_oracle_model_init(_p(rgba), C.c_long(N))
    seen = np.zeros((N + 63) // 64, np.uint64)
    lib().arvx_oracle_carve_ref(X, Y, Z, C.c_float(s), V, _p(K), _p(Rt), _p(m), W, H, Cn,
                                C.c_long(W * Cn), _p(rgba), _p(seen), int(zlo), int(zhi))
    return rgba, seen


def fast_carve(X, Y, Z, s, M, masks, state=None):
    M = _f32(M).reshape(-1, 12)
    m = _masks4(masks)
    V, H, W, Cn = m.shape
    st = fresh_state(X, Y, Z) if state is None else _u8(state).copy().reshape(Z, Y, X)
    lib().arvx_oracle_fast_carve(X, Y, Z, C.c_float(s), V, _p(M), _p(m), W, H, Cn,
                                 C.c_long(W * Cn), _p(st))
    return st


def model_from_state(state):
    st = _u8(state).reshape(-1)
    rgba = np.empty((st.size, 4), np.float32)
    lib().arvx_oracle_state_to_model(_p(st), _p(rgba), C.c_long(st.size))
    return rgba


def handle_unseen(state, rgba):
    st = _u8(state).reshape(-1)
    out = _f32(rgba).copy()
    lib().arvx_oracle_handle_unseen(_p(st), _p(out), C.c_long(st.size))
    return out


def color(X, Y, Z, s, M, campos, images, mode, rgba):
    M = _f32(M).reshape(-1, 12)
    campos = _f32(campos).reshape(-1, 3)
    img = _u8(images)
    V, H, W, Cn = img.shape
    assert Cn == 3
    out = _f32(rgba).copy()
    lib().arvx_oracle_color(X, Y, Z, C.c_float(s), V, _p(M), _p(campos), _p(img), W, H,
                            C.c_long(W * 3), int(mode), _p(out))
    return out


def depth(campos, s, x, y, z):
    cp = _f32(campos).reshape(3)
    return float(lib().arvx_oracle_depth(_p(cp), C.c_float(s), int(x), int(y), int(z)))


def closure(X, Y, Z, rgba):
    out = _f32(rgba).copy()
    lib().arvx_oracle_closure(X, Y, Z, _p(out))
    return out


def mc_cells(X, Y, Z, rgba, threshold=0.5):
    """(n, 4) int32: x, y, z, cube index of the cells marchingCubes triangulates."""
    r = _f32(rgba)
    L = lib()
    L.arvx_oracle_mc_cells.restype = C.c_long
    n = L.arvx_oracle_mc_cells(X, Y, Z, _p(r), C.c_float(threshold), None, C.c_long(0))
    out = np.zeros((max(n, 1), 4), np.int32)
    L.arvx_oracle_mc_cells(X, Y, Z, _p(r), C.c_float(threshold), _p(out), C.c_long(n))
    return out[:n]


def mc_mesh(X, Y, Z, rgba, threshold=0.5):
    """marchingCubes' mesh: (verts (3T, 3) float32 in voxel units, face_rgb (T, 3) uint32);
    triangle t uses vertices 3t, 3t+1, 3t+2."""
    r = _f32(rgba)
    L = lib()
    L.arvx_oracle_mc_mesh.restype = C.c_long
    n = L.arvx_oracle_mc_mesh(X, Y, Z, _p(r), C.c_float(threshold), None, None, C.c_long(0))
    verts = np.zeros((max(n, 1) * 3, 3), np.float32)
    rgb = np.zeros((max(n, 1), 3), np.uint32)
    L.arvx_oracle_mc_mesh(X, Y, Z, _p(r), C.c_float(threshold), _p(verts), _p(rgb), C.c_long(n))
    return verts[:3 * n], rgb[:n]


def off_text(verts, face_rgb, scale_factor=1.0, translation=(0.0, 0.0, 0.0)):
    """SimpleMesh::WriteMesh (src/MarchingCubes.h:60-87): default ostream float format
    (= printf %g, 6 significant digits) of vertex * scaleFactor + translation in fp32."""
    v = np.asarray(verts, np.float32)
    t = np.asarray(translation, np.float32)
    out = (v * np.float32(scale_factor)).astype(np.float32) + t
    n = len(face_rgb)
    lines = ["OFF", f"{len(v)} {n} 0"]
    lines += ["%g %g %g" % (float(a), float(b), float(c)) for a, b, c in out]
    lines += ["3 %d %d %d %d %d %d" % (3 * i, 3 * i + 1, 3 * i + 2, r, g, b)
              for i, (r, g, b) in enumerate(np.asarray(face_rgb).tolist())]
    return "\n".join(lines) + "\n"
