"""ctypes binding of libarvx.so (include/arvx/arvx.h).

This is plumbing: every call goes straight to the C-ABI, which launches the
gfx950 kernels.  There is no Python or CPU fallback -- if the shared library is
missing or the GPU is unusable the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARVX_LIB_PATH: A/B another build of the library on the same box (tools/ab_compare.sh)
LIB_PATH = os.environ.get("ARVX_LIB_PATH") or os.path.join(_HERE, "lib", "libarvx.so")

OCC = 1
SEEN = 2
CARVE_NO_CULL = 1
CARVE_STATS = 2
CARVE_FUSED = 8
CARVE_STREAM = 16
CARVE_FILTER = 64
CARVE_NO_STREAM = 32
# test plumbing: flags OR-ed into every carve of this process, e.g. ARVX_CARVE_EXTRA_FLAGS=16 runs a
# whole test module with the streaming carve forced on every fresh model (the library itself reads
# no environment variable)
_EXTRA_CARVE_FLAGS = int(os.environ.get("ARVX_CARVE_EXTRA_FLAGS", "0"))
COLOR_CLOSEST = 0
COLOR_AVERAGE = 1

# every symbol include/arvx/arvx.h declares
SYMBOLS = [
    "arvx_version", "arvx_last_error", "arvx_device_count", "arvx_projection_assoc",
    "arvx_ctx_create_slab_halo",
    "arvx_set_projection_assoc", "arvx_ctx_set_projection_assoc", "arvx_ctx_projection_assoc",
    "arvx_selftest_project", "arvx_selftest_depth", "arvx_selftest_view_tables",
    "arvx_ctx_create", "arvx_ctx_create_slab", "arvx_ctx_create_striped", "arvx_ctx_destroy",
    "arvx_ctx_set_stream", "arvx_ctx_set_exchange_stream", "arvx_ctx_synchronize", "arvx_ctx_voxels",
    "arvx_compose_projection", "arvx_set_views", "arvx_set_views_device",
    "arvx_set_images", "arvx_state_reset", "arvx_state_upload",
    "arvx_state_download", "arvx_state_device_ptr", "arvx_state_upload_halo",
    "arvx_state_upload_planes", "arvx_state_download_planes", "arvx_state_packet_geometry",
    "arvx_state_download_packets", "arvx_host_register",
    "arvx_host_unregister", "arvx_handle_unseen", "arvx_undistort", "arvx_undistort_device",
    "arvx_pack_occupancy", "arvx_pack_occupancy_global", "arvx_carve", "arvx_carve_views", "arvx_fast_carve",
    "arvx_color", "arvx_surface_count", "arvx_surface_download",
    "arvx_surface_depth_download", "arvx_color_samples",
    "arvx_colors_upload", "arvx_closure", "arvx_closure_count", "arvx_closure_download",
    "arvx_closure_download32",
    "arvx_mc_cells", "arvx_mc_cells_download", "arvx_mc_mesh", "arvx_mc_mesh_download",
    "arvx_mc_mesh_download_faces",
    "arvx_occupancy_packet_words", "arvx_occupancy_compress", "arvx_occupancy_expand",
    "arvx_occupancy_expand_striped", "arvx_occupancy_pack_compress", "arvx_occupancy_expand_striped_others",
    "arvx_export_model", "arvx_get_stats", "arvx_selftest_divide", "arvx_selftest_round",
]


class ArvxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"arvx error {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [
        ("subtiles", C.c_uint64),
        ("subtiles_carved", C.c_uint64),
        ("subtile_views_mixed", C.c_uint64),
        ("subtile_views_total", C.c_uint64),
        ("surface_voxels", C.c_uint64),
        ("reserved", C.c_uint64 * 3),
        ("host_total_fallbacks", C.c_uint64),
    ]


_libs: dict = {}
ASSOC_RIGHT, ASSOC_LEFT = 0, 1  # grouping of the M*world row sums (include/arvx/arvx.h)


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load libarvx.so (built in-tree by ar_voxel_project_amd.build), or another build
    of it at `path` (A/B builds)."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or make -C ar_voxel_project_amd/csrc); there is no CPU fallback")
    # torch (device memory / streams / torch.distributed in bench and tests) ships its
    # own HIP runtime; it must be loaded BEFORE libarvx.so pulls in /opt/rocm's, or a
    # later `import torch` finds "No HIP GPUs".  libarvx itself does not use torch.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    p = C.c_void_p
    f32p = C.POINTER(C.c_float)
    u8p = C.POINTER(C.c_uint8)
    lib.arvx_version.restype = C.c_int
    if hasattr(lib, "arvx_projection_assoc"):
        lib.arvx_projection_assoc.restype = C.c_int
    lib.arvx_last_error.restype = C.c_char_p
    lib.arvx_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.arvx_ctx_create.argtypes = [C.POINTER(p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]
    lib.arvx_ctx_create_slab.argtypes = [C.POINTER(p), C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_int, C.c_int]
    lib.arvx_ctx_create_striped.argtypes = [C.POINTER(p), C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_float, C.c_int, C.c_int]
    lib.arvx_ctx_destroy.argtypes = [p]
    lib.arvx_ctx_set_stream.argtypes = [p, p]
    if hasattr(lib, "arvx_ctx_set_exchange_stream"):
        lib.arvx_ctx_set_exchange_stream.argtypes = [p, p]
    lib.arvx_ctx_synchronize.argtypes = [p]
    lib.arvx_ctx_voxels.argtypes = [p, C.POINTER(C.c_int64)]
    lib.arvx_compose_projection.argtypes = [f32p, f32p, f32p]
    lib.arvx_set_views.argtypes = [p, C.c_int, f32p, f32p, C.POINTER(C.c_void_p), C.c_int,
                                   C.c_int, C.c_int, C.c_size_t]
    lib.arvx_set_views_device.argtypes = [p, C.c_int, f32p, f32p, p, C.c_int, C.c_int, C.c_int]
    lib.arvx_set_images.argtypes = [p, C.POINTER(C.c_void_p), C.c_size_t]
    lib.arvx_state_reset.argtypes = [p]
    lib.arvx_state_upload.argtypes = [p, u8p]
    lib.arvx_state_download.argtypes = [p, u8p]
    lib.arvx_state_device_ptr.argtypes = [p, C.POINTER(p), C.POINTER(C.c_size_t)]
    if hasattr(lib, "arvx_state_upload_planes"):
        lib.arvx_state_upload_planes.argtypes = [p, C.c_void_p, C.c_void_p]
        lib.arvx_state_download_planes.argtypes = [p, C.c_void_p, C.c_void_p]
    if hasattr(lib, "arvx_state_download_packets"):
        lib.arvx_state_packet_geometry.argtypes = [p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.arvx_state_download_packets.argtypes = [p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                                    C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        lib.arvx_host_register.argtypes = [C.c_void_p, C.c_size_t]
        lib.arvx_host_unregister.argtypes = [C.c_void_p]
        lib.arvx_handle_unseen.argtypes = [p]
    lib.arvx_state_upload_halo.argtypes = [p, u8p, u8p]
    lib.arvx_pack_occupancy.argtypes = [p, p]
    lib.arvx_pack_occupancy_global.argtypes = [p, p]
    lib.arvx_carve.argtypes = [p, C.c_uint]
    lib.arvx_carve_views.argtypes = [p, C.c_int, C.c_int, C.c_uint]
    lib.arvx_fast_carve.argtypes = [p]
    lib.arvx_color.argtypes = [p, C.c_int]
    lib.arvx_surface_count.argtypes = [p, C.POINTER(C.c_int64)]
    lib.arvx_surface_download.argtypes = [p, C.POINTER(C.c_int64), f32p]
    lib.arvx_surface_depth_download.argtypes = [p, f32p]
    lib.arvx_colors_upload.argtypes = [p, C.c_int64, C.POINTER(C.c_int64), f32p]
    lib.arvx_closure.argtypes = [p, C.c_int, C.c_int]
    lib.arvx_closure_count.argtypes = [p, C.POINTER(C.c_int64)]
    lib.arvx_closure_download.argtypes = [p, C.POINTER(C.c_int64), f32p]
    lib.arvx_export_model.argtypes = [p, f32p, C.c_int]
    lib.arvx_get_stats.argtypes = [p, C.POINTER(Stats)]
    ab_build = bool(os.environ.get("ARVX_LIB_PATH"))  # an older build may lack newer symbols
    if hasattr(lib, "arvx_selftest_divide") or not ab_build:
        lib.arvx_selftest_divide.argtypes = [p, C.c_int64, f32p, f32p, f32p, f32p]
    if hasattr(lib, "arvx_selftest_round") or not ab_build:
        lib.arvx_selftest_round.argtypes = [p, C.POINTER(C.c_int64)]
    if hasattr(lib, "arvx_mc_cells") or not ab_build:
        lib.arvx_mc_cells.argtypes = [p, C.POINTER(C.c_int64)]
        lib.arvx_mc_cells_download.argtypes = [p, C.POINTER(C.c_int32)]
    if hasattr(lib, "arvx_occupancy_compress") or not ab_build:
        lib.arvx_occupancy_packet_words.argtypes = [C.c_int64, C.c_int64]
        lib.arvx_occupancy_compress.argtypes = [p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
        lib.arvx_occupancy_expand.argtypes = [p, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                              C.c_int64, C.c_void_p, C.c_void_p]
        if hasattr(lib, "arvx_occupancy_expand_striped") or not ab_build:
            lib.arvx_occupancy_expand_striped.argtypes = [p, C.c_void_p, C.c_int, C.c_int64,
                                                          C.c_int64, C.c_int64, C.c_void_p,
                                                          C.c_void_p]
        if hasattr(lib, "arvx_occupancy_pack_compress"):
            lib.arvx_occupancy_pack_compress.argtypes = [p, C.c_void_p, C.c_int64, C.c_void_p]
            lib.arvx_occupancy_expand_striped_others.argtypes = [p, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                                                 C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    for name in SYMBOLS:
        if ab_build and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        if name == "arvx_occupancy_packet_words":
            fn.restype = C.c_int64
        elif name not in ("arvx_last_error",):
            fn.restype = C.c_int
    _libs[path] = lib
    return lib


def occupancy_packet_words(n_words64: int, cap_words64: int) -> int:
    """64-bit words of one compressed-occupancy packet (header + room for cap mixed words)."""
    return int(load_library().arvx_occupancy_packet_words(n_words64, cap_words64))


def _check(rc: int, lib: Optional[C.CDLL] = None) -> None:
    if rc != 0:
        lib = lib or load_library()
        raise ArvxError(rc, lib.arvx_last_error().decode("utf-8", "replace"))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def compose_projection(K, Rt) -> np.ndarray:
    """M = K * Rt as the reference's `intr * pose` (fp32, unfused, left to right)."""
    K = _f32(K).reshape(9)
    Rt = _f32(Rt).reshape(12)
    M = np.empty(12, np.float32)
    _check(load_library().arvx_compose_projection(_fp(K), _fp(Rt), _fp(M)))
    return M.reshape(3, 4)


def stripe_planes(Z: int, world: int, rank: int) -> np.ndarray:
    """Global z of the planes a striped context holds, in local order."""
    groups = np.arange(rank, Z // 8, world)
    return (groups[:, None] * 8 + np.arange(8)[None, :]).reshape(-1)


class Context:
    """One voxel grid (or Z slab) on one GPU: thin wrapper of arvx_ctx."""

    def __init__(self, X: int, Y: int, Z: int, voxel_size: float, device: int = 0,
                 z_range: Optional[Sequence[int]] = None,
                 stripes: Optional[Sequence[int]] = None, lib_path: Optional[str] = None,
                 assoc: Optional[int] = None, halo: int = 1):
        """z_range=(z0,z1): contiguous slab.  stripes=(world, rank): 8-plane groups
        rank, rank+world, ... (load-balanced multi-GPU split).  lib_path: another build of
        the library (A/B).  assoc: ASSOC_LEFT / ASSOC_RIGHT, the grouping of the M*world row
        sums of this context (default: the library's, arvx_projection_assoc).  halo: planes a
        slab keeps (and recomputes) beyond its own on each inner side: arvx_ctx_create_slab_halo."""
        self._lib = load_library(lib_path)
        self._h = C.c_void_p()
        self.X, self.Y, self.Z = int(X), int(Y), int(Z)
        self.voxel_size = float(np.float32(voxel_size))
        if stripes is not None:
            world, rank = int(stripes[0]), int(stripes[1])
            self._ck(self._lib.arvx_ctx_create_striped(C.byref(self._h), device, X, Y, Z,
                                                     C.c_float(voxel_size), world, rank))
            self.planes = stripe_planes(Z, world, rank)
            self.z_range = None
        else:
            z0, z1 = (0, Z) if z_range is None else (int(z_range[0]), int(z_range[1]))
            self.z_range = (z0, z1)
            self._lib.arvx_ctx_create_slab_halo.argtypes = [
                C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                C.c_int, C.c_int]
            self._ck(self._lib.arvx_ctx_create_slab_halo(C.byref(self._h), device, X, Y, Z,
                                                       C.c_float(voxel_size), z0, z1, int(halo)))
            self.planes = np.arange(z0, z1)
        nz = len(self.planes)  # global z of every local plane
        self.shape = (nz, Y, X)  # numpy view of the state plane: [z][y][x]
        self.nvox = nz * Y * X
        self._keep = []
        if assoc is not None:
            self.set_assoc(assoc)

    def set_assoc(self, assoc: int) -> None:
        self._lib.arvx_ctx_set_projection_assoc.argtypes = [C.c_void_p, C.c_int]
        self._ck(self._lib.arvx_ctx_set_projection_assoc(self._h, int(assoc)))

    @property
    def assoc(self) -> int:
        a = C.c_int()
        self._lib.arvx_ctx_projection_assoc.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._ck(self._lib.arvx_ctx_projection_assoc(self._h, C.byref(a)))
        return a.value

    def selftest_project(self, M, voxel_size, xyz):
        """(rows (n, 3), uv (n, 2)): the fp32 rows of M * toWord(x, y, z) and the quotients, as
        the kernels compute them with this context's grouping."""
        xyz = np.ascontiguousarray(xyz, np.int32).reshape(-1, 3)
        n = len(xyz)
        out = np.empty(5 * n, np.float32)
        M = np.ascontiguousarray(M, np.float32).reshape(12)
        self._lib.arvx_selftest_project.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_float,
                                                    C.c_void_p, C.c_void_p]
        self._ck(self._lib.arvx_selftest_project(self._h, n, M.ctypes.data, C.c_float(voxel_size),
                                                 xyz.ctypes.data, out.ctypes.data))
        return out[:3 * n].reshape(n, 3), out[3 * n:].reshape(n, 2)

    def selftest_view_tables(self, view: int, W: int, H: int):
        """(background bit plane as a (H, W) bool array, summed-area table as (H + 1, W + 1)
        uint16) of one view, as derived by set_views (include/arvx/arvx.h)."""
        self._lib.arvx_selftest_view_tables.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                        C.POINTER(C.c_int)]
        ld = C.c_int()
        self._ck(self._lib.arvx_selftest_view_tables(self._h, view, None, None, C.byref(ld)))
        bits = np.zeros((W * H + 31) // 32, np.uint32)
        table = np.zeros((H + 1, ld.value), np.uint16)
        self._ck(self._lib.arvx_selftest_view_tables(self._h, view, bits.ctypes.data, table.ctypes.data,
                                                     C.byref(ld)))
        bg = np.unpackbits(bits.view(np.uint8), bitorder="little")[:W * H].reshape(H, W).astype(bool)
        return bg, table[:, :W + 1].copy()

    def selftest_depth(self, campos, voxel_size, xyz):
        xyz = np.ascontiguousarray(xyz, np.int32).reshape(-1, 3)
        out = np.empty(len(xyz), np.float32)
        c = np.ascontiguousarray(campos, np.float32).reshape(3)
        self._lib.arvx_selftest_depth.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_float,
                                                  C.c_void_p, C.c_void_p]
        self._ck(self._lib.arvx_selftest_depth(self._h, len(xyz), c.ctypes.data,
                                               C.c_float(voxel_size), xyz.ctypes.data,
                                               out.ctypes.data))
        return out

    def _ck(self, rc: int) -> None:
        _check(rc, self._lib)

    def close(self) -> None:
        if self._h:
            self._lib.arvx_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- views --
    def set_views(self, M, masks, campos=None) -> None:
        """M: (V,3,4) float32. masks: (V,H,W) or (V,H,W,C) uint8 host array."""
        M = _f32(M).reshape(-1, 12)
        masks = np.ascontiguousarray(masks, dtype=np.uint8)
        if masks.ndim == 3:
            masks = masks[..., None]
        V, H, W, Cn = masks.shape
        assert M.shape[0] == V
        ptrs = (C.c_void_p * V)(*[masks[i].ctypes.data for i in range(V)])
        cp = None
        if campos is not None:
            campos = _f32(campos).reshape(V, 3)
            cp = _fp(campos)
        self._ck(self._lib.arvx_set_views(self._h, V, _fp(M), cp, ptrs, W, H, Cn, W * Cn))
        self.V, self.W, self.H = V, W, H

    def set_views_device(self, M, dev_masks_ptr: int, W: int, H: int, Cn: int,
                         campos=None) -> None:
        M = _f32(M).reshape(-1, 12)
        V = M.shape[0]
        cp = None
        if campos is not None:
            campos = _f32(campos).reshape(V, 3)
            cp = _fp(campos)
        self._keep = [M, campos]
        self._ck(self._lib.arvx_set_views_device(self._h, V, _fp(M), cp,
                                               C.c_void_p(dev_masks_ptr), W, H, Cn))
        self.V, self.W, self.H = V, W, H

    def set_images(self, images) -> None:
        """images: (V,H,W,3) uint8 BGR, undistorted."""
        images = np.ascontiguousarray(images, dtype=np.uint8)
        V, H, W, Cn = images.shape
        assert Cn == 3 and V == self.V and H == self.H and W == self.W
        ptrs = (C.c_void_p * V)(*[images[i].ctypes.data for i in range(V)])
        self._ck(self._lib.arvx_set_images(self._h, ptrs, W * 3))

    # -- state --
    def reset(self) -> None:
        self._ck(self._lib.arvx_state_reset(self._h))

    def upload_state(self, state) -> None:
        state = np.ascontiguousarray(state, dtype=np.uint8).reshape(-1)
        assert state.size == self.nvox
        self._ck(self._lib.arvx_state_upload(self._h, state.ctypes.data_as(C.POINTER(C.c_uint8))))

    def download_state(self) -> np.ndarray:
        out = np.empty(self.nvox, np.uint8)
        self._ck(self._lib.arvx_state_download(self._h, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out.reshape(self.shape)

    def plane_words(self) -> int:
        return ((self.X + 31) // 32) * self.Y * len(self.planes)

    def download_planes(self, occ=None, seen=None):
        """(occ, seen) bit planes, uint32, rows padded to 32-bit words (into the caller's arrays,
        e.g. page-locked ones, when given)."""
        n = self.plane_words()
        if occ is None:
            occ, seen = np.empty(n, np.uint32), np.empty(n, np.uint32)
        assert occ.size == seen.size == n and occ.dtype == seen.dtype == np.uint32
        self._ck(self._lib.arvx_state_download_planes(self._h, occ.ctypes.data, seen.ctypes.data))
        return occ, seen

    def packet_geometry(self):
        """(n, H): 64-bit words per plane of the owned voxels, header words of a state packet."""
        n, h = C.c_int64(), C.c_int64()
        self._ck(self._lib.arvx_state_packet_geometry(self._h, C.byref(n), C.byref(h)))
        return int(n.value), int(h.value)

    def download_packets(self, occ=None, seen=None):
        """The owned voxels' occupancy and seen planes as compressed packets (arvx.h:
        arvx_state_download_packets) -> (occ_packet, seen_packet, occ_need, seen_need), uint64
        arrays of header + capacity words.  occ / seen: the caller's buffers (e.g. page-locked
        ones); too small a buffer is reported through need > capacity and the caller asks again."""
        n, H = self.packet_geometry()
        if occ is None:
            occ = np.empty(H + max(4096, n // 16), np.uint64)
        if seen is None:
            seen = np.empty(H + max(4096, n // 64), np.uint64)
        for _ in range(2):
            on, sn = C.c_int64(), C.c_int64()
            self._ck(self._lib.arvx_state_download_packets(self._h, occ.ctypes.data, occ.size - H,
                                                           seen.ctypes.data, seen.size - H,
                                                           C.byref(on), C.byref(sn)))
            if on.value <= occ.size - H and sn.value <= seen.size - H:
                break
            if on.value > occ.size - H:
                occ = np.empty(H + on.value, np.uint64)
            if sn.value > seen.size - H:
                seen = np.empty(H + sn.value, np.uint64)
        return occ, seen, int(on.value), int(sn.value)

    def upload_planes(self, occ, seen) -> None:
        occ = np.ascontiguousarray(occ, np.uint32)
        seen = np.ascontiguousarray(seen, np.uint32)
        assert occ.size == seen.size == self.plane_words()
        self._ck(self._lib.arvx_state_upload_planes(self._h, occ.ctypes.data, seen.ctypes.data))

    def undistort(self, images, K, dist) -> np.ndarray:
        """cv::undistort on (V,H,W[,C]) uint8 images; returns the same shape."""
        im = np.ascontiguousarray(images, np.uint8)
        shape = im.shape
        if im.ndim == 3:
            im = im[..., None]
        V, H, W, Cn = im.shape
        out = np.empty_like(im)
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        dist = np.ascontiguousarray(dist, np.float64).reshape(-1)
        sp = (C.c_void_p * V)(*[im[i].ctypes.data for i in range(V)])
        dp = (C.c_void_p * V)(*[out[i].ctypes.data for i in range(V)])
        self._lib.arvx_undistort.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_int,
                                             C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p,
                                             C.c_int, C.POINTER(C.c_void_p)]
        self._ck(self._lib.arvx_undistort(self._h, V, sp, W, H, Cn, W * Cn, K.ctypes.data,
                                          dist.ctypes.data if dist.size else None, dist.size, dp))
        return out.reshape(shape)

    def handle_unseen(self) -> None:
        self._ck(self._lib.arvx_handle_unseen(self._h))

    def state_device_ptr(self) -> int:
        ptr = C.c_void_p()
        n = C.c_size_t()
        self._ck(self._lib.arvx_state_device_ptr(self._h, C.byref(ptr), C.byref(n)))
        return int(ptr.value)

    def upload_halo(self, below=None, above=None) -> None:
        def ptr(a):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=np.uint8).reshape(-1)
            assert a.size == self.X * self.Y
            self._keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_uint8))
        self._ck(self._lib.arvx_state_upload_halo(self._h, ptr(below), ptr(above)))

    def pack_occupancy(self, dev_words_ptr: int) -> None:
        self._ck(self._lib.arvx_pack_occupancy(self._h, C.c_void_p(dev_words_ptr)))

    def pack_occupancy_global(self, dev_global_words_ptr: int) -> None:
        self._ck(self._lib.arvx_pack_occupancy_global(self._h, C.c_void_p(dev_global_words_ptr)))

    # -- compressed occupancy exchange (device pointers; see include/arvx/arvx.h) --
    def occupancy_compress(self, dev_words_ptr: int, n_words64: int, dev_packet_ptr: int,
                           cap_words64: int) -> None:
        self._ck(self._lib.arvx_occupancy_compress(self._h, C.c_void_p(dev_words_ptr), n_words64,
                                                 C.c_void_p(dev_packet_ptr), cap_words64))

    def occupancy_expand(self, dev_packets_ptr: int, world: int, self_rank: int, n_words64: int,
                         cap_words64: int, dev_full_ptr: int, dev_overflow_ptr: int) -> None:
        self._ck(self._lib.arvx_occupancy_expand(self._h, C.c_void_p(dev_packets_ptr), world,
                                               self_rank, n_words64, cap_words64,
                                               C.c_void_p(dev_full_ptr),
                                               C.c_void_p(dev_overflow_ptr)))

    def occupancy_expand_striped(self, dev_packets_ptr: int, world: int, n_words64: int,
                                 cap_words64: int, words_per_group: int, dev_full_ptr: int,
                                 dev_overflow_ptr: int) -> None:
        self._ck(self._lib.arvx_occupancy_expand_striped(
            self._h, C.c_void_p(dev_packets_ptr), world, n_words64, cap_words64, words_per_group,
            C.c_void_p(dev_full_ptr), C.c_void_p(dev_overflow_ptr)))

    def occupancy_pack_compress(self, dev_packet_ptr: int, cap_words64: int, dev_full_ptr: int = 0) -> None:
        """The context's planes -> their packet, straight from the state; with dev_full_ptr the
        rank's own words also go to their place in the whole grid's plane."""
        self._ck(self._lib.arvx_occupancy_pack_compress(self._h, C.c_void_p(dev_packet_ptr), cap_words64,
                                                      C.c_void_p(dev_full_ptr)))

    def occupancy_expand_striped_others(self, dev_packets_ptr: int, world: int, self_rank: int,
                                        n_words64: int, cap_words64: int, words_per_group: int,
                                        dev_full_ptr: int, dev_overflow_ptr: int) -> None:
        self._ck(self._lib.arvx_occupancy_expand_striped_others(
            self._h, C.c_void_p(dev_packets_ptr), world, self_rank, n_words64, cap_words64,
            words_per_group, C.c_void_p(dev_full_ptr), C.c_void_p(dev_overflow_ptr)))

    def set_stream(self, stream_ptr: int) -> None:
        self._ck(self._lib.arvx_ctx_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_exchange_stream(self, stream_ptr: int) -> None:
        """pack_occupancy*, occupancy_compress / _expand* launch on this stream (0: the
        context's); the caller orders it against the context's stream with events."""
        self._ck(self._lib.arvx_ctx_set_exchange_stream(self._h, C.c_void_p(stream_ptr)))

    def synchronize(self) -> None:
        self._ck(self._lib.arvx_ctx_synchronize(self._h))

    # -- hot path --
    def carve(self, flags: int = 0) -> None:
        self._ck(self._lib.arvx_carve(self._h, flags | _EXTRA_CARVE_FLAGS))

    def carve_views(self, first: int, count: int, flags: int = 0) -> None:
        self._ck(self._lib.arvx_carve_views(self._h, first, count, flags | _EXTRA_CARVE_FLAGS))

    def fast_carve(self) -> None:
        self._ck(self._lib.arvx_fast_carve(self._h))

    def color(self, mode: int) -> None:
        self._ck(self._lib.arvx_color(self._h, mode))

    def surface(self):
        n = C.c_int64()
        self._ck(self._lib.arvx_surface_count(self._h, C.byref(n)))
        idx = np.empty(n.value, np.int64)
        rgb = np.empty((n.value, 3), np.float32)
        if n.value:
            self._ck(self._lib.arvx_surface_download(
                self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), _fp(rgb)))
        return idx, rgb

    def surface_depth(self) -> np.ndarray:
        n = C.c_int64()
        self._ck(self._lib.arvx_surface_count(self._h, C.byref(n)))
        d = np.empty(n.value, np.float32)
        if n.value:
            self._ck(self._lib.arvx_surface_depth_download(self._h, _fp(d)))
        return d

    SAMPLE_DTYPE = np.dtype([("r", np.uint8), ("g", np.uint8), ("b", np.uint8), ("valid", np.uint8),
                             ("depth", np.float32)])

    def color_samples(self, index) -> np.ndarray:
        """The colour lists behind the vote (Model::addColor / getColors of the reference):
        (n, V) records r, g, b, valid, depth for n voxels (flat indices over the owned planes)."""
        index = np.ascontiguousarray(index, dtype=np.int64)
        out = np.zeros((len(index), self.V), self.SAMPLE_DTYPE)
        self._lib.arvx_color_samples.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]
        self._ck(self._lib.arvx_color_samples(self._h, len(index), index.ctypes.data, out.shape[1],
                                              out.ctypes.data))
        return out

    def upload_colors(self, index, rgb) -> None:
        index = np.ascontiguousarray(index, dtype=np.int64)
        rgb = _f32(rgb).reshape(-1, 3)
        assert len(index) == len(rgb)
        self._ck(self._lib.arvx_colors_upload(self._h, len(index),
                                            index.ctypes.data_as(C.POINTER(C.c_int64)), _fp(rgb)))

    def closure(self, kernel_size: int = 3, apply_unseen: bool = True, download: bool = True):
        """applyClosure; returns (flat indices of the filled voxels, their RGBA) -- or, with
        download=False, only their number (the list stays on the device)."""
        self._ck(self._lib.arvx_closure(self._h, kernel_size, int(apply_unseen)))
        n = C.c_int64()
        self._ck(self._lib.arvx_closure_count(self._h, C.byref(n)))
        if not download:
            return int(n.value)
        idx = np.empty(n.value, np.int64)
        rgba = np.empty((n.value, 4), np.float32)
        if n.value:
            self._ck(self._lib.arvx_closure_download(
                self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), _fp(rgba)))
            i32 = np.empty(n.value, np.int32)  # the 32-bit form must say the same
            r32 = np.empty_like(rgba)
            self._lib.arvx_closure_download32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            self._ck(self._lib.arvx_closure_download32(self._h, i32.ctypes.data, r32.ctypes.data))
            assert np.array_equal(i32, idx) and np.array_equal(r32.view(np.uint32), rgba.view(np.uint32))
        return idx, rgba

    def mc_cells(self) -> np.ndarray:
        """(n, 4) int32 -- x, y, z, cube index of the cells marchingCubes would
        triangulate, in its visiting order (x outermost, z innermost)."""
        n = C.c_int64()
        self._ck(self._lib.arvx_mc_cells(self._h, C.byref(n)))
        cells = np.empty((n.value, 4), np.int32)
        if n.value:
            self._ck(self._lib.arvx_mc_cells_download(
                self._h, cells.ctypes.data_as(C.POINTER(C.c_int32))))
        return cells

    def mc_mesh(self, apply_unseen: bool = False):
        """(verts (3T, 3) float32 in voxel units, face_rgb (T, 3) uint32) of marchingCubes()."""
        n = C.c_int64()
        self._lib.arvx_mc_mesh.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        self._lib.arvx_mc_mesh_download.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._ck(self._lib.arvx_mc_mesh(self._h, int(apply_unseen), C.byref(n)))
        verts = np.empty((3 * n.value, 3), np.float32)
        rgb = np.empty((n.value, 3), np.uint32)
        if n.value:
            self._ck(self._lib.arvx_mc_mesh_download(self._h, verts.ctypes.data, rgb.ctypes.data))
            # the face-record form must say the same: (3t, 3t+1, 3t+2, r, g, b) per triangle
            v2 = np.empty_like(verts)
            faces = np.empty((n.value, 6), np.uint32)
            self._lib.arvx_mc_mesh_download_faces.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            self._ck(self._lib.arvx_mc_mesh_download_faces(self._h, v2.ctypes.data,
                                                          faces.ctypes.data))
            t3 = 3 * np.arange(n.value, dtype=np.uint32)
            assert np.array_equal(v2, verts) and np.array_equal(faces[:, 3:], rgb)
            assert np.array_equal(faces[:, :3], t3[:, None] + np.arange(3, dtype=np.uint32))
        return verts, rgb

    def mc_mesh_count(self, apply_unseen: bool = False) -> int:
        """arvx_mc_mesh without the download: the triangles stay on the device."""
        n = C.c_int64()
        self._lib.arvx_mc_mesh.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        self._ck(self._lib.arvx_mc_mesh(self._h, int(apply_unseen), C.byref(n)))
        return int(n.value)

    def export_model(self, apply_unseen: bool = False) -> np.ndarray:
        out = np.empty((self.nvox, 4), np.float32)
        self._ck(self._lib.arvx_export_model(self._h, _fp(out), int(apply_unseen)))
        return out

    def selftest_divide(self, a0, a1, b) -> np.ndarray:
        a0, a1, b = _f32(a0).reshape(-1), _f32(a1).reshape(-1), _f32(b).reshape(-1)
        out = np.empty((len(b), 4), np.float32)
        self._ck(self._lib.arvx_selftest_divide(self._h, len(b), _fp(a0), _fp(a1), _fp(b), _fp(out)))
        return out

    def selftest_round(self) -> int:
        """Mismatches between the kernels' pixel rounding and std::round (must be 0)."""
        n = C.c_int64(-1)
        self._ck(self._lib.arvx_selftest_round(self._h, C.byref(n)))
        return int(n.value)

    def stats(self) -> dict:
        s = Stats()
        self._ck(self._lib.arvx_get_stats(self._h, C.byref(s)))
        out = {k: int(getattr(s, k)) for k, _ in Stats._fields_ if k != "reserved"}
        out["slices_evaluated"] = int(s.reserved[0])  # 256-voxel slices projected exactly
        out["open_voxels_in_slices"] = int(s.reserved[1])  # of those voxels, not yet finished
        # (sub-tile, view) pairs that went to the exact kernel although every open voxel
        # got the same answer there: what a perfect classifier would have decided
        out["mixed_pairs_uniform"] = int(s.reserved[2])
        out["host_total_fallbacks"] = int(s.host_total_fallbacks)
        return out
