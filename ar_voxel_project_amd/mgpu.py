"""ctypes binding of libarvx_mgpu.so (include/arvx/arvx_mgpu.h): the carve over several GPUs
from ONE process, merged by one RCCL collective.  Test / tool plumbing, like capi.py."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libarvx_mgpu.so")
MERGE_ALLREDUCE, MERGE_COMPRESSED = 0, 1
SYMBOLS = ["arvx_mgpu_create", "arvx_mgpu_destroy", "arvx_mgpu_devices", "arvx_mgpu_set_views",
           "arvx_mgpu_state_reset", "arvx_mgpu_state_upload_planes", "arvx_mgpu_carve",
           "arvx_mgpu_occupancy_device_ptr", "arvx_mgpu_occupancy_download",
           "arvx_mgpu_state_download_planes", "arvx_mgpu_context", "arvx_mgpu_last_times"]
_lib = None


def load_library() -> C.CDLL:
    global _lib
    if _lib is None:
        capi.load_library()  # libarvx.so first (and torch's HIP runtime before it)
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        _lib = C.CDLL(LIB_PATH)
        for name in SYMBOLS:
            getattr(_lib, name).restype = C.c_int
    return _lib


class MultiGpu:
    def __init__(self, devices, X, Y, Z, voxel_size):
        self._lib = load_library()
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*devices)
        self._ck(self._lib.arvx_mgpu_create(C.byref(self._h), devs, len(devices), X, Y, Z,
                                            C.c_float(voxel_size)))
        self.X, self.Y, self.Z, self.n = X, Y, Z, len(devices)

    def _ck(self, rc):
        if rc != 0:
            raise capi.ArvxError(rc, "arvx_mgpu call failed (see stderr)")

    def close(self):
        if self._h:
            self._lib.arvx_mgpu_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_views(self, M, masks, campos=None):
        M = np.ascontiguousarray(M, np.float32).reshape(-1, 12)
        masks = np.ascontiguousarray(masks, np.uint8)
        if masks.ndim == 3:
            masks = masks[..., None]
        V, H, W, Cn = masks.shape
        ptrs = (C.c_void_p * V)(*[masks[i].ctypes.data for i in range(V)])
        cp = None
        if campos is not None:
            campos = np.ascontiguousarray(campos, np.float32).reshape(V, 3)
            cp = campos.ctypes.data_as(C.c_void_p)
        self._ck(self._lib.arvx_mgpu_set_views(self._h, V, M.ctypes.data_as(C.c_void_p), cp, ptrs,
                                               W, H, Cn, C.c_size_t(W * Cn)))

    def reset(self):
        self._ck(self._lib.arvx_mgpu_state_reset(self._h))

    def carve(self, merge=MERGE_ALLREDUCE, flags=0) -> bool:
        fb = C.c_int(0)
        self._ck(self._lib.arvx_mgpu_carve(self._h, flags, merge, C.byref(fb)))
        return bool(fb.value)

    def occupancy(self) -> np.ndarray:
        out = np.empty((self.X * self.Y * self.Z + 31) // 32, np.uint32)
        self._ck(self._lib.arvx_mgpu_occupancy_download(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def planes(self):
        n = ((self.X + 31) // 32) * self.Y * self.Z
        occ, seen = np.empty(n, np.uint32), np.empty(n, np.uint32)
        self._ck(self._lib.arvx_mgpu_state_download_planes(
            self._h, occ.ctypes.data_as(C.c_void_p), seen.ctypes.data_as(C.c_void_p)))
        return occ, seen

    def upload_planes(self, occ, seen):
        occ = np.ascontiguousarray(occ, np.uint32)
        seen = np.ascontiguousarray(seen, np.uint32)
        self._ck(self._lib.arvx_mgpu_state_upload_planes(
            self._h, occ.ctypes.data_as(C.c_void_p), seen.ctypes.data_as(C.c_void_p)))

    def times(self):
        a, b = C.c_float(), C.c_float()
        self._ck(self._lib.arvx_mgpu_last_times(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value
