"""Deterministic synthetic scenes: analytic silhouettes seen
by a ring of pinhole cameras.  Pure data generation -- no carving happens here.

World frame follows the reference's Model::toWord (src/Model.h:134-140): voxel
(x, y, z) sits at world (y*s, x*s, -z*s), so a grid of extent E spans
[0,E] x [0,E] x [-E,0] and "up" is world -Z.
"""
from __future__ import annotations

import numpy as np

# Data/*/cameracalibration.yml:7-12 of the reference (both data sets share it)
K_DATASET = np.array(
    [[4.9650601017248454e02, 0.0, 3.1217886794504733e02],
     [0.0, 4.9678498089444867e02, 2.5089544695238374e02],
     [0.0, 0.0, 1.0]], dtype=np.float64)
IMAGE_W, IMAGE_H = 640, 480
EXTENT = 0.512  # metres; voxel edge = EXTENT / N


def look_at_rt(cam_center, target, up=(0.0, 0.0, -1.0)) -> np.ndarray:
    """World->camera [R|t] (3x4, float64): x right, y down, z forward."""
    c = np.asarray(cam_center, np.float64)
    f = np.asarray(target, np.float64) - c
    f /= np.linalg.norm(f)
    upv = np.asarray(up, np.float64)
    xr = np.cross(f, upv)
    xr /= np.linalg.norm(xr)
    yd = np.cross(f, xr)
    R = np.stack([xr, yd, f])
    t = -R @ c
    return np.concatenate([R, t[:, None]], axis=1)


def ring_cameras(V: int, extent: float = EXTENT, dist_factor: float = 2.0,
                 elevations=(25.0, 40.0), jitter_deg: float = 0.0, seed: int = 0):
    """V cameras on a ring around the grid centre, azimuth 2*pi*i/V, elevation
    alternating, looking at the centre.  Returns (Rt float32 (V,3,4), centre)."""
    E = extent
    centre = np.array([E / 2, E / 2, -E / 2])
    rng = np.random.default_rng(seed)
    out = np.empty((V, 3, 4), np.float64)
    for i in range(V):
        az = 2.0 * np.pi * i / V
        el = np.deg2rad(elevations[i % len(elevations)])
        if jitter_deg:
            az += np.deg2rad(rng.uniform(-jitter_deg, jitter_deg))
            el += np.deg2rad(rng.uniform(-jitter_deg, jitter_deg))
        d = dist_factor * E
        cam = centre + d * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az),
                                     -np.sin(el)])
        out[i] = look_at_rt(cam, centre)
    return out.astype(np.float32), centre


def compose_m(K, Rt) -> np.ndarray:
    """M = K*[R|t] in float32, a0*b0 + a1*b1 + a2*b2 left to right, unfused: the
    value `intr * pose` has in the reference (cv::gemm small-matrix branch)."""
    K = np.asarray(K, np.float32)
    Rt = np.asarray(Rt, np.float32)
    lead = Rt.shape[:-2]
    Rt2 = Rt.reshape(-1, 3, 4)
    M = np.empty_like(Rt2)
    for r in range(3):
        t = K[r, 0] * Rt2[:, 0, :]
        t = (t + K[r, 1] * Rt2[:, 1, :]).astype(np.float32)
        t = (t + K[r, 2] * Rt2[:, 2, :]).astype(np.float32)
        M[:, r, :] = t
    return M.reshape(*lead, 3, 4)


def campos_from_rt(Rt) -> np.ndarray:
    """What the reference's colour pass uses as camera position: the translation
    column of the world->camera matrix (src/ColorReconstruction.h:21)."""
    Rt = np.asarray(Rt, np.float32).reshape(-1, 3, 4)
    return np.ascontiguousarray(Rt[:, :, 3])


def _rays(K, Rt, W, H):
    K = np.asarray(K, np.float64)
    Rt = np.asarray(Rt, np.float64)
    R, t = Rt[:, :3], Rt[:, 3]
    C = -R.T @ t
    u = np.arange(W, dtype=np.float64)
    v = np.arange(H, dtype=np.float64)
    dx = (u[None, :] - K[0, 2]) / K[0, 0]
    dy = (v[:, None] - K[1, 2]) / K[1, 1]
    dc = np.stack([np.broadcast_to(dx, (H, W)), np.broadcast_to(dy, (H, W)),
                   np.ones((H, W))], axis=-1)
    dw = dc @ R  # R^T applied to each ray
    return C, dw


def sphere_masks(K, Rt, centre, radius, W=IMAGE_W, H=IMAGE_H) -> np.ndarray:
    """(V,H,W) uint8, 255 where the pixel's ray hits the sphere, else 0."""
    K32 = np.asarray(K, np.float32).astype(np.float64)
    Rt = np.asarray(Rt, np.float32).astype(np.float64).reshape(-1, 3, 4)
    out = np.zeros((Rt.shape[0], H, W), np.uint8)
    for i in range(Rt.shape[0]):
        C, d = _rays(K32, Rt[i], W, H)
        oc = np.asarray(centre, np.float64) - C
        b = d @ oc
        dd = np.einsum("hwc,hwc->hw", d, d)
        disc = b * b - dd * (oc @ oc - radius * radius)
        out[i] = np.where((disc >= 0) & (b > 0), 255, 0).astype(np.uint8)
    return out


def box_masks(K, Rt, lo, hi, W=IMAGE_W, H=IMAGE_H) -> np.ndarray:
    """(V,H,W) uint8 silhouettes of the axis-aligned world box [lo,hi] (slab test)."""
    K32 = np.asarray(K, np.float32).astype(np.float64)
    Rt = np.asarray(Rt, np.float32).astype(np.float64).reshape(-1, 3, 4)
    lo = np.asarray(lo, np.float64)
    hi = np.asarray(hi, np.float64)
    out = np.zeros((Rt.shape[0], H, W), np.uint8)
    for i in range(Rt.shape[0]):
        C, d = _rays(K32, Rt[i], W, H)
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (lo - C) / d
            t2 = (hi - C) / d
        tn = np.nanmax(np.minimum(t1, t2), axis=-1)
        tf = np.nanmin(np.maximum(t1, t2), axis=-1)
        out[i] = np.where((tn <= tf) & (tf > 0), 255, 0).astype(np.uint8)
    return out


def pattern_images(V: int, W=IMAGE_W, H=IMAGE_H, seed: int = 1) -> np.ndarray:
    """(V,H,W,3) uint8 BGR test images: smooth gradients plus noise."""
    rng = np.random.default_rng(seed)
    u = np.arange(W)[None, :, None]
    v = np.arange(H)[:, None, None]
    out = np.empty((V, H, W, 3), np.uint8)
    for i in range(V):
        base = (u * (3 + i) + v * (5 + 2 * i) + np.array([0, 85, 170])[None, None, :] + 17 * i)
        noise = rng.integers(0, 64, size=(H, W, 3))
        out[i] = ((base + noise) & 255).astype(np.uint8)
    return out


class Scene:
    """Inputs of one carving job, already in the form the boundary takes."""

    def __init__(self, N, M, Rt, masks, voxel_size, images=None, K=None):
        self.X, self.Y, self.Z = (N, N, N) if np.isscalar(N) else tuple(int(n) for n in N)
        self.M = np.ascontiguousarray(M, np.float32)
        self.Rt = np.ascontiguousarray(Rt, np.float32)
        self.masks = np.ascontiguousarray(masks, np.uint8)
        self.voxel_size = np.float32(voxel_size)
        self.images = images
        self.K = K
        self.V = self.M.shape[0]
        self.H, self.W = self.masks.shape[1:3]

    @property
    def campos(self):
        return campos_from_rt(self.Rt)


def sphere_scene(N: int, V: int, W=IMAGE_W, H=IMAGE_H, extent: float = EXTENT,
                 radius_factor: float = 0.35, with_images: bool = False,
                 jitter_deg: float = 0.0, seed: int = 0) -> Scene:
    """The SURVEY 8(d) synthetic configuration: N^3 grid of extent 0.512 m, sphere
    of radius 0.35*E at the grid centre, V ring cameras at distance 2E."""
    K = K_DATASET.copy()
    if (W, H) != (IMAGE_W, IMAGE_H):  # scaled intrinsics for small test images
        K[0] *= W / IMAGE_W
        K[1] *= H / IMAGE_H
    K32 = K.astype(np.float32)
    Rt, centre = ring_cameras(V, extent, jitter_deg=jitter_deg, seed=seed)
    M = compose_m(K32, Rt)
    masks = sphere_masks(K32, Rt, centre, radius_factor * extent, W, H)
    images = pattern_images(V, W, H) if with_images else None
    return Scene(N, M, Rt, masks, np.float32(extent / N), images, K32)


def box_scene(N, V: int, W=IMAGE_W, H=IMAGE_H, extent: float = EXTENT,
              with_images: bool = False) -> Scene:
    K = K_DATASET.copy()
    if (W, H) != (IMAGE_W, IMAGE_H):
        K[0] *= W / IMAGE_W
        K[1] *= H / IMAGE_H
    K32 = K.astype(np.float32)
    Rt, centre = ring_cameras(V, extent)
    M = compose_m(K32, Rt)
    lo = centre - np.array([0.22, 0.15, 0.18]) * extent
    hi = centre + np.array([0.22, 0.15, 0.18]) * extent
    masks = box_masks(K32, Rt, lo, hi, W, H)
    images = pattern_images(V, W, H) if with_images else None
    n = N if np.isscalar(N) else max(N)
    return Scene(N, M, Rt, masks, np.float32(extent / n), images, K32)
