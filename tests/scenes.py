"""Scene builders shared by the parity tests (inputs only, no carving)."""
import numpy as np

from ar_voxel_project_amd import synthetic as syn


def small_sphere(N=32, V=6, W=160, H=120, **kw):
    return syn.sphere_scene(N, V, W=W, H=H, **kw)


def noise_masks(V, H, W, C=1, p_bg=0.5, block=1, seed=0):
    """Random masks: every rounding decision of the projection changes the result."""
    rng = np.random.default_rng(seed)
    hb, wb = (H + block - 1) // block, (W + block - 1) // block
    coarse = rng.random((V, hb, wb)) >= p_bg
    m = np.repeat(np.repeat(coarse, block, axis=1), block, axis=2)[:, :H, :W]
    out = (m * rng.integers(1, 256, size=(V, H, W))).astype(np.uint8)
    if C == 1:
        return out
    full = np.zeros((V, H, W, C), np.uint8)
    ch = rng.integers(0, C, size=(V, H, W))
    for c in range(C):  # foreground pixels are non-zero in at least one channel
        full[..., c] = np.where(ch == c, out, 0)
    return full


def random_cameras(V, extent, seed=0, W=160, H=120, inside=False):
    """Cameras at random positions looking roughly at the grid; some close or
    inside so that voxels fall behind the camera and near the image borders."""
    rng = np.random.default_rng(seed)
    K = syn.K_DATASET.copy()
    K[0] *= W / syn.IMAGE_W
    K[1] *= H / syn.IMAGE_H
    K32 = K.astype(np.float32)
    centre = np.array([extent / 2, extent / 2, -extent / 2])
    Rt = np.empty((V, 3, 4), np.float64)
    for i in range(V):
        d = rng.uniform(0.1 if inside else 0.8, 2.5) * extent
        dirv = rng.normal(size=3)
        dirv /= np.linalg.norm(dirv)
        cam = centre + d * dirv
        target = centre + rng.normal(scale=0.25 * extent, size=3)
        up = rng.normal(size=3)
        Rt[i] = syn.look_at_rt(cam, target, up)
    Rt32 = Rt.astype(np.float32)
    return K32, Rt32, syn.compose_m(K32, Rt32)
