"""Loader of the tests/golden/*.npz fixtures (written by tools/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0]
                              for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
                  if n not in ("dataset_masks", "box_off1", "box_off23"))


def dataset_masks(which, recentre=False):
    """The reference data set's own silhouettes ('box': 8 views, 'human': all 24), as
    (V,480,640) uint8 0/255 (tools/make_dataset_fixture.py; plumbing inputs only).
    recentre: every silhouette is shifted so that its median pixel sits on the principal
    point.  The data set's real poses need OpenCV-aruco; the tests pair the masks with ring
    cameras that look at the grid centre, and unshifted the 24 human silhouettes have an
    empty visual hull there."""
    z = np.load(os.path.join(GOLDEN_DIR, "dataset_masks.npz"))
    shape = tuple(int(v) for v in z[which + "_shape"])
    bits = np.unpackbits(z[which + "_bits"])[:int(np.prod(shape))]
    masks = (bits.reshape(shape) * 255).astype(np.uint8)
    if not recentre:
        return masks
    out = np.zeros_like(masks)
    H, W = masks.shape[1:]
    for i, mk in enumerate(masks):
        ys, xs = np.nonzero(mk)
        dy, dx = int(round(251 - np.median(ys))), int(round(312 - np.median(xs)))
        src = mk[max(0, -dy):H - max(0, dy), max(0, -dx):W - max(0, dx)]
        out[i, max(0, dy):max(0, dy) + src.shape[0], max(0, dx):max(0, dx) + src.shape[1]] = src
    return out


def dataset_images(which="human"):
    """The data set's colour images as (V,480,640,3) uint8 BGR: stored at half resolution
    (tools/make_dataset_fixture.py), pixels doubled here.  Plumbing inputs only."""
    import io
    from PIL import Image
    z = np.load(os.path.join(GOLDEN_DIR, "dataset_masks.npz"))
    blob, sizes = z[which + "_images_jpeg"], z[which + "_images_sizes"]
    out, pos = [], 0
    for n in sizes:
        rgb = np.array(Image.open(io.BytesIO(blob[pos:pos + int(n)].tobytes())).convert("RGB"))
        pos += int(n)
        big = np.repeat(np.repeat(rgb, 2, axis=0), 2, axis=1)
        out.append(np.ascontiguousarray(big[:, :, ::-1]))
    return np.stack(out)


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    shape = tuple(int(v) for v in g["mask_shape"])
    bits = np.unpackbits(g["mask_nonzero_bits"])[:int(np.prod(shape))]
    g["masks"] = (bits.reshape(shape) * 255).astype(np.uint8)
    g["X"], g["Y"], g["Z"] = (int(v) for v in g["dims"])
    g["s"] = np.float32(g["voxel_size"])
    return g
