"""Loader of the tests/golden/*.npz fixtures (written by tools/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0]
                              for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
                  if n != "dataset_masks")


def dataset_masks(which):
    """The reference data set's own silhouettes ('box': 8 views, 'human': 8 of 24), as
    (V,480,640) uint8 0/255 (tools/make_dataset_fixture.py; plumbing inputs only)."""
    z = np.load(os.path.join(GOLDEN_DIR, "dataset_masks.npz"))
    shape = tuple(int(v) for v in z[which + "_shape"])
    bits = np.unpackbits(z[which + "_bits"])[:int(np.prod(shape))]
    return (bits.reshape(shape) * 255).astype(np.uint8)


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    shape = tuple(int(v) for v in g["mask_shape"])
    bits = np.unpackbits(g["mask_nonzero_bits"])[:int(np.prod(shape))]
    g["masks"] = (bits.reshape(shape) * 255).astype(np.uint8)
    g["X"], g["Y"], g["Z"] = (int(v) for v in g["dims"])
    g["s"] = np.float32(g["voxel_size"])
    return g
