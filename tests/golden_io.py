"""Loader of the tests/golden/*.npz fixtures (written by tools/make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names():
    return sorted(os.path.splitext(os.path.basename(p))[0]
                  for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    g = {k: z[k] for k in z.files}
    shape = tuple(int(v) for v in g["mask_shape"])
    bits = np.unpackbits(g["mask_nonzero_bits"])[:int(np.prod(shape))]
    g["masks"] = (bits.reshape(shape) * 255).astype(np.uint8)
    g["X"], g["Y"], g["Z"] = (int(v) for v in g["dims"])
    g["s"] = np.float32(g["voxel_size"])
    return g
