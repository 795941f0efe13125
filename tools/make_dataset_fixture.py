#!/usr/bin/env python3
"""Packs the reference's own silhouette masks (Data/*/masks/*.jpg) into a small
fixture: tests/golden/dataset_masks.npz.

Only the == 0 / != 0 pattern of the PIL-decoded JPEGs is kept (1 bit per pixel;
that is all the carve reads, reference src/VoxelCarving.cpp:49-50).  These are
PLUMBING inputs: PIL's JPEG decoder is not OpenCV's, no undistortion is
applied and the camera poses used with them are synthetic ring cameras, so nothing
computed from this fixture is a parity claim about the reference's own run on the
data set -- it gives the tests real, ragged silhouettes (JPEG ringing leaves
thousands of isolated non-zero pixels) instead of analytic ones.

    python tools/make_dataset_fixture.py      (needs /root/reference; run in the build container)
"""
import glob
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/Data"


def load(ds, limit=None):
    files = sorted(glob.glob(os.path.join(REF, ds, "masks", "*.jpg")))
    if limit:
        files = files[::max(1, len(files) // limit)][:limit]
    out = []
    for f in files:
        m = np.array(Image.open(f).convert("RGB"))
        out.append((m != 0).any(axis=-1))
    return np.stack(out), [os.path.basename(f) for f in files]


def load_images_small(ds):
    """The data set's colour images at half resolution, re-encoded (JPEG q85): inputs for the
    colour vote's plumbing tests (tests/golden_io.py doubles the pixels back to 640x480).
    Not the reference's pixels -- PIL decode, resampled, re-encoded -- and not meant to be."""
    import io
    files = sorted(glob.glob(os.path.join(REF, ds, "images", "*.jpg")))
    blobs = []
    for f in files:
        im = Image.open(f).convert("RGB").resize((320, 240), Image.BILINEAR)
        buf = io.BytesIO()
        im.save(buf, "JPEG", quality=85)
        blobs.append(np.frombuffer(buf.getvalue(), np.uint8))
    sizes = np.array([len(b) for b in blobs], np.int64)
    return np.concatenate(blobs), sizes, [os.path.basename(f) for f in files]


def main():
    box, box_names = load("box_dataset")
    human, human_names = load("human_dataset")  # all 24 views (BASELINE configs 1 and 4)
    img, img_sizes, img_names = load_images_small("human_dataset")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "dataset_masks.npz"),
                        box_bits=np.packbits(box), box_shape=np.array(box.shape),
                        human_bits=np.packbits(human), human_shape=np.array(human.shape),
                        box_files=np.array(box_names), human_files=np.array(human_names),
                        human_images_jpeg=img, human_images_sizes=img_sizes,
                        human_image_files=np.array(img_names))
    print("box", box.shape, box.mean(), "human", human.shape, human.mean(), "images",
          len(img_sizes), int(img_sizes.sum()), "bytes")


if __name__ == "__main__":
    sys.exit(main())
