"""arvx_undistort: cv::undistort on the device (csrc/undistort_kernels.h).  PARITY UNPINNED --
OpenCV is not in this image; the checks are (i) the identity: without distortion every output
pixel is its source pixel, bit for bit, (ii) agreement with a second restatement of the same
published algorithm in numpy (tests/np_restate.py) on the reference data set's calibration and
images, (iii) properties: constant images stay constant where all four taps exist, the
constant-0 border shows where they do not, the map is symmetric for a centred camera."""
import os

import numpy as np
import pytest

from tests import golden_io, np_restate as npr, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Data/*/cameracalibration.yml of the reference (tests/golden/cameracalibration.yml)
K_DATA = scenes.syn.K_DATASET

def dataset_dist():
    txt = open(os.path.join(ROOT, "tests", "golden", "cameracalibration.yml")).read()
    body = txt[txt.index("distortion_coefficients"):]
    body = body[body.index("[") + 1:body.index("]")]
    return np.array([float(t) for t in body.replace("\n", " ").split(",")])


@pytest.mark.parametrize("C", [1, 3])
def test_identity_without_distortion(arvx, C):
    rng = np.random.default_rng(C)
    img = rng.integers(0, 256, (2, 48, 64, C), dtype=np.uint8)
    with arvx.Context(4, 4, 4, 0.1) as ctx:
        out = ctx.undistort(img, K_DATA, np.zeros(5))
        assert np.array_equal(out, img)
        out = ctx.undistort(img, K_DATA, np.zeros(0))
        assert np.array_equal(out, img)


def test_matches_numpy_restatement_on_dataset_calibration(arvx):
    dist = dataset_dist()
    assert len(dist) == 5
    masks = golden_io.dataset_masks("box")[:3]
    images = golden_io.dataset_images("human")[:2]
    with arvx.Context(4, 4, 4, 0.1) as ctx:
        got = ctx.undistort(masks, K_DATA, dist)
        for i in range(len(masks)):
            assert np.array_equal(got[i], npr.undistort(masks[i], K_DATA, dist)), f"mask {i}"
        assert (got != masks).any()  # the data set's distortion moves pixels
        goti = ctx.undistort(images, K_DATA, dist)
        for i in range(len(images)):
            assert np.array_equal(goti[i], npr.undistort(images[i], K_DATA, dist)), f"image {i}"


@pytest.mark.parametrize("ndist", [4, 5, 8])
def test_random_coefficients_and_sizes(arvx, ndist):
    rng = np.random.default_rng(ndist)
    for (W, H, C) in [(65, 33, 1), (128, 96, 3), (31, 70, 4)]:
        K = np.array([[0.9 * W, 0, W / 2 + rng.normal()], [0, 0.95 * W, H / 2 + rng.normal()],
                      [0, 0, 1]])
        dist = rng.normal(scale=[0.2, 0.2, 0.01, 0.01, 0.1, 0.05, 0.05, 0.05][:ndist])
        img = rng.integers(0, 256, (2, H, W, C), dtype=np.uint8)
        with arvx.Context(4, 4, 4, 0.1) as ctx:
            got = ctx.undistort(img, K, dist)
        for i in range(2):
            assert np.array_equal(got[i], npr.undistort(img[i], K, dist)), (W, H, C, ndist)


def test_properties(arvx):
    W, H = 96, 64
    K = np.array([[80.0, 0, (W - 1) / 2], [0, 80.0, (H - 1) / 2], [0, 0, 1]])
    dist = np.array([0.3, 0.05, 0.0, 0.0, 0.0])  # radial only: barrel
    img = np.full((1, H, W), 200, np.uint8)
    with arvx.Context(4, 4, 4, 0.1) as ctx:
        out = ctx.undistort(img, K, dist)[0]
        # centre: all taps inside -> the constant; the corners read outside the source -> 0 border
        assert (out[H // 4:3 * H // 4, W // 4:3 * W // 4] == 200).all()
        assert out[0, 0] < 200 and out[-1, -1] < 200
        assert set(np.unique(out)) <= set(range(0, 201))
        # a centred camera and radial distortion: mirror symmetry of the result
        grad = (np.arange(W)[None, :] + np.zeros((H, 1))).astype(np.uint8)
        sym = np.minimum(grad, grad[:, ::-1])[None]
        o2 = ctx.undistort(sym, K, dist)[0]
        assert np.array_equal(o2, o2[:, ::-1]) and np.array_equal(o2, o2[::-1, :])
    with pytest.raises(arvx.ArvxError):
        with arvx.Context(4, 4, 4, 0.1) as ctx:
            ctx.undistort(img, K, np.zeros(3))  # 3 coefficients: not a cv::undistort form
