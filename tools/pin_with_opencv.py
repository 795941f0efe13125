#!/usr/bin/env python3
"""Ask a real OpenCV how it computes the three pieces of the carve's arithmetic that live inside
it (run on any host that has cv2 + numpy; no GPU needed):

  1. cv::gemm on a known-answer voxel: which grouping of the four products of a row of
     M * world it uses (reference src/VoxelCarving.cpp:19) -> ARVX_ASSOC_LEFT / _RIGHT
     (include/arvx/arvx.h; the library's default is LEFT, arvx_set_projection_assoc changes it);
  2. cv::norm of a Vec4f (src/ColorReconstruction.h:59) against sqrt of the fp64 sum of squares
     ((d0^2 + d1^2) + d2^2) + d3^2;
  3. cv::undistort of a distorted ramp against the restatement the device kernel follows
     (tests/np_restate.py::undistort == csrc/undistort_kernels.h).

Prints the three answers; exit code 0 when OpenCV agrees with the library's defaults, 1 when
something differs (say which), 2 when cv2 is missing.  The same checks run inside a C++ build
against OpenCV: include/arvx/opencv_dropin.hpp (dropin::self_pin).

    python tools/pin_with_opencv.py --emit tests/golden/opencv_pin.json

writes what this OpenCV computes on the probes of tests/pin_probes.py (its version, the grouping,
64 projected rows, 64 cv::norm depths, the undistorted ramp).  COMMITTING THAT FILE is what turns
the repository's parity from "unpinned" to pinned: tests/test_oracle_cpu.py::test_opencv_pin_file
then holds the C oracle and the numpy twin to it bit for bit, and fails when the recorded grouping
is not the library's default."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    try:
        import cv2
    except ImportError:
        print("cv2 is not installed here: run this on a host that has OpenCV (pip install "
              "opencv-python-headless).")
        return 2
    print("OpenCV", cv2.__version__)
    if len(sys.argv) >= 3 and sys.argv[1] == "--emit":
        import json
        from tests import pin_probes
        doc = pin_probes.emit(cv2)
        with open(sys.argv[2], "w") as f:
            json.dump(doc, f)
        print(f"wrote {sys.argv[2]}: grouping {doc['grouping']}; commit it (tests/golden/) and run "
              "python -m pytest tests/test_oracle_cpu.py -k opencv_pin")
        return 0 if doc["grouping"] in ("LEFT", "RIGHT") else 1
    rc = 0
    # 1. the known-answer voxel of tests/scenes.py::assoc_kat
    M = np.zeros((3, 4), np.float32)
    M[0] = [1.0, 2.0 ** -24, -(2.0 ** -54), 2.0 ** -53]
    M[1, 3] = M[2, 3] = np.float32(1.0 / 3.5)
    w = np.array([[1.0], [1.0], [-1.0], [1.0]], np.float32)  # toWord(1, 1, 1), s = 1
    proj = cv2.gemm(M, w, 1.0, None, 0.0)
    a0 = np.float32(proj[0, 0])
    if a0 == np.float32(1.0):
        print("1. cv::gemm: ((p0 + p1) + p2) + p3  -> ARVX_ASSOC_LEFT (the library's default)")
    elif a0 == np.float32(1.0) + np.float32(2.0 ** -23):
        print("1. cv::gemm: p0 + ((p1 + p2) + p3)  -> ARVX_ASSOC_RIGHT: call "
              "arvx_set_projection_assoc(ARVX_ASSOC_RIGHT) (opencv_dropin.hpp does it itself)")
        rc = 1
    else:
        print(f"1. cv::gemm gave {a0!r}: NEITHER grouping the library knows")
        rc = 1
    # and numpy's matmul operator on cv::Mat semantics: M @ w goes through BLAS, not cv::gemm
    # 2. cv::norm
    rng = np.random.default_rng(0)
    bad = 0
    for _ in range(1000):
        d = rng.normal(size=4).astype(np.float32)
        d[3] = 0
        d64 = d.astype(np.float64)
        want = np.float32(np.sqrt(((d64[0] * d64[0] + d64[1] * d64[1]) + d64[2] * d64[2]) + d64[3] * d64[3]))
        got = np.float32(cv2.norm(d.reshape(4, 1)))
        bad += int(got != want)
    print(f"2. cv::norm: {bad} of 1000 differ from sqrt(fp64 ((d0^2+d1^2)+d2^2)+d3^2)"
          + ("" if bad == 0 else "  <- the colour pass's depths may differ in the last bit"))
    rc = rc or int(bad != 0)
    # 3. cv::undistort
    from tests import np_restate as npr
    W, H = 96, 64
    yy, xx = np.mgrid[0:H, 0:W]
    src = ((xx * 5 + yy * 3 + (xx * yy) // 7) & 255).astype(np.uint8)
    K = np.array([[74.5, 0, 46.8], [0, 74.6, 33.4], [0, 0, 1]], np.float64)
    dist = np.array([0.12, -0.27, 0.0015, -0.0021, 0.11], np.float64)
    got = cv2.undistort(src, K, dist)
    want = npr.undistort(src, K, dist)
    diff = int((got != want).sum())
    print(f"3. cv::undistort: {diff} of {W * H} pixels differ from the restatement the device "
          "kernel follows" + ("" if diff == 0 else "  <- keep undistorting with OpenCV"))
    rc = rc or int(diff != 0)
    return rc


if __name__ == "__main__":
    sys.exit(main())
