#!/usr/bin/env python3
"""Stage times of the drop-in C++ layer (tools/cpp/arvx_dropin_time) on the sphere scene with
the reference's input format (3-channel masks and images, 640x480): python tools/dropin_times.py
[N ...]   (GPU required)"""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from ar_voxel_project_amd import synthetic  # noqa: E402
from tests.test_cpp_host import write_scene  # noqa: E402


def main():
    grids = [int(a) for a in sys.argv[1:]] or [100, 512]
    V = 36
    sc = synthetic.sphere_scene(64, V, with_images=True)
    masks3 = np.repeat(sc.masks[..., None], 3, axis=-1)
    with tempfile.TemporaryDirectory() as d:
        scene = os.path.join(d, "scene.bin")
        write_scene(scene, 1, 1, 1, 1.0, sc.K, sc.Rt, masks3, sc.images, np.ones(1, np.uint8))
        for N in grids:
            r = subprocess.run([os.path.join(ROOT, "tools", "cpp", "arvx_dropin_time"), scene,
                                str(N), str(N), str(N), repr(0.512 / N), "4"],
                               capture_output=True, text=True)
            print(r.stderr.strip() or r.stdout.strip())


if __name__ == "__main__":
    main()
