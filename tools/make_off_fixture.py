#!/usr/bin/env python3
"""Turns the reference's own known-answer file Data/box_dataset/generated_models/1.off into
the fixture tests/golden/box_off1.npz (build container only: it reads /root/reference).

1.off is the output of `-c=5 -z=50` on the box data set (generated_models/README.adoc:6-10):
carve -> handleUnseen -> applyClosure -> marchingCubes on a 100 x 100 x 50 model with voxel
size 0.0028, colour mode 0.  The file alone determines the model marching cubes ran on:

  * every vertex is a lattice point -- the occupied corner a cut edge snaps to
    (src/MarchingCubes.h:428-441) -- so the distinct vertices are exactly the occupied voxels
    with an empty (or out-of-grid) 6-neighbour, i.e. the voxels that are not Model::isInner
    (src/Model.h:126-132);
  * the other voxels are empty or inner.  An empty voxel never touches an inner one, so each
    6-connected component of the rest is all empty or all inner; those reaching the grid border
    are empty (outside the grid counts as empty, src/Model.h:119-122); one component does not
    reach it -- the inside of the box;
  * running marching cubes (oracle/arvx_oracle.c, arvx_oracle_mc_mesh) over surface + inside
    must reproduce the file byte for byte: checked here, or the script fails.

The fixture holds: dims, voxel size, the occupancy (bit-packed), the distinct surface voxels,
and the file's text (zlib) with its SHA-256 -- data the reference holds, no source.

    python tools/make_off_fixture.py
"""
import hashlib
import os
import sys
import zlib

import numpy as np
from scipy import ndimage

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/Data/box_dataset/generated_models"
X, Y, Z, SIZE = 100, 100, 50, np.float32(0.0028)  # src/main.cpp:26-29 defaults, -z=50


def parse_off(path):
    with open(path, "rb") as f:
        raw = f.read()
    lines = raw.decode().split("\n")
    assert lines[0] == "OFF"
    nv, nf, _ = map(int, lines[1].split())
    v = np.array([l.split() for l in lines[2:2 + nv]], np.float64)
    f = np.array([l.split() for l in lines[2 + nv:2 + nv + nf]], np.int64)
    return raw, v, f


def main():
    from ar_voxel_project_amd import build
    build.build_oracle()
    from oracle import pyoracle as po
    raw, v, f = parse_off(os.path.join(REF, "1.off"))
    lat = np.rint(v / float(SIZE)).astype(np.int64)
    assert np.abs(v / float(SIZE) - lat).max() < 1e-3, "vertices are not on the lattice"
    assert (f[:, 0] == 3).all() and (f[:, 1:4].reshape(-1) == np.arange(3 * len(f))).all()
    surf = np.zeros((Z, Y, X), bool)
    surf[lat[:, 2], lat[:, 1], lat[:, 0]] = True
    # components of the rest, 6-connected, on a grid padded by one empty layer
    rest = np.pad(~surf, 1, constant_values=True)
    lab, n = ndimage.label(rest)  # default structure: 6-connectivity
    outside = lab[0, 0, 0]
    sizes = np.bincount(lab.ravel())
    enclosed = [k for k in range(1, n + 1) if k != outside]
    print("distinct surface voxels", int(surf.sum()), "| enclosed components:",
          [int(sizes[k]) for k in enclosed])
    inner = np.isin(lab, enclosed)[1:-1, 1:-1, 1:-1]
    occ = surf | inner
    state = (occ * 1 | 2).astype(np.uint8)  # occupied bit; everything seen (colour mode 0)
    rgba = po.model_from_state(state)
    verts, rgb = po.mc_mesh(X, Y, Z, rgba, 0.5)
    text = po.off_text(verts, rgb, scale_factor=np.float32(1.0) * SIZE).encode()
    assert len(verts) == len(v), (len(verts), len(v))
    assert np.array_equal(verts.astype(np.int64), lat), "vertex sequence differs from 1.off"
    assert text == raw, "OFF text differs from 1.off"
    # and the pieces the product's parity tests compare
    surface_index = np.flatnonzero(surf.ravel()).astype(np.int32)
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "box_off1.npz"),
        dims=np.array([X, Y, Z], np.int32), voxel_size=SIZE,
        occupancy_bits=np.packbits(occ.ravel()),
        surface_index=surface_index,  # flat x + X*(y + Y*z), ascending
        n_vertices=np.int64(len(v)), n_faces=np.int64(len(f)),
        off_sha256=np.frombuffer(hashlib.sha256(raw).digest(), np.uint8),
        off_zlib=np.frombuffer(zlib.compress(raw, 9), np.uint8))
    print("1.off reproduced byte for byte:", len(raw), "bytes,", len(v), "vertices,", len(f),
          "faces; occupied", int(occ.sum()))
    # 2.off / 3.off: same geometry (colour modes 1 and 2); their face colours need the
    # reference's poses and OpenCV's JPEG decode + undistort, which this image cannot produce
    for name in ("2.off", "3.off"):
        raw2, v2, f2 = parse_off(os.path.join(REF, name))
        assert np.array_equal(v2, v) and np.array_equal(f2[:, :4], f[:, :4]), name
        print(name, "same vertices and faces;", len(np.unique(f2[:, 4:], axis=0)), "face colours")


if __name__ == "__main__":
    main()
