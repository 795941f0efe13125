"""tools/cpp/arvx_cli: the reference's `-c=5` command line (src/main.cpp:184-305) -- flag names,
stage order, log lines, output file -- over pre-undistorted PPM/PGM inputs and given poses,
against the oracle's carve -> colour -> handleUnseen -> closure -> marching cubes -> OFF text;
`-intermediateMesh` writes the reference's per-view meshes (src/VoxelCarving.cpp:65-68)."""
import os
import subprocess

import numpy as np
import pytest

from tests import scenes
from tests.test_cpp_host import write_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tools", "cpp", "arvx_cli")
YML = os.path.join(ROOT, "tests", "golden", "cameracalibration.yml")


@pytest.fixture(scope="module")
def cli():
    if not os.path.exists(CLI):
        from ar_voxel_project_amd import build
        build.build_host_tests()
    return CLI


def write_inputs(d, sc):
    os.makedirs(os.path.join(d, "images"))
    os.makedirs(os.path.join(d, "masks"))
    for i in range(sc.V):
        H, W = sc.masks[i].shape
        with open(os.path.join(d, "images", f"image{i:04d}.ppm"), "wb") as f:
            f.write(b"P6\n# undistorted\n%d %d\n255\n" % (W, H))
            f.write(np.ascontiguousarray(sc.images[i][:, :, ::-1]).tobytes())  # BGR -> RGB
        with open(os.path.join(d, "masks", f"mask{i:04d}.pgm"), "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (W, H))
            f.write(np.ascontiguousarray(sc.masks[i]).tobytes())
    with open(os.path.join(d, "poses.txt"), "w") as f:
        for rt in sc.Rt:
            f.write(" ".join(repr(float(v)) for v in rt.reshape(-1)) + "\n")


def expected(oracle, sc, X, Y, Z, s, carve, color, post):
    M = oracle.compose(sc.K, sc.Rt)
    st = (oracle.carve if carve == 1 else oracle.fast_carve)(X, Y, Z, s, M, sc.masks)
    model = oracle.model_from_state(st)
    if color:
        model = oracle.color(X, Y, Z, s, M, sc.campos, sc.images, color - 1, model)
    model = oracle.handle_unseen(st, model)
    if post:
        model = oracle.closure(X, Y, Z, model)
    return model


@pytest.mark.parametrize("carve,color,post", [(1, 2, True), (2, 1, True), (1, 0, False)])
def test_c5_matches_oracle(cli, oracle, tmp_path, carve, color, post):
    X, Y, Z = 40, 36, 20
    s = np.float32(0.512 / 40)
    sc = scenes.syn.sphere_scene(64, 5, with_images=True)  # 640x480, the data set's intrinsics
    d = str(tmp_path)
    write_inputs(d, sc)
    out = os.path.join(d, "mesh.off")
    cmd = [cli, "-c=5", f"-images={d}/images", f"-masks={d}/masks", f"-poses={d}/poses.txt",
           f"-calibration={YML}", f"-x={X}", f"-y={Y}", f"-z={Z}", f"-size={float(s)!r}",
           f"-carve={carve}", f"-color={color}", f"-postprocessing={'true' if post else 'false'}",
           "-scale=2.0", "-dx=0.5", f"-outFile={out}"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    for line in ("LOG - VC: images read.", "LOG - VC: masks read.",
                 "LOG - VC: read cameraMatrix and distCoefficients.",
                 f"LOG - VC: starting carving process (version {carve}).",
                 "LOG - PP: marking unseen voxels from model.",
                 "LOG - MC: Mesh written, marchingCubes completed."):
        assert line in r.stdout
    assert ("LOG - PP: starting postprocessing." in r.stdout) == post
    model = expected(oracle, sc, X, Y, Z, s, carve, color, post)
    verts, rgb = oracle.mc_mesh(X, Y, Z, model)
    want = oracle.off_text(verts, rgb, np.float32(2.0) * s, (0.5, 0.0, 0.0))
    assert open(out, "rb").read() == want.encode()
    for sub in ("out", "out/tmp", "out/intermediate"):  # src/main.cpp:45-47
        assert os.path.isdir(os.path.join(d, sub))


def test_c5_scene_file_and_intermediate_meshes(cli, oracle, tmp_path):
    X, Y, Z, V = 24, 20, 16, 3
    s = np.float32(0.512 / 24)
    sc = scenes.syn.sphere_scene(32, V, W=96, H=72, with_images=True)
    d = str(tmp_path)
    scene = os.path.join(d, "scene.bin")
    write_scene(scene, 1, 1, 1, 1.0, sc.K, sc.Rt, sc.masks, sc.images, np.ones(1, np.uint8))
    r = subprocess.run([cli, "-c=5", f"-scene={scene}", "-calibration=none.yml", f"-x={X}",
                        f"-y={Y}", f"-z={Z}", f"-size={float(s)!r}", "-intermediateMesh=true",
                        "-postprocessing=false", f"-outFile={d}/m.off"],
                       capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout
    M = oracle.compose(sc.K, sc.Rt)
    st = oracle.fresh_state(X, Y, Z)
    for i in range(V):
        assert f"LOG - VC: generating intermediate mesh for image {i}" in r.stdout
        st = oracle.carve_view(X, Y, Z, s, M[i], sc.masks[i], st)
        verts, rgb = oracle.mc_mesh(X, Y, Z, oracle.model_from_state(st))
        # marchingCubes(&model, 1.0f, (i * (X + 2) * size, 0, 0), 0.5f, ...), VoxelCarving.cpp:67
        tx = np.float32(i * (X + 2)) * s
        want = oracle.off_text(verts, rgb, np.float32(1.0) * s, (tx, 0.0, 0.0))
        got = open(os.path.join(d, "out", "intermediate", f"image_{i}_mesh.off"), "rb").read()
        assert got == want.encode(), f"intermediate mesh {i}"


def test_c6_table(cli, tmp_path):
    sc = scenes.syn.box_scene((100, 100, 50), 4, with_images=True)
    d = str(tmp_path)
    scene = os.path.join(d, "scene.bin")
    write_scene(scene, 1, 1, 1, 1.0, sc.K, sc.Rt, sc.masks, sc.images, np.ones(1, np.uint8))
    r = subprocess.run([cli, "-c=6", f"-scene={scene}", f"-calibration={YML}"],
                       capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr + r.stdout[-1500:]
    assert "LOG - Benchmark: read cameraMatrix and distCoefficients." in r.stdout
    assert "Benchmark (all times in milliseconds)" in r.stdout
    assert len([ln for ln in r.stdout.splitlines() if "coloring\t|" in ln]) == 8
    assert os.path.exists(os.path.join(d, "out", "bench", "mesh_large_2_avg.off"))


def test_argument_errors(cli, tmp_path):
    d = str(tmp_path)
    cases = [(["-c=5", "-carve=3"], "Invalid carve argument."),
             (["-c=5"], "You need to define a images path (--images)"),
             (["-c=5", f"-images={d}"], "You need to define a image masks path (--masks)"),
             (["-c=2"], "not part of this library")]
    for args, msg in cases:
        r = subprocess.run([cli] + args, capture_output=True, text=True, cwd=d)
        assert r.returncode != 0 and msg in r.stderr
