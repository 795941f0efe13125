"""The C++ host layer (include/arvx/model.hpp, voxel_carving.hpp): the reference's
Model / carve / fastCarve / reconstruct*Color API over the C-ABI."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session", params=["test_host", "test_host_eigen"])
def host_bin(request):
    """tests/cpp/test_host.cpp built twice: with the C++ layer's stand-in vector types, and with
    arvx::Vec4f / Vec3f / Vec3i = the (mock) Eigen / cv types, the way the layer is compiled
    where those libraries are installed (include/arvx/vec_types.hpp)."""
    exe = os.path.join(ROOT, "tests", "cpp", request.param)
    if not os.path.exists(exe):
        from ar_voxel_project_amd import build
        build.build_host_tests()
    return exe


def test_model_host_logic(host_bin):
    """Model accessors, toWord, isInner, colour lists, handleUnseen: no GPU needed."""
    r = subprocess.run([host_bin, "model"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "model ok" in r.stdout


def test_calibration_file_reader(host_bin):
    """loadCalibrationFile's file format (OpenCV FileStorage YAML), parsed without OpenCV."""
    yml = os.path.join(ROOT, "tests", "golden", "cameracalibration.yml")
    r = subprocess.run([host_bin, "calibration", yml], capture_output=True, text=True)
    assert r.returncode == 0 and "calibration ok" in r.stdout, r.stderr


@pytest.mark.parametrize("n,mix", [(1, (0.0, 0.0)), (63, (0.5, 0.3)), (64, (0.2, 0.2)), (65, (0.6, 0.3)),
                                   (4096 + 7, (0.7, 0.25)), (1 << 14, (0.0, 1.0)), (5000, (1.0, 0.0))])
def test_packet_decoding_on_the_host(host_bin, tmp_path, n, mix):
    """Model::packet_bit / expand_packet (what the host accessors read once a carved model has come
    over as packets, include/arvx/model.hpp) against the numpy restatement of the packet layout
    (tests/occ_codec.py): every bit, the expansion, ragged last groups, all-zero / all-one / all-mixed
    planes.  No GPU."""
    from tests import occ_codec
    rng = np.random.default_rng(n)
    kind = rng.random(n)
    w = rng.integers(1, 2 ** 63, n, dtype=np.uint64) | (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63))
    w[w == occ_codec.ONES] = 7
    w[kind < mix[0]] = 0
    w[(kind >= mix[0]) & (kind < mix[0] + mix[1])] = occ_codec.ONES
    need = int(((w != 0) & (w != occ_codec.ONES)).sum())
    pk = occ_codec.compress(w, need + 3)
    path = tmp_path / "packet.bin"
    with open(path, "wb") as f:
        f.write(np.array([n, occ_codec.header_words(n), len(pk)], np.int64).tobytes())
        f.write(pk.tobytes())
        f.write(w.tobytes())
    r = subprocess.run([host_bin, "packet_decode", str(path)], capture_output=True, text=True)
    assert r.returncode == 0 and "packet decode ok" in r.stdout, r.stderr


@pytest.mark.gpu
def test_opencv_dropin_self_pinning_runs():
    """dropin::run_self_pin (include/arvx/opencv_dropin.hpp) against the stand-ins of
    tests/cpp/mock_opencv: a cv::gemm for either row-sum grouping switches the library to that
    grouping; an unknown gemm, a different cv::norm or cv::undistort are reported
    (tests/cpp/test_selfpin.cpp).  Checks the self-pinning's logic, not OpenCV's arithmetic."""
    exe = os.path.join(ROOT, "tests", "cpp", "test_selfpin")
    if not os.path.exists(exe):
        pytest.fail("tests/cpp/test_selfpin missing: run __graft_entry__.build()")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "selfpin ok" in r.stdout, r.stdout + r.stderr


def write_scene(path, X, Y, Z, s, K, Rt, masks, images, st0):
    m4 = masks if masks.ndim == 4 else masks[..., None]
    V, H, W, C = m4.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<7i", X, Y, Z, V, W, H, C))
        f.write(np.float32(s).tobytes())
        f.write(np.ascontiguousarray(K, np.float32).tobytes())
        f.write(np.ascontiguousarray(Rt, np.float32).tobytes())
        f.write(np.ascontiguousarray(m4).tobytes())
        f.write(np.ascontiguousarray(images).tobytes())
        f.write(np.ascontiguousarray(st0, np.uint8).tobytes())


def read_result(path, n):
    raw = np.fromfile(path, np.uint8)
    rgba = raw[:16 * n].view(np.float32).reshape(n, 4)
    seen = raw[16 * n:16 * n + n].astype(bool)
    return rgba, seen


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["carve", "carve_steps", "closest", "average_unseen", "fast",
                                  "closure", "closure_mc", "recarve_colored", "threads"])
def test_cpp_entry_points_match_oracle(host_bin, oracle, tmp_path, mode):
    X, Y, Z, V = 30, 20, 12, 5
    sc = scenes.syn.sphere_scene(32, V, W=96, H=72, with_images=True)
    s = np.float32(0.512 / 30)
    rng = np.random.default_rng(2)
    st0 = np.where(rng.random((Z, Y, X)) < 0.05, 0, 1).astype(np.uint8)  # some pre-carved voxels
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    write_scene(scene, X, Y, Z, s, sc.K, sc.Rt, sc.masks, sc.images, st0)
    r = subprocess.run([host_bin, "carve", scene, out, mode], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    rgba, seen = read_result(out, X * Y * Z)
    M = oracle.compose(sc.K, sc.Rt)
    if mode == "fast":
        assert "LOG - VC: starting carving process (version 2)." in r.stdout
        st = oracle.fast_carve(X, Y, Z, s, M, sc.masks, state=st0)
    else:
        assert "LOG - VC: starting carving process (version 1)." in r.stdout
        st = oracle.carve(X, Y, Z, s, M, sc.masks, state=st0)
    want = oracle.model_from_state(st)
    if mode in ("closest", "recarve_colored"):
        want = oracle.color(X, Y, Z, s, M, sc.campos, sc.images, 0, want)
        assert "LOG - CR: starting color reconstruction (closest color)." in r.stdout
    if mode == "recarve_colored":
        # handleUnseen paints never-seen voxels; the second carve changes nothing the
        # first did not (it is idempotent) -- but a never-seen voxel stays unseen too
        want = oracle.handle_unseen(st, want)
    if mode in ("average_unseen", "closure", "closure_mc", "threads"):
        want = oracle.color(X, Y, Z, s, M, sc.campos, sc.images, 1, want)
        want = oracle.handle_unseen(st, want)
    if mode in ("closure", "closure_mc", "threads"):  # (threads: three host threads, three models
        # each, all equal to the main thread's -- checked by the binary -- and to the oracle)
        want = oracle.closure(X, Y, Z, want)
        assert "LOG - PP: starting dilution." in r.stdout
    assert np.array_equal(seen, (st.reshape(-1) & 2) == 2)
    assert np.array_equal(rgba, want)
    if mode == "closure_mc":  # the cells the reference's marchingCubes would triangulate
        raw = np.fromfile(out, np.uint8)[17 * X * Y * Z:]
        n = int(raw[:4].view(np.int32)[0])
        cells = raw[4:4 + 16 * n].view(np.int32).reshape(n, 4)
        assert n > 0 and np.array_equal(cells, oracle.mc_cells(X, Y, Z, want))


@pytest.mark.gpu
@pytest.mark.parametrize("X,Y,Z", [(64, 24, 12), (32, 2, 2), (96, 64, 40)])
def test_cpp_model_answers_from_packets(host_bin, oracle, tmp_path, X, Y, Z):
    """A grid with X % 32 == 0 and X * Y % 64 == 0: the carved model crosses PCIe as compressed
    packets (arvx_state_download_packets) and Model::get / isInner / visited answer from them;
    the binary checks the accessor sweep against the plane form and the C-ABI's planes, this test
    the final model against the oracle (reference src/Model.h:119-160, src/Model.cpp:36-47)."""
    V = 5
    sc = scenes.syn.sphere_scene(32, V, W=96, H=72, with_images=True)
    s = np.float32(0.512 / max(X, Y, Z))
    rng = np.random.default_rng(3)
    st0 = np.where(rng.random((Z, Y, X)) < 0.05, 0, 1).astype(np.uint8)
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    write_scene(scene, X, Y, Z, s, sc.K, sc.Rt, sc.masks, sc.images, st0)
    r = subprocess.run([host_bin, "carve", scene, out, "packets"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    rgba, seen = read_result(out, X * Y * Z)
    M = oracle.compose(sc.K, sc.Rt)
    st0[1, 1, 1] &= 0xfe  # the binary's set(1, 1, 1, zero) between its two carves
    st = oracle.carve(X, Y, Z, s, M, sc.masks, state=st0)
    want = oracle.handle_unseen(st, oracle.model_from_state(st))
    assert np.array_equal(seen, (st.reshape(-1) & 2) == 2)
    assert np.array_equal(rgba, want)


@pytest.mark.gpu
def test_cpp_debug_mesh(host_bin, oracle, tmp_path):
    """Model::WriteModel (reference src/Model.cpp:49-107): a cube per voxel that is there and not
    inner, x outermost, in the voxel's colour -- the text restated here from the oracle's model."""
    X, Y, Z, V = 14, 11, 9, 4
    sc = scenes.syn.sphere_scene(16, V, W=96, H=72, with_images=True)
    s = np.float32(0.512 / 14)
    st0 = np.ones((Z, Y, X), np.uint8)
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    write_scene(scene, X, Y, Z, s, sc.K, sc.Rt, sc.masks, sc.images, st0)
    r = subprocess.run([host_bin, "carve", scene, out, "debug_mesh"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "LOG - Debug: generating debug mesh from model..." in r.stdout
    assert "LOG - Debug: debug mesh written." in r.stdout
    assert "LOG(ERR) - Debug: could not open file /nonexistent-dir/x.off" in r.stderr
    M = oracle.compose(sc.K, sc.Rt)
    st = oracle.carve(X, Y, Z, s, M, sc.masks)
    want = oracle.color(X, Y, Z, s, M, sc.campos, sc.images, 0, oracle.model_from_state(st))
    want = oracle.handle_unseen(st, want).reshape(Z, Y, X, 4)
    occ = np.pad(want[..., 3] != 0, 1, constant_values=False)
    verts, faces = [], []
    corner = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
    quads = [(0, 2, 4, 1), (3, 5, 7, 6), (0, 2, 6, 3), (1, 4, 7, 5), (2, 6, 7, 4), (0, 3, 5, 1)]
    for x in range(X):
        for y in range(Y):
            for z in range(Z):
                if not occ[z + 1, y + 1, x + 1]:
                    continue
                if (occ[z + 1, y + 1, x] and occ[z + 1, y + 1, x + 2] and occ[z + 1, y, x + 1] and
                        occ[z + 1, y + 2, x + 1] and occ[z, y + 1, x + 1] and occ[z + 2, y + 1, x + 1]):
                    continue
                b = len(verts)
                verts += [f"{x + dx} {y + dy} {z + dz}" for dx, dy, dz in corner]
                c = want[z, y, x]
                faces += [f"4 {b + q[0]} {b + q[1]} {b + q[2]} {b + q[3]} {int(c[0])} {int(c[1])} {int(c[2])}"
                          for q in quads]
    assert len(verts) > 100
    text = "\n".join(["OFF", f"{len(verts)} {len(faces)} 0"] + verts + faces) + "\n"
    assert open(out + ".off").read() == text


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["carve", "fast", "closest"])
def test_cpp_entry_points_on_a_new_model(host_bin, oracle, tmp_path, mode):
    """A Model nobody touched goes to the GPU as arvx_state_reset, not as an upload."""
    X, Y, Z, V = 24, 20, 16, 4
    sc = scenes.syn.sphere_scene(32, V, W=96, H=72, with_images=True)
    s = np.float32(0.512 / 24)
    st0 = np.ones((Z, Y, X), np.uint8)  # occupied, unseen: the C++ side never calls set()/see()
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    write_scene(scene, X, Y, Z, s, sc.K, sc.Rt, sc.masks, sc.images, st0)
    r = subprocess.run([host_bin, "carve", scene, out, mode], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    rgba, seen = read_result(out, X * Y * Z)
    M = oracle.compose(sc.K, sc.Rt)
    st = (oracle.fast_carve if mode == "fast" else oracle.carve)(X, Y, Z, s, M, sc.masks)
    want = oracle.model_from_state(st)
    if mode == "closest":
        want = oracle.color(X, Y, Z, s, M, sc.campos, sc.images, 0, want)
    assert np.array_equal(seen, (st.reshape(-1) & 2) == 2)
    assert np.array_equal(rgba, want)


@pytest.mark.gpu
def test_cpp_carve_over_a_device_list(host_bin, oracle, tmp_path):
    """arvx::carve(intr, model, views, devices) (include/arvx/multi_gpu.hpp): one process,
    striped contexts, RCCL merge -- with the one device of this box; Z must be a multiple
    of 8 and X*Y of 64."""
    X, Y, Z, V = 32, 16, 24, 5
    sc = scenes.syn.sphere_scene(32, V, W=96, H=72, with_images=True)
    s = np.float32(0.512 / 32)
    rng = np.random.default_rng(4)
    st0 = np.where(rng.random((Z, Y, X)) < 0.05, 0, 1).astype(np.uint8)
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    write_scene(scene, X, Y, Z, s, sc.K, sc.Rt, sc.masks, sc.images, st0)
    r = subprocess.run([host_bin, "carve", scene, out, "carve_devices"], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    rgba, seen = read_result(out, X * Y * Z)
    st = oracle.carve(X, Y, Z, s, oracle.compose(sc.K, sc.Rt), sc.masks, state=st0)
    assert np.array_equal(seen, (st.reshape(-1) & 2) == 2)
    assert np.array_equal(rgba, oracle.model_from_state(st))


@pytest.mark.gpu
def test_bench6_table(host_bin, oracle, tmp_path):
    """The reference's -c=6 benchmark sequence (src/main.cpp:306-440) through the C++
    layer: eight runs, reference table layout, occupancy equal to the oracle's."""
    exe = os.path.join(ROOT, "tools", "cpp", "arvx_bench6")
    V, W, H = 8, 640, 480
    sc = scenes.syn.box_scene((100, 100, 50), V, with_images=True)
    # the reference's box sits in a 0.28 m cube (100 x 0.0028): scale the scene to it
    scale = 0.28 / 0.512
    Rt = sc.Rt.copy()
    Rt[:, :, 3] *= np.float32(scale)
    scene = str(tmp_path / "scene.bin")
    write_scene(scene, 1, 1, 1, 1.0, sc.K, Rt, sc.masks, sc.images, np.ones(1, np.uint8))
    r = subprocess.run([exe, scene], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr + r.stdout[-2000:]
    assert "Benchmark (all times in milliseconds)" in r.stdout
    assert "n/a" not in r.stdout  # the marching-cubes column is measured
    rows = [ln for ln in r.stdout.splitlines() if "coloring\t|" in ln]
    assert len(rows) == 8 and rows[3].startswith("Large, V1, avg. coloring") and "100x100x50" in rows[3]
    occ = [int(ln.split()[1]) for ln in r.stdout.splitlines() if ln.startswith("occupied ")]
    M = oracle.compose(sc.K, Rt)
    for (dims, s, ver), got in zip([((10, 10, 5), 0.028, 1), ((50, 50, 25), 0.0056, 1),
                                    ((50, 50, 25), 0.0056, 1), ((100, 100, 50), 0.0028, 1),
                                    ((10, 10, 5), 0.028, 2), ((50, 50, 25), 0.0056, 2),
                                    ((50, 50, 25), 0.0056, 2), ((100, 100, 50), 0.0028, 2)], occ):
        X, Y, Z = dims
        st = (oracle.carve if ver == 1 else oracle.fast_carve)(X, Y, Z, np.float32(s), M, sc.masks)
        model = oracle.color(X, Y, Z, np.float32(s), M, Rt[:, :, 3], sc.images, 1,
                             oracle.model_from_state(st))
        closed = oracle.closure(X, Y, Z, oracle.handle_unseen(st, model))
        assert got == int((closed[:, 3] != 0).sum()), (dims, ver)
        if dims == (100, 100, 50):  # and the mesh file the run wrote (src/main.cpp:391,435)
            verts, rgb = oracle.mc_mesh(X, Y, Z, closed)
            want = oracle.off_text(verts, rgb, np.float32(1.0) * np.float32(s))
            name = f"out/bench/mesh_large_{ver}_avg.off"
            assert open(os.path.join(str(tmp_path), name), "rb").read() == want.encode()
    print(r.stdout[r.stdout.index("Benchmark (all"):])
