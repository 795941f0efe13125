"""The reference's own caller against the drop-in headers ("keeping the
Model/VoxelCarving C++ API surface so it drops in behind main.cpp").

`/root/reference/src/main.cpp` is compiled (g++ -fsyntax-only) with `include/arvx/dropin/`
standing where the replaced headers stood (Model.h, VoxelCarving.h, ColorReconstruction.h,
MarchingCubes.h, Postprocessing3d.h, Benchmark.h), the headers the reference keeps
(Calibration.h, PoseEstimation.h, Segmentation.h, aruco_samples_utility.hpp) read where they
lie, and declarations-only stand-ins for OpenCV and Eigen (tests/cpp/mock_opencv,
tests/cpp/mock_eigen: this image has neither).  Nothing of the reference is copied into the
repo: main.cpp goes to a temporary directory (so that `#include "Model.h"` does not find its
neighbour), and the tests are skipped where /root/reference does not exist (the GPU box).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/src"
REPLACED = ("Model.h", "VoxelCarving.h", "ColorReconstruction.h", "MarchingCubes.h",
            "Postprocessing3d.h", "Benchmark.h")

needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF_SRC, "main.cpp")),
                                     reason="the reference checkout is not on this machine")


CLANG = "/opt/rocm/lib/llvm/bin/clang++"
COMPILERS = ["g++"] + ([CLANG] if os.path.exists(CLANG) else [])


def _syntax_only(source, *include_dirs, extra=(), cxx="g++"):
    cmd = [cxx, "-std=c++17", "-fsyntax-only", *extra]
    for d in include_dirs:
        cmd.append("-I" + d)
    cmd.append(source)
    return subprocess.run(cmd, capture_output=True, text=True)


MOCKS = (os.path.join(ROOT, "tests", "cpp", "mock_opencv"),
         os.path.join(ROOT, "tests", "cpp", "mock_eigen"))


@needs_reference
@pytest.mark.parametrize("cxx", COMPILERS)
def test_reference_main_compiles_unchanged(tmp_path, cxx):
    """Zero source edits: include/arvx/dropin on the include path, the replaced files gone
    (g++, and ROCm's clang++ where it is installed)."""
    shutil.copy(os.path.join(REF_SRC, "main.cpp"), tmp_path / "main.cpp")
    r = _syntax_only(str(tmp_path / "main.cpp"), os.path.join(ROOT, "include", "arvx", "dropin"),
                     os.path.join(ROOT, "include"), *MOCKS, REF_SRC, cxx=cxx)
    assert r.returncode == 0, r.stderr[-4000:]
    # the headers under test were the ones read (not the reference's own, which sit in REF_SRC)
    deps = subprocess.run(["g++", "-std=c++17", "-MM", "-I" + os.path.join(ROOT, "include", "arvx", "dropin"),
                           "-I" + os.path.join(ROOT, "include"), *("-I" + m for m in MOCKS),
                           "-I" + REF_SRC, str(tmp_path / "main.cpp")], capture_output=True, text=True)
    assert deps.returncode == 0, deps.stderr
    for name in REPLACED:
        assert os.path.join(ROOT, "include", "arvx", "dropin", name) in deps.stdout, name
        assert os.path.join(REF_SRC, name) not in deps.stdout, name
    for kept in ("Calibration.h", "PoseEstimation.h", "Segmentation.h"):
        assert os.path.join(REF_SRC, kept) in deps.stdout, kept


@needs_reference
def test_reference_main_compiles_with_the_include_swap(tmp_path):
    """INTEGRATION.md section 2, second form: only the six include lines change
    (`"Model.h"` -> `"arvx/dropin/Model.h"` ...), the reference's files stay where they are."""
    text = open(os.path.join(REF_SRC, "main.cpp")).read()
    changed = 0
    for name in REPLACED:
        text, n = re.subn(r'#include "%s"' % re.escape(name), '#include "arvx/dropin/%s"' % name, text)
        changed += n
    assert changed == 5  # main.cpp includes five of the six (Model.h comes through the others)
    (tmp_path / "main.cpp").write_text(text)
    r = _syntax_only(str(tmp_path / "main.cpp"), os.path.join(ROOT, "include"), *MOCKS, REF_SRC)
    assert r.returncode == 0, r.stderr[-4000:]


@needs_reference
def test_reference_main_needs_the_dropin_headers(tmp_path):
    """The check has teeth: without the drop-in headers the same command fails."""
    shutil.copy(os.path.join(REF_SRC, "main.cpp"), tmp_path / "main.cpp")
    r = _syntax_only(str(tmp_path / "main.cpp"), *MOCKS)
    assert r.returncode != 0


@pytest.mark.parametrize("cxx", COMPILERS)
@pytest.mark.parametrize("eigen", [False, True])
def test_dropin_surface_type_checks(eigen, cxx):
    """tests/cpp/dropin_typecheck.cpp: the five free functions and Model / SimpleMesh / Benchmark
    used the way the reference's replaced sources use them, with the stand-in vector types and
    with (mock) Eigen's; -Werror."""
    dirs = [os.path.join(ROOT, "include", "arvx", "dropin"), os.path.join(ROOT, "include"), MOCKS[0]]
    if eigen:
        dirs.append(MOCKS[1])
    r = _syntax_only(os.path.join(ROOT, "tests", "cpp", "dropin_typecheck.cpp"), *dirs,
                     extra=("-Wall", "-Wextra", "-Werror"), cxx=cxx)
    assert r.returncode == 0, r.stderr[-4000:]
