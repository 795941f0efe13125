"""In-tree native builds: libarvx.so (hipcc, gfx950), the C++ host layer, and --
as test infrastructure only -- the CPU oracle (gcc)."""
from __future__ import annotations

import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make(directory: str, *targets: str) -> None:
    cmd = ["make", "-C", os.path.join(ROOT, directory), *targets]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed:\n{res.stdout}")


# the kernels of a bench step, the file that holds their launch geometry, and the compiler flags
STEP_KERNEL_SOURCES = ("arvx_device.h", "carve_kernels.h", "views_kernels.h", "arvx_ctx.h",
                       "arvx_capi.hip", "Makefile")


def source_stamp() -> str:
    """sha256 (16 hex digits) over the sources of the kernels of a bench step (view derivation +
    carve), their launch code (arvx_capi.hip, arvx_ctx.h) and the build's flags (the Makefile):
    what ties a PMC pass under profiles/ (tools/make_traffic.py) to the build bench.py is
    running."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ar_voxel_project_amd", "csrc")
    for name in STEP_KERNEL_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def build_library() -> str:
    """hipcc --offload-arch=gfx950 -> ar_voxel_project_amd/lib/libarvx.so"""
    _make("ar_voxel_project_amd/csrc")
    return os.path.join(ROOT, "ar_voxel_project_amd", "lib", "libarvx.so")


def build_oracle() -> str:
    """gcc -> oracle/libarvx_oracle.so (checker for tests/smoke/cpu_baseline only)"""
    _make("oracle")
    return os.path.join(ROOT, "oracle", "libarvx_oracle.so")


def build_host_tests() -> str:
    """g++ -> tests/cpp/test_host: the C++ host layer (include/arvx/*.hpp) over libarvx.so"""
    build_library()
    _make("tests/cpp")
    _make("tools/cpp", "all")  # arvx_bench6 (the reference's -c=6 table), arvx_dropin_time
    return os.path.join(ROOT, "tests", "cpp", "test_host")


if __name__ == "__main__":
    print(build_library())
    print(build_oracle())
    print(build_host_tests())
