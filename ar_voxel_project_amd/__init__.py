"""MI355X-native voxel carving (hot path of alxfox/AR_Voxel_Project).

`capi` binds libarvx.so, the gfx950 library behind the C-ABI in
include/arvx/arvx.h; `synthetic` generates the analytic test/bench scenes;
`build` compiles the native code in-tree.
"""
from . import capi, synthetic  # noqa: F401

__all__ = ["capi", "synthetic"]
