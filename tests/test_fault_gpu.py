"""What the host does when a kernel that waits for other workgroups gave up (the page-locked fault
word, csrc/arvx_ctx.h): the call that synchronises next fails with ARVX_ERR_HIP, and the calls after
it are correct again -- ticket counters, status granules, chunk counts and packets start from clean
memory.  The mark itself is injected (libarvx_experiments.so: arvx_experiment_mark_fault): no kernel
has ever been seen to leave one.  And the lists that outgrow their buffers: a context
that has colour / closure / mesh lists sized for a small model is handed a large one -- the calls'
second attempt, deterministically."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import scenes

pytestmark = pytest.mark.gpu


def pipeline(ctx, arvx, sc):
    ctx.reset()
    ctx.set_views(sc.M, sc.masks, campos=sc.campos)
    ctx.set_images(sc.images)
    ctx.carve()
    ctx.color(arvx.COLOR_AVERAGE)
    idx, rgb = ctx.surface()
    ctx.handle_unseen()
    fidx, frgba = ctx.closure(3, True)
    verts, faces = ctx.mc_mesh(True)
    return [np.asarray(a) for a in (idx, rgb, fidx, frgba, verts, faces)], ctx.stats()["host_total_fallbacks"]


def test_failed_call_and_recovery(arvx):
    exp = os.path.join(os.path.dirname(arvx.LIB_PATH), "libarvx_experiments.so")
    if not os.path.exists(exp):
        pytest.fail("libarvx_experiments.so is missing: run __graft_entry__.build()")
    N, V = 96, 6
    sc = scenes.syn.sphere_scene(N, V, W=320, H=240, with_images=True)
    with arvx.Context(N, N, N, sc.voxel_size, lib_path=exp) as ctx:
        want, _ = pipeline(ctx, arvx, sc)
        ctx._lib.arvx_experiment_mark_fault.argtypes = [C.c_void_p, C.c_uint]
        assert ctx._lib.arvx_experiment_mark_fault(ctx._h, 6) == 0
        with pytest.raises(arvx.ArvxError, match="gave up waiting"):
            ctx.synchronize()
        ctx.synchronize()  # the mark is consumed
        got, fallbacks = pipeline(ctx, arvx, sc)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
        assert fallbacks == 0
        # ... also when the mark is met in the middle of a pipeline
        ctx.reset()
        ctx.set_views(sc.M, sc.masks, campos=sc.campos)
        ctx.set_images(sc.images)
        ctx.carve()
        ctx._lib.arvx_experiment_mark_fault(ctx._h, 1)
        with pytest.raises(arvx.ArvxError, match="gave up waiting"):
            ctx.color(arvx.COLOR_AVERAGE)
        got, _ = pipeline(ctx, arvx, sc)
        for a, b in zip(got, want):
            assert np.array_equal(a, b)


def test_lists_outgrow_their_buffers(arvx):
    """Small model first (its lists size the buffers), then a large one on the same context: colour
    pass, closure and mesh each repeat once with room for all; the results equal a fresh context's."""
    N, V = 128, 6
    small = scenes.syn.sphere_scene(N, V, W=320, H=240, with_images=True, radius_factor=0.06)
    large = scenes.syn.sphere_scene(N, V, W=320, H=240, with_images=True)
    with arvx.Context(N, N, N, large.voxel_size) as fresh:
        want, _ = pipeline(fresh, arvx, large)
    with arvx.Context(N, N, N, large.voxel_size) as ctx:
        tiny, _ = pipeline(ctx, arvx, small)
        got, fallbacks = pipeline(ctx, arvx, large)
    assert 0 < len(tiny[0]) * 8 < len(want[0]) and 0 < len(tiny[2]) * 8 < len(want[2])
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    assert fallbacks == 0
