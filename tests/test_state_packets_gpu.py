"""The carved model's hand-off to the host as compressed packets (arvx_state_download_packets,
include/arvx/arvx.h) against the plane form (arvx_state_download_planes): decoded with the numpy
restatement of the packet (tests/occ_codec.py) both planes must come back bit for bit -- on a
carved model (lazy coarse codes), an uploaded one (every record written), after handleUnseen, on a
slab; too small a buffer is reported and the second call delivers; the packets follow the state.
Reference: the model a caller reads after carve(), src/main.cpp:262-303 / src/Model.h:119-160."""
import numpy as np
import pytest

from tests import occ_codec, scenes

pytestmark = pytest.mark.gpu


def decode(packet, n, need):
    H = occ_codec.header_words(n)
    full = np.zeros(n, np.uint64)
    assert not occ_codec.expand(packet[:H + need], 1, -1, n, need, full)
    return full


def planes64(ctx):
    occ, seen = ctx.download_planes()
    return occ.view(np.uint64), seen.view(np.uint64)


def check(ctx):
    n, H = ctx.packet_geometry()
    assert H == occ_codec.header_words(n)
    po, ps, no, ns = ctx.download_packets()
    occ, seen = planes64(ctx)
    assert n == occ.size
    assert int(po[0]) == no == int(((occ != 0) & (occ != occ_codec.ONES)).sum())
    assert int(ps[0]) == ns == int(((seen != 0) & (seen != occ_codec.ONES)).sum())
    assert np.array_equal(decode(po, n, no), occ), "occupancy"
    assert np.array_equal(decode(ps, n, ns), seen), "seen"
    return no, ns


@pytest.mark.parametrize("X,Y,Z,kw", [(128, 96, 64, {}), (64, 64, 24, {}), (96, 64, 40, {"z_range": (8, 32)}),
                                      (256, 128, 48, {"z_range": (0, 17), "halo": 3}), (32, 2, 1, {})])
def test_packets_equal_planes(arvx, X, Y, Z, kw):
    V, W, H = 6, 320, 240
    s = np.float32(0.3 / max(X, Y, Z))
    _, _, M = scenes.random_cameras(V, 0.3, seed=X + Z, W=W, H=H)
    masks = scenes.noise_masks(V, H, W, block=24, p_bg=0.45, seed=X)
    with arvx.Context(X, Y, Z, s, **kw) as ctx:
        check(ctx)  # a fresh model: all occupied, nothing seen
        ctx.set_views(M, masks)
        ctx.carve_views(0, 2)
        check(ctx)  # (lazy coarse codes; some voxels seen by nobody yet)
        ctx.carve()
        no, _ = check(ctx)
        # too small a buffer: reported, and the second call (no kernel: the device kept the packets) delivers
        n, Hh = ctx.packet_geometry()
        if no > 1:
            po, ps, no2, _ = ctx.download_packets(occ=np.zeros(Hh + no // 2, np.uint64))
            assert no2 == no and po.size == Hh + no
            assert np.array_equal(decode(po, n, no), planes64(ctx)[0])
        ctx.handle_unseen()
        check(ctx)
        st = ctx.download_state()
        rng = np.random.default_rng(5)
        st = np.where(rng.random(st.shape) < 0.02, st ^ 1, st).astype(np.uint8)
        ctx.upload_state(st)
        check(ctx)
        ctx.reset()
        check(ctx)


def test_packets_need_whole_words(arvx):
    with arvx.Context(40, 16, 8, 0.01) as ctx:
        with pytest.raises(arvx.ArvxError, match="X % 32"):
            ctx.packet_geometry()
