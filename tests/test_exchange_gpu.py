"""The compressed occupancy packet (arvx_occupancy_compress / arvx_occupancy_expand,
include/arvx/arvx.h) against its numpy restatement, bit for bit, and end to end on a
carved slab: pack -> compress -> expand must give back the packed words."""
import numpy as np
import pytest
import torch

from tests import occ_codec, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from ar_voxel_project_amd import capi
    c = capi.Context(16, 16, 16, 0.01, device=0)
    yield c
    c.close()


def _words(rng, n, p_zero, p_one):
    kind = rng.random(n)
    w = rng.integers(1, 2 ** 63, n, dtype=np.uint64) | (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63))
    w[(w == occ_codec.ONES)] = 7
    w[kind < p_zero] = 0
    w[(kind >= p_zero) & (kind < p_zero + p_one)] = occ_codec.ONES
    return w


def _dev(a):
    return torch.from_numpy(a.view(np.int64)).cuda()


def _settle():
    """The library runs on its own non-blocking stream: torch's fills and uploads (null
    stream) must have landed before it touches the buffers."""
    torch.cuda.synchronize()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 4096, 4096 * 64 + 17, 1 << 21])
@pytest.mark.parametrize("mix", [(0.5, 0.4), (0.0, 0.0), (1.0, 0.0), (0.0, 1.0)])
def test_compress_matches_restatement(ctx, n, mix):
    from ar_voxel_project_amd import capi
    rng = np.random.default_rng(n * 7 + int(mix[0] * 10))
    w = _words(rng, n, *mix)
    nm = int(((w != 0) & (w != occ_codec.ONES)).sum())
    for cap in {nm, nm + 9, nm // 2}:
        S = capi.occupancy_packet_words(n, cap)
        assert S == occ_codec.header_words(n) + cap
        d_w = _dev(w)
        d_pk = torch.full((S,), 0, dtype=torch.int64, device="cuda")
        _settle()
        ctx.occupancy_compress(d_w.data_ptr(), n, d_pk.data_ptr(), cap)
        ctx.synchronize()
        got = d_pk.cpu().numpy().view(np.uint64)
        want = occ_codec.compress(w, cap)
        H = occ_codec.header_words(n)
        nb = (n + 63) // 64
        assert int(got[0]) == nm
        assert np.array_equal(got[1:1 + 2 * nb], want[1:1 + 2 * nb])
        # the u32 offsets: nb of them (an odd nb leaves half a word unused)
        assert np.array_equal(got[1 + 2 * nb:H].view(np.uint32)[:nb],
                              want[1 + 2 * nb:H].view(np.uint32)[:nb])
        k = min(nm, cap)
        assert np.array_equal(got[H:H + k], want[H:H + k])


@pytest.mark.parametrize("n,world,me", [(65, 2, 0), (4096 + 5, 3, 1), (1 << 18, 8, 7)])
def test_expand_matches_restatement(ctx, n, world, me):
    rng = np.random.default_rng(n + world)
    slabs = [_words(rng, n, 0.6, 0.3) for _ in range(world)]
    need = max(int(((w != 0) & (w != occ_codec.ONES)).sum()) for w in slabs)
    for cap, over in ((need + 3, False), (need - 1, True)):
        S = occ_codec.header_words(n) + cap
        d_pk = torch.zeros(world * S, dtype=torch.int64, device="cuda")
        d_ws = [_dev(w) for w in slabs]
        _settle()
        for q, d_w in enumerate(d_ws):
            ctx.occupancy_compress(d_w.data_ptr(), n, d_pk[q * S:].data_ptr(), cap)
        ctx.synchronize()
        pk = d_pk.cpu().numpy().view(np.uint64)
        for q, w in enumerate(slabs):
            k = min(int(pk[q * S]), cap)
            H = occ_codec.header_words(n)
            assert np.array_equal(pk[q * S + H:q * S + H + k], occ_codec.compress(w, cap)[H:H + k])
        full0 = rng.integers(0, 2 ** 62, world * n, dtype=np.uint64)
        d_full = _dev(full0)
        d_flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        _settle()
        ctx.occupancy_expand(d_pk.data_ptr(), world, me, n, cap, d_full.data_ptr(),
                             d_flag.data_ptr())
        ctx.synchronize()
        want = full0.copy()
        want_over = occ_codec.expand(pk, world, me, n, cap, want)
        assert want_over == over and bool(d_flag.item()) == over
        assert np.array_equal(d_full.cpu().numpy().view(np.uint64), want)
        if not over:
            for q, w in enumerate(slabs):
                if q != me:
                    assert np.array_equal(want[q * n:(q + 1) * n], w)


def test_carved_slabs_round_trip():
    """Two slab contexts carve a sphere; each packs and compresses its slab; expanding the
    two packets on either side gives the packed occupancy of the whole grid."""
    from ar_voxel_project_amd import capi
    X = Y = Z = 64
    sc = scenes.small_sphere(64, 6, W=160, H=120)
    world = 2
    n = X * Y * (Z // world) // 64
    packed, packets = [], []
    cap = n
    S = capi.occupancy_packet_words(n, cap)
    for r in range(world):
        c = capi.Context(X, Y, Z, sc.voxel_size, device=0, z_range=(r * Z // world, (r + 1) * Z // world))
        c.set_views(sc.M, sc.masks)
        c.carve()
        d_w = torch.zeros(n, dtype=torch.int64, device="cuda")
        d_pk = torch.zeros(S, dtype=torch.int64, device="cuda")
        _settle()
        c.pack_occupancy(d_w.data_ptr())
        c.occupancy_compress(d_w.data_ptr(), n, d_pk.data_ptr(), cap)
        c.synchronize()
        packed.append(d_w)
        packets.append(d_pk)
        c.close()
    assert int(packets[0][0]) > 0 and int(packets[0][0]) < n  # a sphere: some mixed words, not all
    both = torch.cat(packets)
    c = capi.Context(8, 8, 8, 0.01, device=0)
    for me in range(world):
        d_full = torch.zeros(world * n, dtype=torch.int64, device="cuda")
        d_full[me * n:(me + 1) * n] = packed[me]
        d_flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        _settle()
        c.occupancy_expand(both.data_ptr(), world, me, n, cap, d_full.data_ptr(), d_flag.data_ptr())
        c.synchronize()
        assert not d_flag.item()
        assert torch.equal(d_full, torch.cat(packed))
    c.close()


@pytest.mark.parametrize("N,V,W,H", [(64, 6, 160, 120), (256, 36, 640, 480)])
def test_exchange_stream_pipeline(N, V, W, H):
    """arvx_ctx_set_exchange_stream: the hand-off of job k (pack, compress, expand) on a second
    stream beside the views + carve of job k + 1, ordered by two events -- as bench.py and a
    multi-GPU caller run it.  Jobs alternate between two scenes; every expanded plane must be
    the packed occupancy of ITS job.  At 256^3 x 36 views of 640x480 the compress of job k
    (262 144 words) really runs beside the summed-area-table kernels of job k + 1: the two
    must not share a work buffer (they once did: ctx->d_scratch)."""
    from ar_voxel_project_amd import capi
    X = Y = Z = N
    scs = [scenes.syn.sphere_scene(N, V, W=W, H=H), scenes.syn.box_scene((N, N, N), V, W=W, H=H)]
    n = X * Y * Z // 64
    cap = n
    S = capi.occupancy_packet_words(n, cap)
    main, side = torch.cuda.Stream(), torch.cuda.Stream()
    want = []
    with capi.Context(X, Y, Z, scs[0].voxel_size) as c:  # reference planes, one stream
        for sc in scs:
            c.set_views(sc.M, sc.masks)
            c.reset()
            c.carve()
            d = torch.zeros(n, dtype=torch.int64, device="cuda")
            _settle()
            c.pack_occupancy(d.data_ptr())
            c.synchronize()
            want.append(d.clone())
    assert not torch.equal(want[0], want[1])
    d_masks = [torch.from_numpy(np.ascontiguousarray(sc.masks)).cuda() for sc in scs]
    jobs = 8
    words = [torch.zeros(n, dtype=torch.int64, device="cuda") for _ in range(2)]
    packet = [torch.zeros(S, dtype=torch.int64, device="cuda") for _ in range(2)]
    full = [torch.zeros(n, dtype=torch.int64, device="cuda") for _ in range(jobs)]
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    _settle()
    with capi.Context(X, Y, Z, scs[0].voxel_size) as c:
        c.set_stream(main.cuda_stream)
        c.set_exchange_stream(side.cuda_stream)
        packed = None
        for k in range(jobs):
            sc, b = scs[k % 2], k % 2
            c.reset()
            c.set_views_device(sc.M, d_masks[k % 2].data_ptr(), sc.W, sc.H, 1)
            if packed is not None:
                main.wait_event(packed)  # the previous pack has read the records
            c.carve()
            carved = torch.cuda.Event()
            carved.record(main)
            side.wait_event(carved)
            c.pack_occupancy(words[b].data_ptr())
            packed = torch.cuda.Event()
            packed.record(side)
            c.occupancy_compress(words[b].data_ptr(), n, packet[b].data_ptr(), cap)
            c.occupancy_expand_striped(packet[b].data_ptr(), 1, n, cap, n, full[k].data_ptr(),
                                       flag.data_ptr())
        torch.cuda.synchronize()
        c.set_exchange_stream(0)
    assert not flag.item()
    for k in range(jobs):
        assert torch.equal(full[k], want[k % 2]), f"job {k}"


@pytest.mark.parametrize("wpg,groups,world", [(2, 3, 2), (32, 5, 3), (1024, 4, 8)])
def test_expand_striped_matches_restatement(ctx, wpg, groups, world):
    """Striped slabs: word i of rank q lands at ((i / wpg) * world + q) * wpg + i % wpg, for
    every rank including the caller; an overflowing packet leaves its words alone."""
    rng = np.random.default_rng(wpg + world)
    n = wpg * groups
    slabs = [_words(rng, n, 0.6, 0.3) for _ in range(world)]
    need = max(int(((w != 0) & (w != occ_codec.ONES)).sum()) for w in slabs)
    for cap, over in ((need + 1, False), (max(need - 1, 0), need > 0)):
        S = occ_codec.header_words(n) + cap
        d_pk = torch.zeros(world * S, dtype=torch.int64, device="cuda")
        d_ws = [_dev(w) for w in slabs]
        _settle()
        for q, d_w in enumerate(d_ws):
            ctx.occupancy_compress(d_w.data_ptr(), n, d_pk[q * S:].data_ptr(), cap)
        ctx.synchronize()
        pk = d_pk.cpu().numpy().view(np.uint64)
        full0 = rng.integers(0, 2 ** 62, world * n, dtype=np.uint64)
        d_full = _dev(full0)
        d_flag = torch.zeros(1, dtype=torch.int32, device="cuda")
        _settle()
        ctx.occupancy_expand_striped(d_pk.data_ptr(), world, n, cap, wpg, d_full.data_ptr(),
                                     d_flag.data_ptr())
        ctx.synchronize()
        want = full0.copy()
        assert occ_codec.expand_striped(pk, world, n, cap, wpg, want) == over
        assert bool(d_flag.item()) == over
        assert np.array_equal(d_full.cpu().numpy().view(np.uint64), want)
        if not over:
            for q, w in enumerate(slabs):
                for gl in range(groups):
                    g = gl * world + q
                    assert np.array_equal(want[g * wpg:(g + 1) * wpg], w[gl * wpg:(gl + 1) * wpg])


@pytest.mark.parametrize("X,Y,Z,kind", [(128, 96, 64, "whole"), (96, 64, 40, "slab"), (800 // 5, 96, 48, "striped"),
                                         (256, 128, 48, "striped"), (64, 64, 24, "whole")])
@pytest.mark.parametrize("state", ["carved", "uploaded"])
def test_pack_compress_is_pack_then_compress(X, Y, Z, kind, state):
    """arvx_occupancy_pack_compress (records -> packet, two launches) against arvx_pack_occupancy +
    arvx_occupancy_compress, bit for bit: the count, both bitmaps, the offsets, the mixed words --
    on a carved model (lazy coarse codes) and on an uploaded one (every record written); the
    rank's own words land at their place in the whole grid's plane; and the expansion that leaves
    the caller's packet out completes that plane."""
    from ar_voxel_project_amd import capi
    V, W, H = 6, 320, 240
    s = np.float32(0.3 / max(X, Y, Z))
    _, _, M = scenes.random_cameras(V, 0.3, seed=X + Z, W=W, H=H)
    masks = scenes.noise_masks(V, H, W, block=24, p_bg=0.45, seed=X)
    world, rank = (3, 1) if kind == "striped" else (1, 0)
    kw = {"stripes": (world, rank)} if kind == "striped" else ({"z_range": (8, 32)} if kind == "slab" else {})
    with capi.Context(X, Y, Z, s, **kw) as ctx:
        ctx.set_views(M, masks)
        ctx.carve()
        if state == "uploaded":
            st = ctx.download_state()
            rng = np.random.default_rng(5)
            st = np.where(rng.random(st.shape) < 0.02, st ^ 1, st).astype(np.uint8)
            ctx.upload_state(st)
        nz = len(ctx.planes)
        n = X * Y * nz // 64
        d_local = torch.zeros(n, dtype=torch.int64, device="cuda")
        d_full = torch.zeros(X * Y * Z // 64, dtype=torch.int64, device="cuda")
        _settle()
        ctx.pack_occupancy(d_local.data_ptr())
        ctx.synchronize()
        local = d_local.cpu().numpy().view(np.uint64)
        nm = int(((local != 0) & (local != occ_codec.ONES)).sum())
        for cap in (nm + 5, max(0, nm // 2)):
            S = capi.occupancy_packet_words(n, cap)
            d_a = torch.zeros(S, dtype=torch.int64, device="cuda")
            d_b = torch.zeros(S, dtype=torch.int64, device="cuda")
            d_full.zero_()
            _settle()
            ctx.occupancy_compress(d_local.data_ptr(), n, d_a.data_ptr(), cap)
            ctx.occupancy_pack_compress(d_b.data_ptr(), cap, d_full.data_ptr())
            ctx.synchronize()
            a, b = d_a.cpu().numpy().view(np.uint64), d_b.cpu().numpy().view(np.uint64)
            Hh, nb, k = occ_codec.header_words(n), (n + 63) // 64, min(nm, cap)
            assert int(b[0]) == nm == int(a[0])
            assert np.array_equal(b[1:1 + 2 * nb], a[1:1 + 2 * nb]), "bitmaps"
            assert np.array_equal(b[1 + 2 * nb:Hh].view(np.uint32)[:nb], a[1 + 2 * nb:Hh].view(np.uint32)[:nb]), "offsets"
            assert np.array_equal(b[Hh:Hh + k], a[Hh:Hh + k]), "mixed words"
            # the rank's own words at their place in the whole grid's plane, nothing anywhere else
            full = d_full.cpu().numpy().view(np.uint64).reshape(Z, -1)
            assert np.array_equal(full[ctx.planes].reshape(-1), local), "own words in the plane"
            other = np.ones(Z, bool)
            other[ctx.planes] = False
            assert not full[other].any()
        if kind == "striped":  # the other ranks' packets complete the plane; the caller's is skipped
            cap = nm + 5
            S = capi.occupancy_packet_words(n, cap)
            d_pks = torch.zeros(world * S, dtype=torch.int64, device="cuda")
            d_flag = torch.zeros(1, dtype=torch.int32, device="cuda")
            d_full.zero_()
            _settle()
            for q in range(world):  # (every "rank" ships this rank's planes: only the placement is looked at)
                ctx.occupancy_compress(d_local.data_ptr(), n, d_pks[q * S:].data_ptr(), cap)
            # poison the caller's own packet: it must not be read
            d_pks[rank * S:(rank + 1) * S] = -1
            ctx.occupancy_pack_compress(d_pks[rank * S:].data_ptr(), cap, d_full.data_ptr())
            d_pks[rank * S + 1:(rank + 1) * S] = -1
            _settle()
            ctx.occupancy_expand_striped_others(d_pks.data_ptr(), world, rank, n, cap, X * Y * 8 // 64,
                                                d_full.data_ptr(), d_flag.data_ptr())
            ctx.synchronize()
            full = d_full.cpu().numpy().view(np.uint64).reshape(Z // 8, 8, -1)
            assert int(d_flag.item()) == 0
            for q in range(world):  # group g of rank q = global group g * world + q
                got = full[q::world].reshape(-1)
                assert np.array_equal(got, local), f"rank {q}'s stripes"


@pytest.mark.parametrize("layout", ["slab", "striped"])
@pytest.mark.parametrize("who", ["self", "other", "nobody"])
def test_every_rank_sees_an_overflowing_packet(ctx, layout, who):
    """One packet of the gathered buffer announces more mixed words than the cap.  The expansion
    that skips the caller's own packet must still raise the flag when that packet is the caller's
    own -- every rank has to arrive at the same verdict, or one of them repairs the exchange (a
    collective) alone -- and must still expand the packets that fit."""
    world, me, wpg, groups = 3, 1, 32, 4
    n = wpg * groups
    rng = np.random.default_rng(17)
    slabs = [_words(rng, n, 0.6, 0.3) for _ in range(world)]
    need = [int(((w != 0) & (w != occ_codec.ONES)).sum()) for w in slabs]
    cap = max(need) + 2
    S = occ_codec.header_words(n) + cap
    d_pk = torch.zeros(world * S, dtype=torch.int64, device="cuda")
    d_ws = [_dev(w) for w in slabs]
    _settle()
    for q, d_w in enumerate(d_ws):
        ctx.occupancy_compress(d_w.data_ptr(), n, d_pk[q * S:].data_ptr(), cap)
    ctx.synchronize()
    over = {"self": me, "other": 2, "nobody": None}[who]
    if over is not None:
        d_pk[over * S] = cap + 1  # the header a packet carries when its slab outgrew the cap
    pk = d_pk.cpu().numpy().view(np.uint64)
    full0 = rng.integers(0, 2 ** 62, world * n, dtype=np.uint64)
    d_full = _dev(full0)
    d_flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    _settle()
    want = full0.copy()
    if layout == "slab":
        ctx.occupancy_expand(d_pk.data_ptr(), world, me, n, cap, d_full.data_ptr(), d_flag.data_ptr())
        want_over = occ_codec.expand(pk, world, me, n, cap, want)
    else:
        ctx.occupancy_expand_striped_others(d_pk.data_ptr(), world, me, n, cap, wpg, d_full.data_ptr(),
                                            d_flag.data_ptr())
        own = want.reshape(-1, world, wpg)[:, me, :].copy()
        want_over = occ_codec.expand_striped(pk, world, n, cap, wpg, want)
        want.reshape(-1, world, wpg)[:, me, :] = own
    ctx.synchronize()
    assert want_over == (over is not None)
    assert bool(d_flag.item()) == want_over
    assert np.array_equal(d_full.cpu().numpy().view(np.uint64), want)
