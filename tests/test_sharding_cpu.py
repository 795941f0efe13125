"""The multi-GPU path's host logic on CPU: Z-slab partition + the end-of-carve
occupancy collective, run with world_size 2 over gloo.  Each rank's slab comes
from the CPU oracle; the merged plane must equal the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ar_voxel_project_amd import sharding


def pack_bits(state_flat):
    """Same layout as arvx_pack_occupancy: voxel i -> bit i%32 of word i//32."""
    occ = (np.asarray(state_flat).reshape(-1) & 1).astype(np.uint8)
    pad = (-len(occ)) % 32
    occ = np.concatenate([occ, np.zeros(pad, np.uint8)])
    return np.packbits(occ, bitorder="little").view(np.int32).copy()


def test_grid_and_slabs():
    assert sharding.grid_for(1, 512) == (512, 512, 512)
    for world in (2, 4, 8):
        X, Y, Z = sharding.grid_for(world, 512)
        assert Z % (8 * world) == 0 and X % 32 == 0
        ratio = X * Y * Z / (world * 512 ** 3)
        assert 0.93 < ratio < 1.07, (world, X, Y, Z, ratio)
        slabs = [sharding.slab_of(Z, world, r) for r in range(world)]
        assert slabs[0][0] == 0 and slabs[-1][1] == Z
        assert all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
        assert len({b - a for a, b in slabs}) == 1
    assert sharding.grid_for(8, 512) == (1024, 1024, 1024)
    assert [sharding.slab_of(10, 3, r) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, tmpdir, layout="slab", tight_cap=None):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle
        from tests import scenes
        X, Y, Z, V = 16, 16, 32, 4
        sc = scenes.small_sphere(32, V, W=96, H=72)
        s = np.float32(0.512 / 32)
        full = pyoracle.carve(X, Y, Z, s, sc.M, sc.masks, threads=1)
        codec = None
        if mode == "compressed":
            from tests import occ_codec
            codec = occ_codec.NumpyCodec()
        ex = sharding.OccupancyExchange(X, Y, Z, world, rank, "cpu", mode=mode, buffers=2,
                                        layout=layout, codec=codec)
        # the rank contributes ONLY the planes it owns
        if layout == "striped":
            planes = sharding.stripe_planes(Z, world, rank)
        else:
            planes = list(range(ex.z0, ex.z1))
        wpp = X * Y // 32  # words per plane
        for step in range(3):  # buffers are reused across steps
            b = step % 2
            ex.prepare(b)
            if mode == "compressed" and layout == "striped":
                # what arvx_pack_occupancy does on the device: the rank's planes, local order
                loc = ex.local[b].numpy().view(np.int32)
                for k, z in enumerate(planes):
                    loc[k * wpp:(k + 1) * wpp] = pack_bits(full[z])
            else:
                buf = ex.full[b].numpy()
                for z in planes:  # what arvx_pack_occupancy_global does on the device
                    buf[z * wpp:(z + 1) * wpp] = pack_bits(full[z])
            ex.launch(b, async_op=True)
            if mode == "compressed" and step == 0:
                # the first exchange ships worst-case packets; size the next ones to need
                ex.wait_all()
                assert ex.cap == ex.cap_max
                cap = ex.retune(0)
                assert 16 <= cap <= ex.cap_max
                if tight_cap is not None:
                    ex.cap = tight_cap  # too small for the sphere's mixed words: must fall back
        ex.wait_all()
        want = pack_bits(full)
        for b in range(2):
            got = ex.full[b].numpy()
            assert np.array_equal(got, want), f"rank {rank} buffer {b} mode {mode}"
        if mode == "compressed":
            assert ex.fallbacks == (2 if tight_cap is not None else 0)
            assert not ex.overflowed()
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,layout", [("allreduce", "slab"), ("allgather", "slab"),
                                         ("allreduce", "striped"), ("compressed", "slab"),
                                         ("compressed", "striped")])
def test_occupancy_exchange_world2_gloo(tmp_path, oracle, mode, layout):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), layout), nprocs=world,
             join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("layout", ["slab", "striped"])
def test_compressed_exchange_overflow_falls_back_gloo(tmp_path, oracle, layout):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), "compressed", str(tmp_path), layout, 1),
             nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def _fused_worker(rank, world, port, tmpdir, layout):
    """The fused hand-off's host logic (pack() = Context.occupancy_pack_compress, the expansion that
    skips the caller's packet) with a packet cap that only ONE rank's packet outgrows, and with the
    context carving the next job before the exchange is waited for."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import occ_codec
        X, Y, Z = 32, 16, 32
        n = X * Y * (Z // world) // 64
        wpg = X * Y * 8 // 64 if layout == "striped" else 0
        rng = np.random.default_rng(11)  # the same numbers on every rank

        def job_words(mixed_of_rank):
            out = []
            for q in range(world):
                w = np.where(rng.random(n) < 0.5, occ_codec.ONES, occ_codec.U64(0))
                at = rng.choice(n, mixed_of_rank[q], replace=False)
                w[at] = rng.integers(1, 2 ** 62, len(at), dtype=np.uint64)
                out.append(w)
            return out

        def merged(words):
            if layout == "striped":
                full = np.zeros(world * n, np.uint64)
                for q in range(world):
                    full.reshape(-1, world, wpg)[:, q, :] = words[q].reshape(-1, wpg)
                return full
            return np.concatenate(words)

        jobs = [job_words([5, 40]), job_words([6, 41]), job_words([30, 3])]
        codec = occ_codec.FusedNumpyCodec(world, rank, wpg)
        ex = sharding.OccupancyExchange(X, Y, Z, world, rank, "cpu", mode="compressed", buffers=2,
                                        layout=layout, codec=codec)
        assert ex.can_fuse and ex.n64 == n
        codec.carve(jobs[0][rank])
        ex.prepare(0)
        ex.pack(codec, 0)
        ex.launch(0)
        ex.wait_all()
        assert np.array_equal(ex.full[0].numpy().view(np.uint64), merged(jobs[0]))
        assert ex.retune(0) >= 40 and ex.fallbacks == 0
        ex.cap = 20  # rank 0's packets fit, rank 1's (41 mixed words) do not
        codec.carve(jobs[1][rank])
        ex.prepare(1)
        ex.pack(codec, 1)
        ex.launch(1, async_op=True)
        codec.carve(jobs[2][rank])  # the context moves on to the next job before the wait
        ex.wait(1)
        assert ex.fallbacks == 1, f"rank {rank}: every rank must see the one packet that overflowed"
        assert np.array_equal(ex.full[1].numpy().view(np.uint64), merged(jobs[1])), \
            f"rank {rank}: the repaired exchange ships the job that was packed"
        assert not ex.overflowed()
        # and the exchange keeps working afterwards (cap is back at its worst case)
        ex.prepare(0)
        ex.pack(codec, 0)
        ex.launch(0)
        ex.wait_all()
        assert np.array_equal(ex.full[0].numpy().view(np.uint64), merged(jobs[2]))
        assert ex.fallbacks == 1
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def _lazy_worker(rank, world, port, tmpdir, layout):
    """lazy_expand: the exchange's product stays packets; merged_plane() builds the plane on demand;
    an overflowing packet is seen from the headers alone."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from tests import occ_codec
        X, Y, Z = 32, 16, 32
        n = X * Y * (Z // world) // 64
        wpg = X * Y * 8 // 64 if layout == "striped" else 0
        rng = np.random.default_rng(23)
        words = []
        for q in range(world):
            w = np.where(rng.random(n) < 0.5, occ_codec.ONES, occ_codec.U64(0))
            at = rng.choice(n, 9 + 20 * q, replace=False)
            w[at] = rng.integers(1, 2 ** 62, len(at), dtype=np.uint64)
            words.append(w)
        if layout == "striped":
            want = np.zeros(world * n, np.uint64)
            for q in range(world):
                want.reshape(-1, world, wpg)[:, q, :] = words[q].reshape(-1, wpg)
        else:
            want = np.concatenate(words)
        codec = occ_codec.FusedNumpyCodec(world, rank, wpg)
        ex = sharding.OccupancyExchange(X, Y, Z, world, rank, "cpu", mode="compressed", buffers=2,
                                        layout=layout, codec=codec, lazy_expand=True)
        codec.carve(words[rank])
        ex.prepare(0)
        ex.pack(codec, 0)
        ex.launch(0)
        ex.wait_all()
        assert ex.unexpanded[0] and not ex.overflowed()  # still packets: nothing was expanded
        own = ex.full[0].numpy().view(np.uint64).copy()
        assert not np.array_equal(own, want)  # (only the rank's own words are in the plane so far)
        assert np.array_equal(ex.merged_plane(0).numpy().view(np.uint64), want)
        assert not ex.unexpanded[0]
        ex.retune(0)
        ex.cap = 12  # rank 0's 9 mixed words fit, rank 1's 29 do not
        ex.prepare(1)
        ex.pack(codec, 1)
        ex.launch(1, async_op=True)
        ex.wait(1, verify=False)  # a pipelined caller: nothing read yet ...
        assert ex.overflowed()    # ... the headers say it at the caller's own synchronisation point
        ex.wait(1)                # and the exchange is repaired (every rank: a collective)
        assert ex.fallbacks == 1
        assert np.array_equal(ex.merged_plane(1).numpy().view(np.uint64), want)
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("layout", ["slab", "striped"])
def test_lazy_exchange_keeps_packets_gloo(tmp_path, layout):
    world = 2
    mp.spawn(_lazy_worker, args=(world, _free_port(), str(tmp_path), layout), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


@pytest.mark.parametrize("layout", ["slab", "striped"])
def test_fused_exchange_one_rank_overflows_gloo(tmp_path, layout):
    """a packet cap that only one rank's packet outgrows must send EVERY rank
    into the repair (the overflowing rank skips its own packet in the expansion, so it has to look
    at its header all the same) -- otherwise one rank enters a collective alone; (medium) and the
    repair re-ships the job that was packed, not whatever the context holds by then."""
    world = 2
    mp.spawn(_fused_worker, args=(world, _free_port(), str(tmp_path), layout), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_occupancy_packet_restatement():
    """The numpy packet codec round-trips and flags overflow (it is the GPU tests' checker)."""
    from tests import occ_codec
    rng = np.random.default_rng(5)
    for n in (1, 63, 64, 65, 1000, 4096 + 7):
        kind = rng.integers(0, 4, n)
        w = rng.integers(1, 2 ** 63, n, dtype=np.uint64)
        w[kind == 0] = 0
        w[kind == 1] = occ_codec.ONES
        nm = int(((w != 0) & (w != occ_codec.ONES)).sum())
        world = 3
        for cap, over in ((nm, False), (nm + 5, False), (max(nm - 1, 0), nm > 0)):
            S = occ_codec.header_words(n) + cap
            assert S == sharding.packet_header_words(n) + cap
            pk = occ_codec.compress(w, cap)
            assert len(pk) == S and int(pk[0]) == nm
            full = np.full(world * n, 0x5A5A, np.uint64)
            got_over = occ_codec.expand(np.tile(pk, world), world, 1, n, cap, full)
            assert got_over == over
            assert np.all(full[n:2 * n] == 0x5A5A)  # own slab untouched
            if not over:
                assert np.array_equal(full[:n], w) and np.array_equal(full[2 * n:], w)
            else:
                assert np.all(full[:n] == 0x5A5A)


def test_stripe_planes():
    assert sharding.stripe_planes(32, 2, 1) == list(range(8, 16)) + list(range(24, 32))
    got = sorted(z for r in range(3) for z in sharding.stripe_planes(40, 3, r))
    assert got == list(range(40))
    from ar_voxel_project_amd import capi
    assert list(capi.stripe_planes(40, 3, 1)) == sharding.stripe_planes(40, 3, 1)


def test_exchange_rejects_unaligned_slabs():
    with pytest.raises(ValueError):
        sharding.OccupancyExchange(5, 3, 4, 2, 1, "cpu")
    with pytest.raises(ValueError):
        sharding.OccupancyExchange(8, 8, 9, 2, 0, "cpu", mode="allgather")
    with pytest.raises(ValueError):
        sharding.OccupancyExchange(8, 8, 16, 2, 0, "cpu", mode="allgather", layout="striped")
    with pytest.raises(ValueError):  # Z must hold whole stripes of every rank
        sharding.OccupancyExchange(8, 8, 24, 2, 0, "cpu", mode="compressed", layout="striped",
                                   codec=object())
    with pytest.raises(ValueError):
        sharding.OccupancyExchange(8, 8, 12, 2, 0, "cpu", layout="striped")


def test_merge_mesh_restores_the_visiting_order():
    """sharding.merge_mesh: a whole-grid mesh cut into the parts three slabs would deliver (cells
    by the owner of their upper plane, each part in the reference's order) and put together again
    is the mesh; the per-cell triangle counts come from the table the kernels use."""
    from ar_voxel_project_amd import sharding
    from oracle import pyoracle
    rng = np.random.default_rng(3)
    X, Y, Z = 12, 9, 14
    state = (rng.random((Z, Y, X)) < 0.4).astype(np.uint8) * 3
    model = pyoracle.model_from_state(state)
    model[:, :3] = np.where(model[:, 3:4] != 0, rng.integers(0, 256, size=(X * Y * Z, 3)), 0)
    cells = pyoracle.mc_cells(X, Y, Z, model)
    verts, rgb = pyoracle.mc_mesh(X, Y, Z, model)
    ntri = sharding.triangles_per_cube_index()
    count = ntri[cells[:, 3]]
    assert count.sum() == len(rgb) and count.min() >= 1 and count.max() <= 5
    start = np.concatenate([[0], np.cumsum(count)[:-1]])
    parts = []
    for z0, z1 in ((0, 5), (5, 6), (6, Z)):
        lo, hi = z0 - 1, (Z if z1 == Z else z1 - 1)  # cells whose upper plane the slab owns
        sel = np.flatnonzero((cells[:, 2] >= lo) & (cells[:, 2] < hi))
        tri = np.concatenate([np.arange(start[c], start[c] + count[c]) for c in sel])
        parts.append((cells[sel], verts.reshape(-1, 9)[tri].reshape(-1, 3), rgb[tri]))
    c2, v2, r2 = sharding.merge_mesh(parts)
    assert np.array_equal(c2, cells) and np.array_equal(v2, verts) and np.array_equal(r2, rgb)
