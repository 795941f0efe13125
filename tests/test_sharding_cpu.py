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
        assert Z % (8 * world) == 0 and X % 8 == 0
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


def _worker(rank, world, port, mode, tmpdir, layout="slab"):
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
        ex = sharding.OccupancyExchange(X, Y, Z, world, rank, "cpu", mode=mode, buffers=2,
                                        layout=layout)
        # the rank contributes ONLY the planes it owns
        if layout == "striped":
            planes = sharding.stripe_planes(Z, world, rank)
        else:
            planes = list(range(ex.z0, ex.z1))
        wpp = X * Y // 32  # words per plane
        for step in range(3):  # buffers are reused across steps
            b = step % 2
            ex.prepare(b)
            buf = ex.full[b].numpy()
            for z in planes:  # what arvx_pack_occupancy_global does on the device
                buf[z * wpp:(z + 1) * wpp] = pack_bits(full[z])
            ex.launch(b, async_op=True)
        ex.wait_all()
        want = pack_bits(full)
        for b in range(2):
            got = ex.full[b].numpy()
            assert np.array_equal(got, want), f"rank {rank} buffer {b} mode {mode}"
        open(os.path.join(tmpdir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,layout", [("allreduce", "slab"), ("allgather", "slab"),
                                         ("allreduce", "striped")])
def test_occupancy_exchange_world2_gloo(tmp_path, oracle, mode, layout):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path), layout), nprocs=world,
             join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


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
    with pytest.raises(ValueError):
        sharding.OccupancyExchange(8, 8, 12, 2, 0, "cpu", layout="striped")
