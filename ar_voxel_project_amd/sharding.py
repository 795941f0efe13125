"""Z-slab sharding of a voxel grid over the GPUs of one node.

Every voxel's carve result depends only on its own coordinates and on the
read-only views (reference src/VoxelCarving.cpp:39-55), so rank r simply carves
planes [z0, z1) of the grid.  No data-path exchange is needed to carve; the one
collective at the end merges the ranks' bit-packed occupancy planes so that
every rank holds the whole grid's occupancy.  Two interchangeable forms:

  allreduce  every rank holds a full-size word plane, zero outside its slab;
             SUM over ranks of int32 words (RCCL has no bitwise OR/AND; exactly
             one rank holds non-zero words at any position, so the wrapping sum
             equals the owner's words; MAX would be wrong for words whose top
             bit is set, since the words travel as signed int32)
  allgather  the slabs are equal-sized, so the full plane is simply the
             concatenation of the ranks' slab words

Two ways to cut the grid:

  slab     rank r owns the contiguous planes slab_of(Z, world, r)
  striped  with the planes cut into groups of 8, rank r owns groups r, r+world, ...
           Surface voxels (where the per-voxel work is) cluster in z, so contiguous
           slabs leave the ranks that own empty space idle; stripes balance the load.
           Only the allreduce form applies (a rank's words are scattered).

Host logic only: runs on CPU tensors with gloo (tests) and on GPU tensors with
nccl == RCCL (bench.py).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def grid_for(world: int, base: int) -> Tuple[int, int, int]:
    """Grid holding ~world * base^3 voxels: X = Y = a multiple of 8 close to
    base * world^(1/3), Z the nearest multiple of 8*world, so that all slabs are
    equal and start on a 32-voxel boundary of the packed plane."""
    if world == 1:
        return base, base, base
    n = int(round(base * world ** (1.0 / 3.0) / 8.0)) * 8
    z = max(1, int(round(n / (8.0 * world)))) * 8 * world
    return n, n, z


def slab_of(Z: int, world: int, rank: int) -> Tuple[int, int]:
    """Planes [z0, z1) owned by `rank`. Equal slabs when world divides Z, else the
    first Z % world ranks get one plane more."""
    q, r = divmod(Z, world)
    z0 = rank * q + min(rank, r)
    return z0, z0 + q + (1 if rank < r else 0)


def stripe_planes(Z: int, world: int, rank: int):
    """Global z of the planes rank owns under the striped layout, in local order."""
    return [g * 8 + k for g in range(rank, Z // 8, world) for k in range(8)]


def merge_mc_cells(per_rank):
    """Marching-cubes cell lists of the slab contexts (Context.mc_cells, one (n,4)
    array per rank) -> one list in the reference's visiting order (x outermost,
    then y, then z; src/MarchingCubes.cpp:12-18).  Slabs own disjoint z ranges, so
    this is a stable sort of the concatenation by (x, y, z)."""
    import numpy as np
    parts = [np.asarray(c, np.int32).reshape(-1, 4) for c in per_rank]
    cells = np.concatenate(parts) if parts else np.zeros((0, 4), np.int32)
    order = np.lexsort((cells[:, 2], cells[:, 1], cells[:, 0]))
    return cells[order]


def words_of(nvox: int) -> int:
    return (nvox + 31) // 32


class OccupancyExchange:
    """The end-of-carve collective on the bit-packed occupancy (1 bit per voxel,
    voxel i -> bit i % 32 of int32 word i // 32, x fastest)."""

    def __init__(self, X: int, Y: int, Z: int, world: int, rank: int, device,
                 mode: str = "allreduce", buffers: int = 2, layout: str = "slab"):
        assert mode in ("allreduce", "allgather") and layout in ("slab", "striped")
        self.X, self.Y, self.Z, self.world, self.rank, self.mode = X, Y, Z, world, rank, mode
        self.layout = layout
        self.z0, self.z1 = slab_of(Z, world, rank)
        plane = X * Y
        if layout == "striped":
            if mode != "allreduce":
                raise ValueError("striped slabs are merged by allreduce only")
            if Z % 8 or plane % 64:
                raise ValueError("striped slabs need Z % 8 == 0 and X*Y % 64 == 0")
        if (plane * self.z0) % 32 or (world > 1 and (plane * (self.z1 - self.z0)) % 32):
            raise ValueError("slab boundaries must fall on 32-voxel words of the packed plane")
        if mode == "allgather" and Z % world:
            raise ValueError("allgather needs equal slabs (world must divide Z)")
        self.off_words = plane * self.z0 // 32
        self.my_words = words_of(plane * (self.z1 - self.z0))
        self.total_words = words_of(plane * Z)
        self.full = [torch.zeros(self.total_words, dtype=torch.int32, device=device)
                     for _ in range(buffers)]
        self.pending = [None] * buffers

    def my_slice(self, b: int) -> torch.Tensor:
        """Where this rank's pack_occupancy output goes inside buffer b."""
        return self.full[b][self.off_words:self.off_words + self.my_words]

    def prepare(self, b: int) -> None:
        """Make buffer b reusable: wait for its previous collective, and for
        allreduce re-zero the words owned by other ranks."""
        self.wait(b)
        if self.mode == "allreduce" and self.world > 1:
            if self.layout == "striped":
                self.full[b].zero_()  # own stripes are rewritten by the pack that follows
            else:
                self.full[b][:self.off_words].zero_()
                self.full[b][self.off_words + self.my_words:].zero_()

    def launch(self, b: int, async_op: bool = True) -> None:
        """Start the collective on buffer b (own slab words already written)."""
        if self.world == 1:
            return
        t = self.full[b]
        if self.mode == "allreduce":
            w = dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)
        else:
            w = dist.all_gather_into_tensor(t, self.my_slice(b), async_op=async_op)
        self.pending[b] = w if async_op else None

    def wait(self, b: int) -> None:
        w = self.pending[b]
        if w is not None:
            w.wait()
            self.pending[b] = None

    def wait_all(self) -> None:
        for b in range(len(self.full)):
            self.wait(b)
