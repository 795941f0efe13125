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
  compressed allgather of PACKETS instead of words: most 64-bit words of a packed slab
             are all-zero or all-one, so a slab travels as two bitmaps plus its mixed
             words (arvx_occupancy_compress / _expand, include/arvx/arvx.h).  The packet
             size is fixed per exchange by `cap` (room for mixed words); it starts at the
             worst case and retune() shrinks it to what the last exchange needed + 25 %.
             A slab that outgrows cap raises a flag and wait() redoes that exchange as a
             plain allgather.

Two ways to cut the grid:

  slab     rank r owns the contiguous planes slab_of(Z, world, r)
  striped  with the planes cut into groups of 8, rank r owns groups r, r+world, ...
           Surface voxels (where the per-voxel work is) cluster in z, so contiguous
           slabs leave the ranks that own empty space idle; stripes balance the load
           (1024^3 over 8 ranks: the slowest slab carves in 0.110 ms, every stripe set
           in 0.082 ms).  A rank's words are scattered over the plane: the allreduce
           form applies, and the compressed one (a rank packs and compresses its planes
           in local order; the expand puts every rank's words in place, its own too).

Host logic only: runs on CPU tensors with gloo (tests) and on GPU tensors with
nccl == RCCL (bench.py).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def grid_for(world: int, base: int) -> Tuple[int, int, int]:
    """Grid holding ~world * base^3 voxels: X = Y = the multiple of 32 closest to
    base * world^(1/3) (rows of whole 32-bit words of the packed plane: the kernels' widest
    paths), Z the multiple of 8*world that keeps the volume, so that all slabs are equal and
    start on a 32-voxel boundary of the packed plane."""
    if world == 1:
        return base, base, base
    n = max(32, int(round(base * world ** (1.0 / 3.0) / 32.0)) * 32)
    z = max(1, int(round(world * float(base) ** 3 / (n * n) / (8.0 * world)))) * 8 * world
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


def merge_closure(per_rank, X: int, Y: int, slabs):
    """The filled voxels of the slab contexts (Context.closure: (index, rgba) per rank, indices
    numbered over the rank's OWN planes) -> the whole grid's list, ascending flat index: slabs
    are ordered by z, so this is a concatenation with the index moved to the grid's numbering.
    slabs: (z0, z1) per rank."""
    import numpy as np
    idx = [np.asarray(i, np.int64) + X * Y * z0 for (i, _), (z0, _z1) in zip(per_rank, slabs)]
    rgba = [np.asarray(c, np.float32).reshape(-1, 4) for _, c in per_rank]
    return np.concatenate(idx), np.concatenate(rgba)


def merge_surface(per_rank, X: int, Y: int, slabs):
    """The coloured voxels of the slab contexts (Context.surface: (index, rgb)) -> the whole
    grid's list, as merge_closure."""
    import numpy as np
    idx = [np.asarray(i, np.int64) + X * Y * z0 for (i, _), (z0, _z1) in zip(per_rank, slabs)]
    rgb = [np.asarray(c, np.float32).reshape(-1, 3) for _, c in per_rank]
    return np.concatenate(idx), np.concatenate(rgb)


def triangles_per_cube_index():
    """Triangles Bourke's table emits per cube index (include/arvx/mc_triangles.inc: three hex
    digits per triangle)."""
    import os
    import re
    import numpy as np
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include",
                       "arvx", "mc_triangles.inc")
    rows = re.findall(r'"([0-9a-f]*)"', open(inc).read().split("*/", 1)[1])
    assert len(rows) == 256
    return np.array([len(r) // 3 for r in rows], np.int64)


def merge_mesh(per_rank):
    """The meshes of the slab contexts -- per rank (cells (n,4), verts (3T,3), face_rgb (T,3)):
    Context.mc_cells and Context.mc_mesh of the same state -- -> the mesh marchingCubes() builds
    of the whole grid: cells in the reference's visiting order (x outermost, then y, then z;
    slabs own disjoint z ranges), every cell's triangles kept together and in their order."""
    import numpy as np
    ntri = triangles_per_cube_index()
    cells = np.concatenate([np.asarray(c, np.int32).reshape(-1, 4) for c, _, _ in per_rank])
    verts = np.concatenate([np.asarray(v, np.float32).reshape(-1, 9) for _, v, _ in per_rank])
    rgb = np.concatenate([np.asarray(r, np.uint32).reshape(-1, 3) for _, _, r in per_rank])
    count = ntri[cells[:, 3] & 255]
    start = np.concatenate([[0], np.cumsum(count)[:-1]])  # first triangle of a cell, slab order
    assert int(count.sum()) == len(verts) == len(rgb)
    order = np.lexsort((cells[:, 2], cells[:, 1], cells[:, 0]))
    take = np.concatenate([np.arange(start[c], start[c] + count[c]) for c in order]) \
        if len(order) else np.zeros(0, np.int64)
    return cells[order], verts[take].reshape(-1, 3), rgb[take]


def words_of(nvox: int) -> int:
    return (nvox + 31) // 32


class OccupancyExchange:
    """The end-of-carve collective on the bit-packed occupancy (1 bit per voxel,
    voxel i -> bit i % 32 of int32 word i // 32, x fastest).

    mode "compressed" needs `codec`: an object with occupancy_compress /
    occupancy_expand taking device pointers (capi.Context has them).  The codec must run
    on torch's current stream (Context.set_stream(torch.cuda.current_stream().cuda_stream)):
    the collective is ordered against that stream, and so must the kernels around it be."""

    def __init__(self, X: int, Y: int, Z: int, world: int, rank: int, device,
                 mode: str = "allreduce", buffers: int = 2, layout: str = "slab", codec=None,
                 lazy_expand: bool = False):
        """lazy_expand (mode "compressed"): the exchange's product stays what travelled -- every
        rank's packet, each with its own index (bitmaps + per-group offsets: a word is found
        without expanding anything, include/arvx/arvx.h) -- and the plain merged plane is built
        only when somebody asks for it (merged_plane(b)): the expansion writes N/8 bytes per job that
        no stage of the sharded pipeline reads (colour pass, closure and mesh run on slabs)."""
        assert mode in ("allreduce", "allgather", "compressed") and layout in ("slab", "striped")
        self.lazy_expand = bool(lazy_expand) and mode == "compressed"
        self.X, self.Y, self.Z, self.world, self.rank, self.mode = X, Y, Z, world, rank, mode
        self.layout = layout
        self.z0, self.z1 = slab_of(Z, world, rank)
        plane = X * Y
        if layout == "striped":
            if mode == "allgather":
                raise ValueError("striped slabs are merged by allreduce or compressed packets")
            if Z % 8 or plane % 64:
                raise ValueError("striped slabs need Z % 8 == 0 and X*Y % 64 == 0")
            if mode == "compressed" and (Z % (8 * world) or plane % 16):
                raise ValueError("compressed striped slabs need Z % (8*world) == 0, X*Y % 16 == 0")
        if layout == "slab" and ((plane * self.z0) % 32 or
                                 (world > 1 and (plane * (self.z1 - self.z0)) % 32)):
            raise ValueError("slab boundaries must fall on 32-voxel words of the packed plane")
        if mode != "allreduce" and Z % world:
            raise ValueError(f"{mode} needs equal slabs (world must divide Z)")
        self.off_words = plane * self.z0 // 32
        self.my_words = words_of(plane * (self.z1 - self.z0))
        self.total_words = words_of(plane * Z)
        self.full = [torch.zeros(self.total_words, dtype=torch.int32, device=device)
                     for _ in range(buffers)]
        self.pending = [None] * buffers
        if mode == "compressed":
            if codec is None:
                raise ValueError("compressed exchange needs a codec (a capi.Context)")
            if (plane * (self.z1 - self.z0)) % 64:
                raise ValueError("compressed exchange needs slabs of whole 64-bit words")
            self.codec = codec
            self.n64 = self.my_words // 2
            # striped: the rank's planes packed in local order (Context.pack_occupancy), and
            # how many 64-bit words an 8-plane group has
            self.wpg = plane * 8 // 64 if layout == "striped" else 0
            self.local = ([torch.zeros(self.n64, dtype=torch.int64, device=device)
                           for _ in range(buffers)] if layout == "striped" else None)
            self.header = packet_header_words(self.n64)
            self.cap_max = self.n64  # every word mixed: cannot overflow
            self.cap = self.cap_max
            smax = self.header + self.cap_max
            self.packet = [torch.zeros(smax, dtype=torch.int64, device=device)
                           for _ in range(buffers)]
            self.gathered = [torch.zeros(world * smax, dtype=torch.int64, device=device)
                             for _ in range(buffers)]
            self.cap_of = [self.cap] * buffers  # cap the exchange in flight on buffer b used
            self.unexpanded = [False] * buffers
            # the packet straight from the context's state (Context.occupancy_pack_compress: no
            # plain words in between, own words written to their place in `full`): where the
            # library can (X % 32 == 0, whole 64-bit words per plane)
            self.can_fuse = X % 32 == 0 and plane % 64 == 0 and hasattr(codec, "occupancy_pack_compress")
            self.fused = [False] * buffers
            self.overflow = torch.zeros(1, dtype=torch.int32, device=device)  # sticky
            self.fallbacks = 0

    def my_slice(self, b: int) -> torch.Tensor:
        """Where this rank's pack_occupancy output goes inside buffer b."""
        return self.full[b][self.off_words:self.off_words + self.my_words]

    def prepare(self, b: int, verify: bool = True) -> None:
        """Make buffer b reusable: wait for its previous collective, and for
        allreduce re-zero the words owned by other ranks."""
        self.wait(b, verify)
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
        elif self.mode == "allgather":
            w = dist.all_gather_into_tensor(t, self.my_slice(b), async_op=async_op)
        else:
            if self.fused[b]:  # pack() has written the packet
                cap = self.cap_of[b]
            else:
                cap = self.cap
                src = self.local[b] if self.layout == "striped" else self.my_slice(b)
                self.codec.occupancy_compress(src.data_ptr(), self.n64, self.packet[b].data_ptr(), cap)
            S = self.header + cap
            w = dist.all_gather_into_tensor(self.gathered[b][:self.world * S],
                                            self.packet[b][:S], async_op=async_op)
            self.cap_of[b] = cap
            self.unexpanded[b] = True
            if not async_op and not self.lazy_expand:
                self._expand(b, True)
        self.pending[b] = w if async_op else None

    def pack(self, ctx, b: int) -> None:
        """ctx's occupancy into buffer b, in the form the collective ships.  compressed: the packet
        straight from the state where the library can (Context.occupancy_pack_compress: two
        launches instead of pack + compress's four, the rank's own words stored in full[b] on
        the way, so that the expansion leaves its own packet out); launch(b) then only starts the
        all-gather."""
        if self.mode == "compressed" and self.can_fuse:
            ctx.occupancy_pack_compress(self.packet[b].data_ptr(), self.cap, self.full[b].data_ptr())
            self.fused[b] = True
            self.cap_of[b] = self.cap
        elif self.mode == "compressed" and self.layout == "striped":
            ctx.pack_occupancy(self.local[b].data_ptr())
        elif self.mode == "compressed":
            ctx.pack_occupancy(self.my_slice(b).data_ptr())
        else:
            ctx.pack_occupancy_global(self.full[b].data_ptr())

    def _expand(self, b: int, verify: bool) -> None:
        if self.layout == "striped" and self.fused[b]:
            self.codec.occupancy_expand_striped_others(self.gathered[b].data_ptr(), self.world, self.rank,
                                                       self.n64, self.cap_of[b], self.wpg,
                                                       self.full[b].data_ptr(), self.overflow.data_ptr())
        elif self.layout == "striped":
            self.codec.occupancy_expand_striped(self.gathered[b].data_ptr(), self.world, self.n64,
                                                self.cap_of[b], self.wpg, self.full[b].data_ptr(),
                                                self.overflow.data_ptr())
        else:
            self.codec.occupancy_expand(self.gathered[b].data_ptr(), self.world, self.rank,
                                        self.n64, self.cap_of[b], self.full[b].data_ptr(),
                                        self.overflow.data_ptr())
        self.unexpanded[b] = False
        was_fused, self.fused[b] = self.fused[b], False
        if verify and self.overflowed():
            # some slab had more mixed words than cap: redo this exchange in plain words.
            # Every rank sees the same packets and every expansion looks at all of them -- the
            # caller's own included, which it otherwise skips --, so every rank takes this branch.
            self.overflow.zero_()
            self.fallbacks += 1
            self.cap = self.cap_max
            if self.layout == "striped":
                # worst-case packets cannot overflow: the same exchange again at full size
                S = self.header + self.cap_max
                if was_fused:
                    # THIS job's own words are in full[b] (pack() stored them there); the
                    # context may have carved another job since, so its state is not asked again
                    own = self.full[b].view(torch.int64).view(-1, self.world, self.wpg)[:, self.rank, :]
                    self.local[b].copy_(own.reshape(-1))
                self.codec.occupancy_compress(self.local[b].data_ptr(), self.n64,
                                              self.packet[b].data_ptr(), self.cap_max)
                dist.all_gather_into_tensor(self.gathered[b][:self.world * S], self.packet[b][:S])
                self.codec.occupancy_expand_striped(self.gathered[b].data_ptr(), self.world,
                                                    self.n64, self.cap_max, self.wpg,
                                                    self.full[b].data_ptr(),
                                                    self.overflow.data_ptr())
            else:
                dist.all_gather_into_tensor(self.full[b], self.my_slice(b))

    def overflowed(self) -> bool:
        """Has any expand since the last reset met a packet that outgrew its cap?
        (Reads a device flag: synchronises.)"""
        if self.mode != "compressed":
            return False
        if self.lazy_expand and self.world > 1:  # (no expansion has looked at the packets yet)
            if any(self.unexpanded[b] and self.pending[b] is None and self._packets_overflow(b)
                   for b in range(len(self.full))):
                return True
        return bool(self.overflow.item())

    def wait(self, b: int, verify: bool = True) -> None:
        """Buffer b's exchange is complete (in stream order).  compressed: verify=True reads
        the overflow flag (a host sync) and repairs an overflowed exchange; a pipelined
        caller passes False and checks overflowed() where it synchronises anyway."""
        w = self.pending[b]
        if w is not None:
            w.wait()
            self.pending[b] = None
        if self.mode == "compressed" and self.unexpanded[b] and not self.lazy_expand:
            self._expand(b, verify)
        elif self.mode == "compressed" and self.unexpanded[b] and verify and self._packets_overflow(b):
            self._expand(b, True)  # (an overflowed exchange is repaired at once: a collective, all ranks)

    def _packets_overflow(self, b: int) -> bool:
        """lazy_expand: did a packet of buffer b's exchange outgrow its cap?  (Reads the gathered
        headers: synchronises.  Every rank holds the same packets and arrives at the same answer.)"""
        S = self.header + self.cap_of[b]
        counts = self.gathered[b][:self.world * S].view(self.world, S)[:, 0]
        return bool((counts > self.cap_of[b]).any().item())

    def merged_plane(self, b: int, verify: bool = True) -> "torch.Tensor":
        """The whole grid's packed occupancy of buffer b's exchange (int32 words), expanded now if
        the exchange kept it as packets (lazy_expand)."""
        self.wait(b, verify)
        if self.mode == "compressed" and self.unexpanded[b]:
            self._expand(b, verify)
        return self.full[b]

    def wait_all(self, verify: bool = True) -> None:
        for b in range(len(self.full)):
            self.wait(b, verify)

    def retune(self, b: int = 0) -> int:
        """compressed: size the packets for what buffer b's last exchange needed (+25 %).
        Call with no exchange in flight; reads the packets' counts (synchronises).  All
        ranks hold the same packets, so all arrive at the same cap."""
        assert self.mode == "compressed" and all(p is None for p in self.pending)
        if self.world == 1:
            return self.cap
        S = self.header + self.cap_of[b]
        counts = self.gathered[b][:self.world * S].view(self.world, S)[:, 0]
        need = int(counts.max().item())
        self.cap = min(self.cap_max, need + need // 4 + 16)
        return self.cap


def packet_header_words(n64: int) -> int:
    """64-bit words before the mixed words of a packet (include/arvx/arvx.h)."""
    nb = (n64 + 63) // 64
    return 1 + 2 * nb + (nb + 1) // 2
