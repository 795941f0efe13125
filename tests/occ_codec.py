"""numpy restatement of the compressed occupancy packet (include/arvx/arvx.h,
arvx_occupancy_compress / arvx_occupancy_expand): the checker for the HIP kernels, and
the codec the gloo tests hand to sharding.OccupancyExchange (test infrastructure only)."""
import ctypes as C

import numpy as np

U64 = np.uint64
ONES = U64(0xFFFFFFFFFFFFFFFF)


def header_words(n):
    nb = (n + 63) // 64
    return 1 + 2 * nb + (nb + 1) // 2


def _bitmap(flags, nb):
    f = np.zeros(nb * 64, np.uint8)
    f[:len(flags)] = flags
    return np.packbits(f, bitorder="little").view(U64)


def compress(words, cap):
    """words: (n,) uint64 -> packet (header + cap,) uint64; slots past the stored mixed
    words are left zero (the device leaves them untouched)."""
    words = np.asarray(words, U64)
    n = len(words)
    nb = (n + 63) // 64
    H = header_words(n)
    ones = words == ONES
    mixed = (words != 0) & ~ones
    out = np.zeros(H + cap, U64)
    out[0] = int(mixed.sum())
    out[1:1 + nb] = _bitmap(ones, nb)
    out[1 + nb:1 + 2 * nb] = _bitmap(mixed, nb)
    per_group = np.add.reduceat(np.pad(mixed, (0, nb * 64 - n)).astype(np.int64),
                                np.arange(0, nb * 64, 64))
    offs = np.concatenate([[0], np.cumsum(per_group)[:-1]]).astype(np.uint32)
    out[1 + 2 * nb:H].view(np.uint32)[:nb] = offs
    mw = words[mixed][:cap]
    out[H:H + len(mw)] = mw
    return out


def expand(packets, world, self_rank, n, cap, full):
    """packets: (world * S,) uint64; full: (world * n,) uint64, updated in place for every
    q != self_rank.  -> True if some packet -- the caller's own included, so that every rank
    arrives at the same verdict -- had outgrown cap (that slab is skipped)."""
    nb = (n + 63) // 64
    H = header_words(n)
    S = H + cap
    overflow = False
    for q in range(world):
        pk = packets[q * S:(q + 1) * S]
        if int(pk[0]) > cap:
            overflow = True
            continue
        if q == self_rank:
            continue
        ones = np.unpackbits(pk[1:1 + nb].view(np.uint8), bitorder="little")[:n].astype(bool)
        mixed = np.unpackbits(pk[1 + nb:1 + 2 * nb].view(np.uint8), bitorder="little")[:n].astype(bool)
        w = np.where(ones, ONES, U64(0))
        w[mixed] = pk[H:H + int(mixed.sum())]
        full[q * n:(q + 1) * n] = w
    return overflow


def expand_striped(packets, world, n, cap, wpg, full):
    """Every rank's packet (the caller's too) -> its 8-plane groups q, q + world, ... of the
    whole plane (wpg words per group).  -> True if some packet had outgrown cap."""
    tmp = np.zeros(world * n, U64)
    over = expand(packets, world, -1, n, cap, tmp)
    H = header_words(n)
    S = H + cap
    for q in range(world):
        if int(packets[q * S]) > cap:
            continue
        loc = tmp[q * n:(q + 1) * n].reshape(-1, wpg)
        for gl in range(loc.shape[0]):
            g = gl * world + q
            full[g * wpg:(g + 1) * wpg] = loc[gl]
    return over


def _arr(ptr, nwords, dtype=U64):
    ct = {U64: C.c_uint64, np.int32: C.c_int32}[dtype]
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(nwords,))


class NumpyCodec:
    """Same calls as capi.Context.occupancy_compress / occupancy_expand, on host pointers."""

    def occupancy_compress(self, words_ptr, n, packet_ptr, cap):
        pk = _arr(packet_ptr, header_words(n) + cap)
        pk[:] = compress(_arr(words_ptr, n), cap)

    def occupancy_expand_striped(self, packets_ptr, world, n, cap, wpg, full_ptr, overflow_ptr):
        S = header_words(n) + cap
        if expand_striped(_arr(packets_ptr, world * S), world, n, cap, wpg,
                          _arr(full_ptr, world * n)):
            _arr(overflow_ptr, 1, np.int32)[0] = 1

    def occupancy_expand(self, packets_ptr, world, self_rank, n, cap, full_ptr, overflow_ptr):
        S = header_words(n) + cap
        if expand(_arr(packets_ptr, world * S), world, self_rank, n, cap,
                  _arr(full_ptr, world * n)):
            _arr(overflow_ptr, 1, np.int32)[0] = 1


class FusedNumpyCodec(NumpyCodec):
    """A codec that also has the fused calls of capi.Context (occupancy_pack_compress,
    occupancy_expand_striped_others): it stands for a context whose current state the test sets
    with `carve(words)` -- the rank's occupancy words in local order -- so that the host logic of
    sharding.OccupancyExchange's fused path runs over gloo."""

    def __init__(self, world, rank, wpg):
        self.world, self.rank, self.wpg = world, rank, wpg  # wpg = 0: contiguous slabs
        self.state = None
        self.carves = 0

    def carve(self, words):
        self.state = np.array(words, U64)
        self.carves += 1

    def occupancy_pack_compress(self, packet_ptr, cap, full_ptr):
        n = len(self.state)
        _arr(packet_ptr, header_words(n) + cap)[:] = compress(self.state, cap)
        if full_ptr:
            full = _arr(full_ptr, self.world * n)
            if self.wpg:
                full.reshape(-1, self.world, self.wpg)[:, self.rank, :] = self.state.reshape(-1, self.wpg)
            else:
                full[self.rank * n:(self.rank + 1) * n] = self.state

    def occupancy_expand_striped_others(self, packets_ptr, world, self_rank, n, cap, wpg, full_ptr,
                                        overflow_ptr):
        S = header_words(n) + cap
        packets = _arr(packets_ptr, world * S)
        full = _arr(full_ptr, world * n)
        own = full.reshape(-1, world, wpg)[:, self_rank, :].copy()
        over = expand_striped(packets, world, n, cap, wpg, full)
        full.reshape(-1, world, wpg)[:, self_rank, :] = own  # the caller's packet is skipped
        if over:
            _arr(overflow_ptr, 1, np.int32)[0] = 1
