// fast_carve_kernels.h -- the reference's greedy carve, fastCarve()
// (src/VoxelCarving.cpp:74-167), as a data-parallel flood fill on gfx950.
//
// The reference pops voxels from a queue seeded with (0,0,0): an unvisited voxel
// is marked visited (== seen, src/Model.h:154-160) and tested against the views
// in order; the first view that sees background there carves it, and a carved
// voxel pushes its unvisited 6-neighbours.  The fixed point is order free:
//   carvable(v) = some view projects v inside the image onto a background pixel
//   open(v)     = carvable(v) and not seen before the call
//   E           = 6-connected component of `open` that contains (0,0,0)
//   carved = E ;  seen += E + (6-neighbours of E) + (0,0,0)
// `carvable` is exactly what the dense carve kernel clears on a fresh plane, so
// it is computed by that kernel (exact projection, culling and all).  The
// component is grown on bit planes, 64 voxels of a row per 64-bit word: one
// thread owns one word, fills it along x with a Kogge-Stone occluded fill, and
// exchanges with its four row neighbours through LDS until its 64x16x16 tile
// is stable; launches repeat until no tile changes.
//
// A front that moves one tile per launch would need as many launches as the grid
// is wide in tiles.  Most of `open` is empty space far from the object, though:
// tiles that are open in every voxel.  Two such tiles that share a face are
// connected, so the component is first grown over WHOLE tiles (one bit per tile,
// a row of tiles per 64-bit word, one workgroup, LDS only) and every tile reached
// that way is seeded completely; the word-level launches then only have to enter
// the partly open tiles around the object.
#pragma once

#include "arvx_device.h"

namespace arvx {

struct FloodParams {
    int X, Y, Z;
    int XW;  // 64-bit words per row
    unsigned long long *open;
    unsigned long long *reach;
    int *changed;
    uint8_t *dirty_cur;   // per tile: some word of the tile or of a face neighbour changed
    uint8_t *dirty_next;  //           in the previous launch (flood_step_kernel)
    int fresh;  // the model is fresh (all occupied, none seen): its plane is not read, and
                // flood_apply writes every voxel instead of the changed ones
};

// open = carvable & !seen0, one wave per (row, word)
__global__ __launch_bounds__(256) void flood_pack_open_kernel(const uint8_t *__restrict__ carved_tmp,
                                                              const uint8_t *__restrict__ state,
                                                              const FloodParams p) {
    const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nrows = (size_t)p.Y * p.Z;
    if (wv >= nrows * p.XW) return;
    const size_t row = wv / p.XW;
    const int xw = (int)(wv % p.XW);
    const int x = xw * 64 + (threadIdx.x & 63);
    bool o = false;
    if (x < p.X) {
        const size_t i = row * p.X + x;
        o = !(carved_tmp[i] & 1u) && (p.fresh || !(state[i] & 2u));
    }
    const unsigned long long b = __ballot(o);
    if ((threadIdx.x & 63) == 0) {
        p.open[wv] = b;
        p.reach[wv] = (wv == 0) ? (b & 1ull) : 0ull;  // seed: voxel (0,0,0) if it is open
    }
}

// The same for X % 8 == 0: one thread packs 8 voxels into one BYTE of the bit plane
// (byte j of a little-endian 64-bit word = voxels 8j..8j+7), 8-byte loads, byte store.
// Rows are padded to whole words; the padding bytes are written as zero.  `reach`
// is cleared by the host; the thread of voxel (0,0,0) plants the seed.
__global__ __launch_bounds__(256) void flood_pack_open8_kernel(const uint8_t *__restrict__ carved_tmp,
                                                               const uint8_t *__restrict__ state,
                                                               const FloodParams p) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int rowBytes = p.XW * 8;
    const size_t nrows = (size_t)p.Y * p.Z;
    if (t >= nrows * rowBytes) return;
    const size_t row = t / rowBytes;
    const int g = (int)(t % rowBytes);
    uint8_t b = 0;
    if (g * 8 < p.X) {
        const size_t i = row * p.X + (size_t)g * 8;
        const unsigned long long c = *(const unsigned long long *)(carved_tmp + i);
        const unsigned long long s =
            p.fresh ? 0x0101010101010101ull : *(const unsigned long long *)(state + i);
        // open = carved on the fresh plane (bit0 clear) and not seen (bit1 clear)
        const unsigned long long m = ~c & ~(s >> 1) & 0x0101010101010101ull;
        b = (uint8_t)((m * 0x0102040810204080ull) >> 56);  // byte j's bit 0 -> bit j
    }
    ((uint8_t *)p.open)[t] = b;
    if (t == 0) ((uint8_t *)p.reach)[0] = b & 1u;
}

// The same for X % 32 == 0: 32 voxels per thread, 16-byte loads, one 32-bit half word out.
__global__ __launch_bounds__(256) void flood_pack_open32_kernel(const uint8_t *__restrict__ carved_tmp,
                                                                const uint8_t *__restrict__ state,
                                                                const FloodParams p) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int rowHalves = p.XW * 2;
    const size_t nrows = (size_t)p.Y * p.Z;
    if (t >= nrows * rowHalves) return;
    const size_t row = t / rowHalves;
    const int h = (int)(t % rowHalves);
    uint32_t w = 0;
    if (h * 32 < p.X) {
        const size_t i = row * p.X + (size_t)h * 32;
        const ulonglong2 *c = (const ulonglong2 *)(carved_tmp + i);
        const ulonglong2 c0 = c[0], c1 = c[1];
        ulonglong2 s0, s1;
        s0.x = s0.y = s1.x = s1.y = 0x0101010101010101ull;
        if (!p.fresh) {
            const ulonglong2 *sp = (const ulonglong2 *)(state + i);
            s0 = sp[0];
            s1 = sp[1];
        }
        const unsigned long long one = 0x0101010101010101ull, mul = 0x0102040810204080ull;
        auto pack = [&](unsigned long long cc, unsigned long long ss) -> uint32_t {
            return (uint32_t)(((~cc & ~(ss >> 1) & one) * mul) >> 56);
        };
        w = pack(c0.x, s0.x) | (pack(c0.y, s0.y) << 8) | (pack(c1.x, s1.x) << 16) |
            (pack(c1.y, s1.y) << 24);
    }
    ((uint32_t *)p.open)[t] = w;
    if (t == 0) ((uint8_t *)p.reach)[0] = (uint8_t)(w & 1u);
}

__device__ __forceinline__ unsigned long long fill_row(unsigned long long seed,
                                                       unsigned long long open) {
    // Kogge-Stone occluded fill, both directions, within one 64-voxel word
    unsigned long long g = seed, pr = open;
    g |= pr & (g << 1);  pr &= pr << 1;
    g |= pr & (g << 2);  pr &= pr << 2;
    g |= pr & (g << 4);  pr &= pr << 4;
    g |= pr & (g << 8);  pr &= pr << 8;
    g |= pr & (g << 16); pr &= pr << 16;
    g |= pr & (g << 32);
    unsigned long long h = seed;
    pr = open;
    h |= pr & (h >> 1);  pr &= pr >> 1;
    h |= pr & (h >> 2);  pr &= pr >> 2;
    h |= pr & (h >> 4);  pr &= pr >> 4;
    h |= pr & (h >> 8);  pr &= pr >> 8;
    h |= pr & (h >> 16); pr &= pr >> 16;
    h |= pr & (h >> 32);
    return g | h;
}

// ---- whole-tile pre-pass ---------------------------------------------------------------

constexpr int kFloodMaxTileRows = 4096;  // rows of tiles (tilesY * tilesZ): 2 words each = 64 KB of LDS

// full[(bz * tilesY + by)] bit xw = every voxel of tile (xw, by, bz) that lies inside
// the grid is open.  One workgroup per tile, thread = one 64-voxel word.  Also sets the
// tile's first wake flag: only a PARTLY open tile can grow on its own account -- a full
// one is either seeded completely by the pre-pass or woken later by a neighbour, one
// without open voxels never changes.
__global__ __launch_bounds__(256) void flood_tile_full_kernel(const FloodParams p,
                                                              unsigned long long *__restrict__ full,
                                                              uint8_t *__restrict__ wake) {
    // one workgroup per row of tiles (by, bz); consecutive threads read consecutive
    // words of a voxel row (xw fastest), several voxel rows per step.  XW <= 64.
    __shared__ unsigned s_notfull[2], s_any[2];  // bit xw, as two 32-bit halves
    const int tilesY = (p.Y + 15) >> 4;
    const int by = blockIdx.x % tilesY, bz = blockIdx.x / tilesY;
    if (threadIdx.x < 2) s_notfull[threadIdx.x] = s_any[threadIdx.x] = 0u;
    __syncthreads();
    const int per = 256 / p.XW;  // voxel rows per step
    const int xw = threadIdx.x % p.XW, sub = threadIdx.x / p.XW;
    bool notfull = false, any = false;
    if (sub < per) {
        const int nbits = min(64, p.X - xw * 64);
        const unsigned long long want = (nbits == 64) ? ~0ull : ((1ull << nbits) - 1ull);
        for (int r = sub; r < 256; r += per) {  // r = tz * 16 + ty inside the tile row
            const int y = by * 16 + (r & 15), z = bz * 16 + (r >> 4);
            if (y >= p.Y || z >= p.Z) continue;
            const unsigned long long o = p.open[((size_t)z * p.Y + y) * p.XW + xw];
            notfull = notfull || (o != want);
            any = any || (o != 0ull);
        }
        if (notfull) atomicOr(&s_notfull[xw >> 5], 1u << (xw & 31));
        if (any) atomicOr(&s_any[xw >> 5], 1u << (xw & 31));
    }
    __syncthreads();
    const unsigned long long nf = s_notfull[0] | ((unsigned long long)s_notfull[1] << 32);
    const unsigned long long an = s_any[0] | ((unsigned long long)s_any[1] << 32);
    const unsigned long long valid = (p.XW == 64) ? ~0ull : ((1ull << p.XW) - 1ull);
    if (threadIdx.x == 0) full[blockIdx.x] = ~nf & valid;
    if (threadIdx.x < p.XW)
        wake[(size_t)blockIdx.x * p.XW + threadIdx.x] =
            (((an >> threadIdx.x) & 1ull) && ((nf >> threadIdx.x) & 1ull)) ? 1 : 0;
}

// Component of the tile (0,0,0) among the full tiles, 6-connected.  ONE workgroup;
// rows of tiles live in LDS as 64-bit words (bit = tile along x).  reached[] is
// written for every row.  Requires XW <= 64 and tilesY*tilesZ <= kFloodMaxTileRows.
__global__ __launch_bounds__(1024) void flood_tile_fill_kernel(
    const unsigned long long *__restrict__ full, unsigned long long *__restrict__ reached,
    int tilesY, int tilesZ) {
    extern __shared__ unsigned long long lds[];  // [rows] full, then [rows] reached
    const int rows = tilesY * tilesZ;
    unsigned long long *f = lds, *r = lds + rows;
    for (int i = threadIdx.x; i < rows; i += 1024) {
        f[i] = full[i];
        r[i] = 0ull;
    }
    __syncthreads();
    if (threadIdx.x == 0) r[0] = f[0] & 1ull;  // seed: the tile of voxel (0,0,0), if full
    __syncthreads();
    // every round that reports a change adds at least one tile, so this ends.  Inside a
    // round the rows are relaxed a few times without a barrier: a racing reader sees the
    // old or the new word of a neighbour, both are sound (the set only grows), and most
    // of the front moves on without waiting for the whole workgroup.
    volatile unsigned long long *rv = r;
    for (;;) {
        int ch = 0;
        for (int rep = 0; rep < 4; ++rep)
            for (int i = threadIdx.x; i < rows; i += 1024) {
                const unsigned long long mine = rv[i];
                if (mine == f[i]) continue;  // every full tile of this row is in already
                const int by = i % tilesY, bz = i / tilesY;
                unsigned long long in = mine;
                if (by > 0) in |= rv[i - 1];
                if (by + 1 < tilesY) in |= rv[i + 1];
                if (bz > 0) in |= rv[i - tilesY];
                if (bz + 1 < tilesZ) in |= rv[i + tilesY];
                if (!(f[i] & in & ~mine) && !((mine << 1 | mine >> 1) & f[i] & ~mine))
                    continue;  // no new seed from the four neighbours, none along x
                const unsigned long long rn = fill_row(f[i] & in, f[i]);
                if (rn != mine) {
                    rv[i] = rn;
                    ch = 1;
                }
            }
        if (!__syncthreads_or(ch)) break;
    }
    for (int i = threadIdx.x; i < rows; i += 1024) reached[i] = r[i];
}

// reach = open in every tile the pre-pass reached; one workgroup per row of tiles,
// threads along the words of a voxel row
__global__ __launch_bounds__(256) void flood_tile_seed_kernel(
    const FloodParams p, const unsigned long long *__restrict__ reached) {
    const unsigned long long rbits = reached[blockIdx.x];
    if (!rbits) return;
    const int tilesY = (p.Y + 15) >> 4;
    const int by = blockIdx.x % tilesY, bz = blockIdx.x / tilesY;
    const int per = 256 / p.XW;
    const int xw = threadIdx.x % p.XW, sub = threadIdx.x / p.XW;
    if (sub >= per || !((rbits >> xw) & 1ull)) return;
    for (int r = sub; r < 256; r += per) {
        const int y = by * 16 + (r & 15), z = bz * 16 + (r >> 4);
        if (y >= p.Y || z >= p.Z) continue;
        const size_t w = ((size_t)z * p.Y + y) * p.XW + xw;
        p.reach[w] = p.open[w];
    }
}

__global__ __launch_bounds__(256) void flood_step_kernel(const FloodParams p) {
    __shared__ unsigned long long tile[18][18];  // [z][y] with a one-row halo
    const int ty = threadIdx.x & 15, tz = threadIdx.x >> 4;
    const int tilesY = (p.Y + 15) >> 4;
    const int xw = blockIdx.x % p.XW;
    const int by = (blockIdx.x / p.XW) % tilesY;
    const int bz = blockIdx.x / (p.XW * tilesY);
    // nothing that this tile reads changed in the previous launch: it is stable
    if (!p.dirty_cur[blockIdx.x]) return;
    __syncthreads();  // every thread has read the flag
    if (threadIdx.x == 0) p.dirty_cur[blockIdx.x] = 0;  // this buffer is "next" two launches on
    const int y = by * 16 + ty, z = bz * 16 + tz;
    const bool ok = (y < p.Y) && (z < p.Z);
    auto word = [&](const unsigned long long *a, int yy, int zz, int xx) -> unsigned long long {
        if (yy < 0 || yy >= p.Y || zz < 0 || zz >= p.Z || xx < 0 || xx >= p.XW) return 0ull;
        return a[((size_t)zz * p.Y + yy) * p.XW + xx];
    };
    const unsigned long long o = ok ? word(p.open, y, z, xw) : 0ull;
    const unsigned long long r0 = ok ? word(p.reach, y, z, xw) : 0ull;
    unsigned long long r = r0;
    // x neighbours: bit 63 of the word to the left feeds bit 0, bit 0 of the right feeds bit 63
    const unsigned long long side =
        ok ? ((word(p.reach, y, z, xw - 1) >> 63) | (word(p.reach, y, z, xw + 1) << 63)) : 0ull;
    // halo rows of the neighbouring tiles (fixed during this launch)
    tile[tz + 1][ty + 1] = r;
    if (ty == 0) tile[tz + 1][0] = word(p.reach, y - 1, z, xw);
    if (ty == 15) tile[tz + 1][17] = word(p.reach, y + 1, z, xw);
    if (tz == 0) tile[0][ty + 1] = word(p.reach, y, z - 1, xw);
    if (tz == 15) tile[17][ty + 1] = word(p.reach, y, z + 1, xw);
    __syncthreads();
    for (int it = 0; it < 64; ++it) {  // bounded; 16+16 hops cross the tile
        const unsigned long long in = r | side | tile[tz + 1][ty] | tile[tz + 1][ty + 2] |
                                      tile[tz][ty + 1] | tile[tz + 2][ty + 1];
        const unsigned long long rn = fill_row(o & in, o);
        const int ch = (rn != r);
        r = rn;
        __syncthreads();
        tile[tz + 1][ty + 1] = r;
        if (!__syncthreads_or(ch)) break;
    }
    const bool mine = ok && r != r0;
    if (mine) p.reach[((size_t)z * p.Y + y) * p.XW + xw] = r;
    if (__syncthreads_or(mine) && threadIdx.x < 7) {
        // wake this tile and its face neighbours for the next launch
        const int tilesZ = (p.Z + 15) >> 4;
        const int dx[7] = {0, -1, 1, 0, 0, 0, 0}, dy[7] = {0, 0, 0, -1, 1, 0, 0},
                  dz[7] = {0, 0, 0, 0, 0, -1, 1};
        const int nx = xw + dx[threadIdx.x], ny = by + dy[threadIdx.x], nz = bz + dz[threadIdx.x];
        if (nx >= 0 && nx < p.XW && ny >= 0 && ny < tilesY && nz >= 0 && nz < tilesZ)
            p.dirty_next[((size_t)nz * tilesY + ny) * p.XW + nx] = 1;
        if (threadIdx.x == 0) *p.changed = 1;
    }
}

// occ &= !E ; seen |= E | N6(E) | origin      (state plane, one byte per voxel)
__global__ __launch_bounds__(256) void flood_apply_kernel(uint8_t *__restrict__ state,
                                                          const FloodParams p) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)p.X * p.Y * p.Z;
    if (i >= n) return;
    const int x = (int)(i % p.X);
    const size_t t = i / p.X;
    const int y = (int)(t % p.Y), z = (int)(t / p.Y);
    auto bit = [&](int xx, int yy, int zz) -> bool {
        if (xx < 0 || xx >= p.X || yy < 0 || yy >= p.Y || zz < 0 || zz >= p.Z) return false;
        return (p.reach[((size_t)zz * p.Y + yy) * p.XW + (xx >> 6)] >> (xx & 63)) & 1ull;
    };
    uint8_t s = p.fresh ? (uint8_t)1 : state[i];
    if (bit(x, y, z)) {
        s = (uint8_t)((s & ~1u) | 2u);  // carved (src/VoxelCarving.cpp:125) and visited (:108)
    } else if (!(s & 2u)) {
        const bool nb = bit(x - 1, y, z) || bit(x + 1, y, z) || bit(x, y - 1, z) ||
                        bit(x, y + 1, z) || bit(x, y, z - 1) || bit(x, y, z + 1);
        if (nb || i == 0) s |= 2u;  // pushed by a carved neighbour (:132-163) or the seed (:100)
    }
    state[i] = s;
}

// The same for X % 8 == 0 (T = uint8_t: one thread owns the 8 voxels of one byte of
// the bit plane, 8-byte state access) and X % 16 == 0 (T = uint16_t: 16 voxels, 16-byte
// state access).
template <typename T>
__global__ __launch_bounds__(256) void flood_apply_wide_kernel(uint8_t *__restrict__ state,
                                                               const FloodParams p) {
    constexpr int kV = 8 * (int)sizeof(T);  // voxels per thread
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int rowUnits = p.XW * 64 / kV, gmax = p.X / kV;
    const size_t nrows = (size_t)p.Y * p.Z;
    if (t >= nrows * gmax) return;
    const size_t row = t / gmax;
    const int g = (int)(t % gmax);
    const int y = (int)(row % p.Y), z = (int)(row / p.Y);
    const T *rb = (const T *)p.reach + row * rowUnits + g;
    const unsigned top = kV - 1, mask = (kV == 8) ? 0xFFu : 0xFFFFu;
    const unsigned e = rb[0];
    unsigned nb = ((e << 1) | (e >> 1)) & mask;
    if (g > 0) nb |= (unsigned)rb[-1] >> top;
    if (g + 1 < gmax) nb |= ((unsigned)rb[1] & 1u) << top;
    if (y > 0) nb |= rb[-(ptrdiff_t)rowUnits];
    if (y + 1 < p.Y) nb |= rb[rowUnits];
    if (z > 0) nb |= rb[-(ptrdiff_t)rowUnits * p.Y];
    if (z + 1 < p.Z) nb |= rb[(ptrdiff_t)rowUnits * p.Y];
    if (t == 0) nb |= 1u;  // the seed is visited whatever happens (:100)
    if ((e | nb) == 0u && !p.fresh) return;
    // bit j -> 0x01 in byte j
    auto spread = [](unsigned b) -> unsigned long long {
        const unsigned long long v = ((b & 0xFFu) * 0x0101010101010101ull) & 0x8040201008040201ull;
        return ((v + 0x7F7F7F7F7F7F7F7Full) >> 7) & 0x0101010101010101ull;
    };
    // carved: clear bit0, set bit1; pushed by a carved neighbour: set bit1
    unsigned long long *sp = (unsigned long long *)(state + row * p.X + (size_t)g * kV);
    if (kV == 8) {
        const unsigned long long s = p.fresh ? 0x0101010101010101ull : sp[0];
        const unsigned long long E = spread(e), N = spread(nb);
        const unsigned long long ns = (s & ~E) | ((E | N) << 1);
        if (ns != s || p.fresh) sp[0] = ns;
    } else {
        ulonglong2 s;
        if (p.fresh)
            s.x = s.y = 0x0101010101010101ull;
        else
            s = *(const ulonglong2 *)sp;
        const unsigned long long E0 = spread(e), N0 = spread(nb);
        const unsigned long long E1 = spread(e >> 8), N1 = spread(nb >> 8);
        ulonglong2 ns;
        ns.x = (s.x & ~E0) | ((E0 | N0) << 1);
        ns.y = (s.y & ~E1) | ((E1 | N1) << 1);
        if (ns.x != s.x || ns.y != s.y || p.fresh) *(ulonglong2 *)sp = ns;
    }
}

}  // namespace arvx
