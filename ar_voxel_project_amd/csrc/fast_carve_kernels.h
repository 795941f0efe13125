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

// The bit planes of the flood fill are row-major (64 voxels of a row per word, XW words per
// row); the records are tile-major (a tile = 64 x 8 x 8 voxels = four records; entry r of a
// record = row (y & 7, z & 7) of the tile).  Both conversions below take one workgroup per ROW
// OF TILES (all tiles with the same ty, tz) and go through LDS, so that each side is read and
// written in whole lines: with lane = entry r on the record side (128 contiguous bytes per
// access) and lane = tile along x on the plane side (a row's XW words are contiguous).
// LDS: 64 rows x (chunk of up to 32 tiles + 1) words.
constexpr int kFloodChunk = 32;  // tiles along x per pass (apply: 2 x 64 x 33 words = 33 KB of LDS)

// the tile row's (ty, tz) and the records of its tile tx
__device__ __forceinline__ void flood_tile_row(const CarveParams &g, int b, int &ty, int &tz) {
    ty = b % g.tilesY;
    tz = b / g.tilesY;
}

// open = carvable & !seen0 from the two sets of sub-tile records (arvx_device.h): `carv` is a
// fresh model carved by all views -- a voxel is carvable where its occupancy bit is gone --,
// g.rec the model's own records (not read for a fresh model).  reach = 0, and the seed at
// voxel (0,0,0) if it is open.
// carv_code (may be null): per coarse tile, 0 = the tile's records hold what the carve left, 1 =
// the carve emptied the whole tile (all carvable; its records were never written), 2 / 3 = it left
// the whole tile untouched (nothing carvable) -- the carve's lazy codes, arvx_device.h.
__global__ __launch_bounds__(256) void flood_open_from_rec_kernel(const CarveParams g,
                                                                  const uint16_t *__restrict__ carv,
                                                                  const uint8_t *__restrict__ carv_code,
                                                                  const FloodParams p) {
    extern __shared__ unsigned long long lds[];
    int ty, tz;
    flood_tile_row(g, blockIdx.x, ty, tz);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int tx0 = 0; tx0 < p.XW; tx0 += kFloodChunk) {
        const int nt = min(kFloodChunk, p.XW - tx0), ld = nt + 1;
        for (int t = wave; t < nt; t += 4) {  // lane = entry r of the tile's records
            const int tx = tx0 + t;
            const size_t rec0 = rec_index(g, tx, ty, tz, 0) * kRecU16;
            const int code = carv_code ? (int)carv_code[coarse_of(g, tx, ty, tz)] : 0;
            unsigned long long w = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // the tile's four sub-tiles: 16 voxels each
                if (64 * tx + 16 * k >= p.X) break;
                // carvable = the occupancy bit is gone (voxels outside the grid never had one)
                unsigned long long o =
                    code ? (uint16_t)~lazy_occ(g, code, tx, ty, tz, k, lane) : (uint16_t)~carv[rec0 + k * kRecU16 + lane];
                if (!p.fresh)
                    o &= (unsigned long long)(uint16_t)~g.rec[rec0 + k * kRecU16 + 64 + lane];
                w |= o << (16 * k);
            }
            const int nx = p.X - 64 * tx;  // voxels behind the end of the row: not open
            if (nx < 64) w &= (1ull << nx) - 1ull;
            lds[lane * ld + t] = w;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 64 * nt; idx += 256) {  // threads along the rows
            const int r = idx / nt, t = idx % nt;
            const int y = ty * kTileY + (r & 7), z = tz * kTileZ + (r >> 3);
            if (y >= p.Y || z >= p.Z) continue;
            const size_t i = ((size_t)z * p.Y + y) * p.XW + tx0 + t;
            const unsigned long long w = lds[r * ld + t];
            p.open[i] = w;
            p.reach[i] = (i == 0) ? (w & 1ull) : 0ull;
        }
        __syncthreads();
    }
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
    // r: reached, pr: the propagator of the sweep in progress.  A thread owns rows
    // threadIdx.x + 1024 k (k < 4: rows <= kFloodMaxTileRows) and keeps their `full` words.
    unsigned long long *r = lds, *pr = lds + rows;
    unsigned long long fr[4], mine[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 1024 * k;
        fr[k] = i < rows ? full[i] : 0ull;
        mine[k] = (i == 0) ? (fr[k] & 1ull) : 0ull;  // seed: the tile of voxel (0,0,0), if full
    }
    // Every round: an occluded fill along x inside the words, then Kogge-Stone occluded fills
    // along +y, -y, +z, -z over the rows (log2 steps each: g |= p & g[-d]; p &= p[-d]), so a
    // round carries the front across the whole grid in each direction and the number of
    // rounds is the number of bends of the longest shortest path, not its length.  (Before:
    // chaotic relaxation, one tile per step: 59 us at 512^3, 319 us at 1024^3.)  A round that
    // changes nothing ends the kernel; every other round adds at least one tile.
    for (;;) {
        int ch = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = threadIdx.x + 1024 * k;
            if (i < rows) {
                const unsigned long long rn = fill_row(mine[k], fr[k]);
                ch |= (rn != mine[k]);
                mine[k] = rn;
            }
        }
        for (int dir = 0; dir < 4; ++dir) {
            const int len = dir < 2 ? tilesY : tilesZ;
            const int stride = dir < 2 ? 1 : tilesY;
            const bool back = dir & 1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = threadIdx.x + 1024 * k;
                if (i < rows) {
                    r[i] = mine[k];
                    pr[i] = fr[k];
                }
            }
            __syncthreads();
            for (int d = 1; d < len; d <<= 1) {
                unsigned long long gn[4], pn[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = threadIdx.x + 1024 * k;
                    gn[k] = pn[k] = 0ull;
                    if (i < rows) {
                        const int pos = dir < 2 ? i % tilesY : i / tilesY;
                        const int from = back ? pos + d : pos - d;
                        if (from >= 0 && from < len) {
                            const int j = back ? i + d * stride : i - d * stride;
                            gn[k] = r[j];
                            pn[k] = pr[j];
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = threadIdx.x + 1024 * k;
                    if (i < rows) {
                        const unsigned long long p0 = pr[i];
                        const unsigned long long rn = r[i] | (p0 & gn[k]);
                        r[i] = rn;
                        pr[i] = p0 & pn[k];
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = threadIdx.x + 1024 * k;
                if (i < rows) {
                    const unsigned long long rn = r[i];
                    ch |= (rn != mine[k]);
                    mine[k] = rn;
                }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(ch)) break;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = threadIdx.x + 1024 * k;
        if (i < rows) reached[i] = mine[k];
    }
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

// occ &= !E ; seen |= E | N6(E) | origin, on the model's records; one workgroup per row of
// tiles as above: the reach plane is read row by row (the row itself and its four neighbours in
// y and z, lane = tile along x), the records are written entry by entry.  For a fresh model
// nothing is read from the records and EVERY record of the row is written (padding as occ 0 /
// seen 1).  `nrows` = tile rows of the records (the grid's, padded to whole coarse tiles).
__global__ __launch_bounds__(256) void flood_apply_rec_kernel(const CarveParams g,
                                                              const FloodParams p) {
    extern __shared__ unsigned long long lds[];
    // tile rows in the order of the coarse padding: (ty, tz) over the padded extent
    const int padY = g.coarseY << g.cyShift;
    const int ty = blockIdx.x % padY, tz = blockIdx.x / padY;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int padX = g.coarseX;  // tiles along x in the records (= XW)
    for (int tx0 = 0; tx0 < padX; tx0 += kFloodChunk) {
        const int nt = min(kFloodChunk, padX - tx0), ld = nt + 1;
        unsigned long long *e_s = lds, *v_s = lds + 64 * ld;
        for (int idx = threadIdx.x; idx < 64 * nt; idx += 256) {  // threads along the rows
            const int r = idx / nt, t = idx % nt;
            const int y = ty * kTileY + (r & 7), z = tz * kTileZ + (r >> 3);
            const int tx = tx0 + t;
            unsigned long long e = 0, vis = 0;
            if (y < p.Y && z < p.Z && tx < p.XW) {
                const size_t row = (size_t)z * p.Y + y;
                const unsigned long long *c = p.reach + row * p.XW + tx;
                e = c[0];
                unsigned long long nb = (e << 1) | (e >> 1);
                if (tx > 0) nb |= c[-1] >> 63;
                if (tx + 1 < p.XW) nb |= c[1] << 63;
                if (y > 0) nb |= c[-(ptrdiff_t)p.XW];
                if (y + 1 < p.Y) nb |= c[p.XW];
                if (z > 0) nb |= c[-(ptrdiff_t)p.XW * p.Y];
                if (z + 1 < p.Z) nb |= c[(ptrdiff_t)p.XW * p.Y];
                if (row == 0 && tx == 0) nb |= 1ull;  // the seed is visited anyway (:100)
                const int nx = p.X - 64 * tx;
                const unsigned long long in = nx >= 64 ? ~0ull : ((1ull << nx) - 1ull);
                // carved (:125, :108) or pushed by a carved neighbour (:132-163)
                vis = (e | nb) & in;
            }
            e_s[r * ld + t] = e;
            v_s[r * ld + t] = vis;
        }
        __syncthreads();
        for (int t = wave; t < nt; t += 4) {  // lane = entry r of the tile's records
            const int tx = tx0 + t;
            const unsigned long long e = e_s[lane * ld + t], vis = v_s[lane * ld + t];
            if (!p.fresh && __ballot(vis != 0) == 0) continue;
            const int y = ty * kTileY + (lane & 7), z = tz * kTileZ + (lane >> 3);
            unsigned long long in = 0;
            if (y < p.Y && z < p.Z && tx < p.XW) {
                const int nx = p.X - 64 * tx;
                in = nx >= 64 ? ~0ull : ((1ull << nx) - 1ull);
            }
            uint16_t *rec = g.rec + rec_index(g, tx, ty, tz, 0) * kRecU16;
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // the tile's four sub-tiles: 16 voxels each
                const uint32_t e16 = (uint32_t)(e >> (16 * k)) & 0xffffu;
                const uint32_t v16 = (uint32_t)(vis >> (16 * k)) & 0xffffu;
                const uint32_t in16 = (uint32_t)(in >> (16 * k)) & 0xffffu;
                uint16_t *o = rec + k * kRecU16 + lane, *sn = o + 64;
                if (p.fresh) {
                    *o = (uint16_t)(in16 & ~e16);
                    *sn = (uint16_t)((~in16 & 0xffffu) | v16);
                } else if (v16) {
                    if (e16) *o = (uint16_t)(*o & ~e16);
                    *sn = (uint16_t)(*sn | v16);
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace arvx
