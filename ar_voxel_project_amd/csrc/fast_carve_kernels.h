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
#pragma once

#include "arvx_device.h"

namespace arvx {

struct FloodParams {
    int X, Y, Z;
    int XW;  // 64-bit words per row
    unsigned long long *open;
    unsigned long long *reach;
    int *changed;
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
        o = !(carved_tmp[i] & 1u) && !(state[i] & 2u);
    }
    const unsigned long long b = __ballot(o);
    if ((threadIdx.x & 63) == 0) {
        p.open[wv] = b;
        p.reach[wv] = (wv == 0) ? (b & 1ull) : 0ull;  // seed: voxel (0,0,0) if it is open
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

__global__ __launch_bounds__(256) void flood_step_kernel(const FloodParams p) {
    __shared__ unsigned long long tile[18][18];  // [z][y] with a one-row halo
    const int ty = threadIdx.x & 15, tz = threadIdx.x >> 4;
    const int tilesY = (p.Y + 15) >> 4;
    const int xw = blockIdx.x % p.XW;
    const int by = (blockIdx.x / p.XW) % tilesY;
    const int bz = blockIdx.x / (p.XW * tilesY);
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
    if (ok && r != r0) {
        p.reach[((size_t)z * p.Y + y) * p.XW + xw] = r;
        *p.changed = 1;
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
    uint8_t s = state[i];
    if (bit(x, y, z)) {
        s = (uint8_t)((s & ~1u) | 2u);  // carved (src/VoxelCarving.cpp:125) and visited (:108)
    } else if (!(s & 2u)) {
        const bool nb = bit(x - 1, y, z) || bit(x + 1, y, z) || bit(x, y - 1, z) ||
                        bit(x, y + 1, z) || bit(x, y, z - 1) || bit(x, y, z + 1);
        if (nb || i == 0) s |= 2u;  // pushed by a carved neighbour (:132-163) or the seed (:100)
    }
    state[i] = s;
}

}  // namespace arvx
