// views_kernels.h -- what arvx_set_views[_device] derives from the undistorted u8 masks:
// per view a 1-bit background plane (bit = all channel bytes zero, reference
// src/VoxelCarving.cpp:49-50) and the summed-area table of FOREGROUND pixels that the
// rectangle tests of the carve kernels query.
//
// Three launches for all views (below: "summed-area table") --
//   views_bits_kernel        mask bytes -> background bits, four pixels per lane
//   views_tile_sums_kernel   per-tile row and column sums from the bit plane (small arrays)
//   views_table_kernel       the table, written once: per row a popcount, two adds, a store
// (Round 1: memset + mask_to_bits + sat_rows + sat_cols, 79 us for 36 views of 640 x 480: the
// row pass wrote the table, the column pass read and rewrote it -- 132 MB for a 44 MB table.
// Tried and dropped in round 2: a table per BLOCK of 2 x 2 pixels, rectangles rounded outwards
// -- four times smaller and cheaper to derive, but the one-pixel rim of extra "mixed" answers
// cost the exact kernel more than the derivation saved: 512^3 exact +11 %, 1024^3 +29 %.)
#pragma once

#include "arvx_device.h"

namespace arvx {

// bit i of plane v = 1 iff all C channel bytes of pixel i are zero; the word behind the last
// pixel word stays zero (voxels outside the image read it).  npix4 = ceil(npix / 4).
template <int C>
__global__ __launch_bounds__(256) void views_bits_kernel(const uint8_t *__restrict__ masks,
                                                         int npix, uint32_t *__restrict__ bg,
                                                         int bgWords) {
    const int v = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;  // pixels 4 q .. 4 q + 3
    uint32_t *dst = bg + (size_t)v * bgWords;
    if (q == 0) dst[bgWords - 1] = 0;
    const uint8_t *src = masks + ((size_t)v * npix + 4 * (size_t)q) * C;
    uint32_t nib = 0;
    if (4 * q + 3 < npix && (((uintptr_t)src) & 3u) == 0) {
        uint32_t w[C];
#pragma unroll
        for (int c = 0; c < C; ++c) w[c] = reinterpret_cast<const uint32_t *>(src)[c];
        const uint8_t *b = reinterpret_cast<const uint8_t *>(w);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool isbg = true;
#pragma unroll
            for (int c = 0; c < C; ++c) isbg = isbg && (b[j * C + c] == 0);
            nib |= (isbg ? 1u : 0u) << j;
        }
    } else {
        for (int j = 0; j < 4; ++j) {
            if (4 * q + j >= npix) break;
            bool isbg = true;
            for (int c = 0; c < C; ++c) isbg = isbg && (src[j * C + c] == 0);
            nib |= (isbg ? 1u : 0u) << j;
        }
    }
    // eight neighbouring lanes make one 32-bit word
    uint32_t word = nib << (4 * (threadIdx.x & 7));
    word |= __shfl_xor(word, 1);
    word |= __shfl_xor(word, 2);
    word |= __shfl_xor(word, 4);
    const int w0 = q >> 3;
    if ((threadIdx.x & 7) == 0 && w0 < bgWords - 1) dst[w0] = word;
}

// one-channel masks, sixteen pixels per lane (one 16-byte load), two lanes per word
__global__ __launch_bounds__(256) void views_bits16_kernel(const uint8_t *__restrict__ masks,
                                                           int npix, uint32_t *__restrict__ bg,
                                                           int bgWords) {
    const int v = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;  // pixels 16 q .. 16 q + 15 (npix % 32 == 0)
    uint32_t *dst = bg + (size_t)v * bgWords;
    if (q == 0) dst[bgWords - 1] = 0;
    uint32_t half = 0;
    if (16 * (size_t)q < (size_t)npix) {
        const uint4 w = *reinterpret_cast<const uint4 *>(masks + (size_t)v * npix + 16 * (size_t)q);
        const uint32_t d[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // a zero byte -> bit: (x - 0x01010101) & ~x & 0x80808080 marks zero bytes exactly
            // when no borrow crosses a byte, which the per-byte form below avoids
            uint32_t z = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) z |= (((d[k] >> (8 * b)) & 0xffu) == 0u ? 1u : 0u) << b;
            half |= z << (4 * k);
        }
    }
    uint32_t word = half << (16 * (threadIdx.x & 1));
    word |= __shfl_xor(word, 1);
    const int w0 = q >> 1;
    if ((threadIdx.x & 1) == 0 && w0 < bgWords - 1) dst[w0] = word;
}

// any channel count (the reference's masks have 3, the bench's 1)
__global__ __launch_bounds__(256) void views_bits_generic_kernel(const uint8_t *__restrict__ masks,
                                                                 int C, int npix,
                                                                 uint32_t *__restrict__ bg,
                                                                 int bgWords) {
    const int v = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    uint32_t *dst = bg + (size_t)v * bgWords;
    if (pix == 0) dst[bgWords - 1] = 0;
    bool isbg = false;
    if (pix < npix) {
        const uint8_t *q = masks + ((size_t)v * npix + pix) * C;
        isbg = true;
        for (int c = 0; c < C; ++c) isbg = isbg && (q[c] == 0);
    }
    const unsigned long long b = __ballot(isbg);
    if ((threadIdx.x & 63) == 0) {
        const int w0 = pix >> 5;
        if (w0 < bgWords - 1) dst[w0] = (uint32_t)b;
        if (w0 + 1 < bgWords - 1) dst[w0 + 1] = (uint32_t)(b >> 32);
    }
}

// ---- summed-area table, in one pass over the output ------------------------------------------
//
// table[Y][X] = foreground pixels in rows < Y, columns < X, for Y <= H, X <= W, stored modulo
// 2^16 (two bytes per entry: see classify_box); ld = 64 * ceil((W + 1) / 64) entries per row, so
// that every row starts on a 128-byte line.  The table is cut into tiles of 64 columns x 64 image
// rows: tile (I, J) holds the entries X = 64 J + c, c = lane, of the rows Y = 64 I + r + 1 --
// whole lines -- and covers the pixel columns 64 J .. 64 J + 63 (for 640-pixel rows a row of the
// tile is two aligned words of the bit plane).  With fg(y, c) = foreground of pixel (64 J + c, y):
//   table[y + 1][64 J + c] = LT(I, J) + Lin(y, J) + A(I, J, c) + sum over the tile's rows
//                            y' <= y of below(y', c)
//   below(y', c) = fg of row y' in the tile's columns < c           (v_mbcnt: bits below the lane)
//   Lin(y, J)    = fg of the tile's rows <= y in the tile columns left of J
//   LT(I, J)     = fg above tile row I and left of tile column J
//   A(I, J, c)   = fg of the rows above tile row I in the tile's columns < c
// views_tile_sums_kernel takes three small arrays from the bit plane (per row and tile column
// the row's count, per tile row and tile column the running count along c, per tile its total);
// views_table_kernel sums what it needs of them (a few independent loads per lane) and writes the
// table: per row one broadcast LDS read (the row's bits, its count to the left), the two halves
// of v_mbcnt adding onto the running value, one add, the pointer and the store -- four vector
// instructions for 64 entries.  The table's bytes are written once and never read back; the lanes of the last tile
// column beyond X = W write into the row's padding (ld = 64 TJ), so that no store is predicated.
// In both kernels lane r first LOADS row r's 64 bits (one round trip for the whole tile) and the
// rows are then taken from the lanes one by one.  Four tiles per workgroup, one per wave.
// (Round 3, measured with tools/vt_probe.sh: the tile's columns used to be shifted by one pixel
// (inclusive counts: a 64-bit shift and a mask per row on top, the row loop was 13 instructions
// and bound by their issue at three waves per SIMD).)
#ifndef ARVX_TILE_ROWS
#define ARVX_TILE_ROWS 64
#endif
constexpr int kTileRows = ARVX_TILE_ROWS;

// 64 foreground bits of row y starting at pixel column x0 >= 0 (bit j = pixel x0 + j; 0
// outside the image)
__device__ __forceinline__ unsigned long long row_fg64(const uint32_t *__restrict__ bits,
                                                       int bgWords, int W, int H, int y, int x0) {
    if (y >= H || x0 >= W) return 0ull;
    const long long pos = (long long)y * W + x0;
    const int w = (int)(pos >> 5), sh = (int)(pos & 31);
    const int last = bgWords - 1;  // the always-zero word
    const int n = min(64, W - x0);  // valid columns
    const unsigned long long valid = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    const unsigned long long lo = bits[min(w, last)], mid = bits[min(w + 1, last)];
    unsigned long long bg = lo | (mid << 32);
    if (sh) {  // (wave-uniform whenever W is a multiple of 32)
        const unsigned long long hi = bits[min(w + 2, last)];
        bg = (bg >> sh) | (hi << (64 - sh));
    }
    return ~bg & valid;
}

// the tile's 64 columns of row y: bit c = foreground of pixel column 64 J + c
__device__ __forceinline__ unsigned long long tile_row_fg(const uint32_t *__restrict__ bits,
                                                          int bgWords, int W, int H, int y, int J) {
    return row_fg64(bits, bgWords, W, H, y, 64 * J);
}

// a wave's own LDS writes are visible to its own reads behind this (no workgroup barrier needed)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// four tiles per workgroup: this wave's tile, or false behind the last one
__device__ __forceinline__ bool wave_tile(int TJ, int TI, int V, int &J, int &I, int &v) {
    const int tile = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4 + (threadIdx.x >> 6)));
    if (tile >= TJ * TI * V) return false;
    J = tile % TJ;
    I = (tile / TJ) % TI;
    v = tile / (TJ * TI);
    return true;
}

// one wave per tile: rowsum[v][y][J] = fg of row y inside tile column J; T[v][I][J][c] = fg of
// tile row I in the tile's columns < c; tilesum[v][I][J] = fg of the tile
__global__ __launch_bounds__(256) void views_tile_sums_kernel(const uint32_t *__restrict__ bg,
                                                              int bgWords, int W, int H, int TJ,
                                                              int TI, int V, int *__restrict__ rowsum,
                                                              int *__restrict__ T,
                                                              int *__restrict__ tilesum) {
    int J, I, v;
    if (!wave_tile(TJ, TI, V, J, I, v)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t *bits = bg + (size_t)v * bgWords;
    const int yr = I * kTileRows + lane;
    const unsigned long long mine = tile_row_fg(bits, bgWords, W, H, yr, J);  // row `lane`
    if (yr < H) rowsum[((size_t)v * H + yr) * TJ + J] = __popcll(mine);
    // column c's count: the rows that have bit c set, counted by a ballot (scalar) and handed to
    // lane c -- two vector instructions per column
    const uint32_t lo = (uint32_t)mine, hi = (uint32_t)(mine >> 32);
    int col = 0;
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        const unsigned long long rows = __ballot((((c < 32) ? lo : hi) >> (c & 31)) & 1u);
        const int n = __popcll(rows);  // (scalar)
        asm("v_writelane_b32 %0, %1, %2" : "+v"(col) : "s"(n), "n"(c));
    }
    int sc = col;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(sc, d);
        if (lane >= d) sc += t;
    }
    T[(((size_t)v * TI + I) * TJ + J) * 64 + lane] = sc - col;
    if (lane == 63) tilesum[((size_t)v * TI + I) * TJ + J] = sc;
}

// one wave per tile: the table entries of its 64 columns x 64 rows
__global__ __launch_bounds__(256) void views_table_kernel(const uint32_t *__restrict__ bg,
                                                          int bgWords, int W, int H, int TJ, int TI,
                                                          int V, const int *__restrict__ rowsum,
                                                          const int *__restrict__ T,
                                                          const int *__restrict__ tilesum,
                                                          uint16_t *__restrict__ sat, int satStride,
                                                          int ld) {
    int J, I, v;
    if (!wave_tile(TJ, TI, V, J, I, v)) return;
    const int lane = threadIdx.x & 63;
    const uint32_t *bits = bg + (size_t)v * bgWords;
    uint16_t *tab = sat + (size_t)v * satStride;  // entries modulo 2^16 (carve_kernels.h, classify_box)
    const int yr = I * kTileRows + lane;
    const unsigned long long mine = tile_row_fg(bits, bgWords, W, H, yr, J);  // row `lane`
    // row `lane`'s count in the tile columns to the left (Lin grows by it from row to row)
    int left = 0;
    if (yr < H)
        for (int k = 0; k < J; ++k) left += rowsum[((size_t)v * H + yr) * TJ + k];
    // LT: the tiles above and to the left, a few per lane
    int lt = 0;
    for (int e = lane; e < I * J; e += 64) lt += tilesum[((size_t)v * TI + e / J) * TJ + e % J];
    // A: the tile rows above, this lane's column
    int acc = 0;
    for (int k = 0; k < I; ++k) acc += T[(((size_t)v * TI + k) * TJ + J) * 64 + lane];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) lt += __shfl_xor(lt, d);
    const unsigned off = 64u * (unsigned)J + (unsigned)lane;  // this lane's table column X
    if (I == 0) tab[off] = 0;                                   // row 0 of the table
    const int nrows = min(kTileRows, H - I * kTileRows);
    // the rows go through LDS: one broadcast read per row (bits + count to the left) instead of
    // three lane reads -- 4 vector instructions per row of 64 entries
    __shared__ uint4 s_row[4][kTileRows];
    uint4 *const rows = s_row[threadIdx.x >> 6];
    rows[lane] = make_uint4((uint32_t)mine, (uint32_t)(mine >> 32), (uint32_t)left, 0u);
    wave_lds_fence();
    uint32_t run = (uint32_t)(acc + lt);
    uint16_t *row = tab + (size_t)(I * kTileRows + 1) * ld;  // (wave-uniform: scalar registers)
#pragma unroll 4
    for (int r = 0; r < nrows; ++r) {
        const uint4 f = rows[r];
        run = __builtin_amdgcn_mbcnt_hi(f.y, __builtin_amdgcn_mbcnt_lo(f.x, run)) + f.z;
        row[off] = (uint16_t)run;
        row += ld;
    }
}

}  // namespace arvx
