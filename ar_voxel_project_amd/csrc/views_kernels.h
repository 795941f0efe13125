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

// ---- everything in ONE launch (one-channel masks, W a multiple of 64, W < 1024) -----------------
//
// Round 4.  The three launches above are a chain of dependent grids of a few thousand waves: 22 us
// for 36 views of 640 x 480 of which most is dispatch, ramp and the small arrays going through
// memory from one launch to the next.  Here a workgroup owns a STRIP of one view -- tile row I, 64
// image rows -- with one wave per tile column J (the same tiles and the same row loop as above) and
// reads the mask bytes itself:
//   own rows          16 bytes per lane (row 4 (lane / 4) + k, columns 64 J + 16 (lane % 4)), four
//                     loads; the four lanes of a row OR their 16 flags into the row's 64-bit word,
//                     and lane r ends up with row r (k = lane % 4 is its own register);
//   bit plane         the strip's words go through LDS and out as one contiguous run (whole lines);
//   left of the tile  the waves of the workgroup are the tile columns: per-row counts through LDS;
//   above the tile    every strip counts the foreground of its OWN rows per column (the flags of
//                     the bytes it has loaded, added as packed bytes, summed over the 16 row slots)
//                     and publishes the counts; a strip sums what the strips above it published.
// That hand-off is the only thing that crosses workgroups.  It follows the guide's granule form
// (MI355X_MICROARCH.md, inter-workgroup visibility): 8-byte words {two 16-bit counts, 32-bit tag}
// written by one sc1 store each and polled with sc1 loads -- a word is valid when its tag is this
// launch's, no flag, no fence, nothing to reset.  Forward progress does not depend on dispatch
// order or residency: a workgroup takes its strip from a per-view TICKET (64-bit counter, never
// reset: ticket t = strip t % TI of launch t / TI, which is also the tag), so the strips it waits
// for (lower tickets of the same launch) belong to workgroups that are already running and that
// wait for nothing above them.  (First version of this round: no hand-off, every strip counted the
// rows above it again from the mask bytes -- 287 KB of reads on ONE compute unit for the lowest
// strip, 11 B/cycle: 17 us, and 32 us when 92 registers left room for one workgroup per unit.)
// tests/test_carve_gpu.py::test_view_tables_against_numpy compares every entry of plane and table.
constexpr int kStripMaxWaves = 16;

// 16 mask bytes -> 16 flags, bit i = byte i is not zero
__device__ __forceinline__ uint32_t nonzero16(const uint4 w) {
    const uint32_t d[4] = {w.x, w.y, w.z, w.w};
    uint32_t f = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // bit 7 of a byte of t: the byte is not zero (no carry leaves a byte: 0x7f + 0x7f < 0x100)
        const uint32_t t = (((d[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d[k]) & 0x80808080u;
        f |= bytes_to_nibble(t >> 7) << (4 * k);
    }
    return f;
}
// the same as packed bytes: byte i of the result = 1 iff byte i of x is not zero
__device__ __forceinline__ uint32_t nonzero_bytes(const uint32_t x) {
    return ((((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) >> 7) & 0x01010101u;
}

// 16 three-channel pixels (48 bytes) -> 16 flags, bit i = some byte of pixel i is not zero (the
// reference's masks are three-channel images: background = all channel bytes zero,
// src/VoxelCarving.cpp:49-50)
__device__ __forceinline__ uint32_t nonzero16x3(const uint4 a, const uint4 b, const uint4 c) {
    const uint32_t d[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
    // bit i of m[h] = byte 24 h + i of the run is not zero
    uint32_t m[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint32_t t = (((d[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d[k]) & 0x80808080u;
        m[k / 6] |= bytes_to_nibble(t >> 7) << (4 * (k % 6));
    }
    uint32_t f = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {  // eight pixels per 24 byte flags: OR the triples, keep every third bit
        uint32_t x = (m[h] | (m[h] >> 1) | (m[h] >> 2)) & 0x249249u;
        x = (x | (x >> 2)) & 0x0C30C3u;
        x = (x | (x >> 4)) & 0x00F00Fu;
        x = (x | (x >> 8)) & 0xFFu;
        f |= x << (8 * h);
    }
    return f;
}

// ctr: one 64-bit ticket counter per view (zero when allocated, never reset); gran: the published
// column counts, [view][strip][W / 2] granules; err: set when a wait gives up (never seen; the
// host checks it at its next synchronisation).  C: channel bytes per pixel, 1 or 3.
template <int C>
__global__ __launch_bounds__(64 * kStripMaxWaves) void views_strip_kernel(
    const uint8_t *__restrict__ masks, int W, int H, int TI, uint32_t *__restrict__ bg, int bgWords,
    uint16_t *__restrict__ sat, int satStride, int ld, unsigned long long *__restrict__ ctr,
    unsigned long long *__restrict__ gran, unsigned *__restrict__ err) {
    __shared__ unsigned long long s_bits[64 * (kStripMaxWaves - 1)];  // [row][pixel tile column]
    __shared__ int s_rowcnt[kStripMaxWaves][64];
    __shared__ uint16_t s_col[kStripMaxWaves][64];
    __shared__ int s_tot[kStripMaxWaves];
    __shared__ uint4 s_row[kStripMaxWaves][64];
    __shared__ unsigned long long s_ticket;
    const int v = blockIdx.y;
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ctr[v], 1ull);
    __syncthreads();
    const unsigned long long ticket = s_ticket;
    const int I = (int)(ticket % (unsigned long long)TI);
    const uint32_t tag = (uint32_t)(ticket / (unsigned long long)TI) + 1u;
    const int J = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int TP = W >> 6;  // tile columns that hold pixels; wave TP writes table column W (+ padding)
    const int G = W >> 1;   // granules per strip
    const uint8_t *img = masks + (size_t)v * W * H * C;
    const int y0 = I * 64;
    const int nrows = min(64, H - y0);
    const int rr = lane >> 2, q = lane & 3;

    // ---- this strip's rows: lane r <- the 64 foreground flags of row y0 + r, columns 64 J ..
    uint32_t mine_lo = 0, mine_hi = 0;
    uint32_t acc[4] = {0u, 0u, 0u, 0u};  // columns 64 J + 16 q + 4 d + b: byte b of acc[d]
    if (J < TP) {
        uint4 w[4][C];  // rows 4 rr + k, pixels 64 J + 16 q .. + 15: 16 C bytes
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int y = y0 + 4 * rr + k;
            const uint4 *src = reinterpret_cast<const uint4 *>(img + ((size_t)y * W + 64 * J + 16 * q) * C);
#pragma unroll
            for (int c = 0; c < C; ++c) w[k][c] = (y < H) ? src[c] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t f;
            if constexpr (C == 1)
                f = nonzero16(w[k][0]);
            else
                f = nonzero16x3(w[k][0], w[k][1], w[k][2]);
            uint32_t lo = q < 2 ? f << (16 * q) : 0u, hi = q >= 2 ? f << (16 * (q - 2)) : 0u;
            lo |= __shfl_xor(lo, 1);
            hi |= __shfl_xor(hi, 1);
            lo |= __shfl_xor(lo, 2);
            hi |= __shfl_xor(hi, 2);
            if (k == q) {  // row 4 rr + q = lane
                mine_lo = lo;
                mine_hi = hi;
            }
            if constexpr (C == 1) {
                acc[0] += nonzero_bytes(w[k][0].x);
                acc[1] += nonzero_bytes(w[k][0].y);
                acc[2] += nonzero_bytes(w[k][0].z);
                acc[3] += nonzero_bytes(w[k][0].w);
            } else {
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[d] += nibble_to_bytes((f >> (4 * d)) & 15u);
            }
        }
        // background bits of the row (reference src/VoxelCarving.cpp:49-50: all bytes zero)
        s_bits[lane * TP + J] = (lane < nrows)
                                    ? ~(((unsigned long long)mine_hi << 32) | mine_lo)
                                    : 0ull;
    }
    s_rowcnt[J][lane] = __popc(mine_lo) + __popc(mine_hi);

    // ---- the strip's own foreground per column: packed bytes (<= 4 each) -> packed halves, summed
    // over the 16 row slots (the lanes with the same q), published for the strips below
    if (J < TP && I + 1 < TI) {
        uint32_t e[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            e[2 * d] = acc[d] & 0x00ff00ffu;             // bytes 0 and 2
            e[2 * d + 1] = (acc[d] >> 8) & 0x00ff00ffu;  // bytes 1 and 3
        }
#pragma unroll
        for (int m = 4; m < 64; m <<= 1)
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] += __shfl_xor(e[i], m);
        if (rr == 0) {
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint16_t *o = &s_col[J][16 * q + 4 * d];
                o[0] = (uint16_t)(e[2 * d] & 0xffffu);
                o[1] = (uint16_t)(e[2 * d + 1] & 0xffffu);
                o[2] = (uint16_t)(e[2 * d] >> 16);
                o[3] = (uint16_t)(e[2 * d + 1] >> 16);
            }
        }
        wave_lds_fence();
        if (lane < 32) {
            const uint32_t pay = (uint32_t)s_col[J][2 * lane] | ((uint32_t)s_col[J][2 * lane + 1] << 16);
            granule_store(gran + ((size_t)v * TI + I) * G + 32 * J + lane, pay, tag);
        }
    }
    __syncthreads();

    // ---- the bit plane: the strip's nrows * TP words are one contiguous run
    {
        unsigned long long *dst =
            reinterpret_cast<unsigned long long *>(bg + (size_t)v * bgWords) + (size_t)y0 * TP;
        for (int i = threadIdx.x; i < nrows * TP; i += blockDim.x) dst[i] = s_bits[i];
        if (I == 0 && threadIdx.x == 0) bg[(size_t)v * bgWords + bgWords - 1] = 0u;
    }

    // ---- foreground of column 64 J + lane in the rows above the strip: the counts the strips
    // above have published (all loads of a round in flight together)
    int above = 0;
    if (J < TP && I > 0) {
        const unsigned long long *g0 = gran + (size_t)v * TI * G + 32 * J + (lane >> 1);
        for (unsigned spin = 0;; ++spin) {
            bool ok = true;
            int sum = 0;
            for (int k = 0; k < I; ++k) {
                const unsigned long long g = granule_load(g0 + (size_t)k * G);
                ok = ok && (uint32_t)(g >> 32) == tag;
                sum += (int)(((uint32_t)g >> (16 * (lane & 1))) & 0xffffu);
            }
            if (__all(ok)) {
                above = sum;
                break;
            }
            if (spin > (1u << 20)) {  // (seconds: something is broken; leave a mark and go on)
                if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    int sc = above;  // ... inclusive along the lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(sc, d);
        if (lane >= d) sc += t;
    }
    if (lane == 63) s_tot[J] = sc;
    __syncthreads();

    // ---- the table, as in views_table_kernel
    int left = 0, lt = 0;
    for (int k = 0; k < J; ++k) {
        left += s_rowcnt[k][lane];
        lt += s_tot[k];
    }
    uint16_t *tab = sat + (size_t)v * satStride;
    const unsigned off = 64u * (unsigned)J + (unsigned)lane;
    if (I == 0) tab[off] = 0;
    uint4 *const rows = s_row[J];
    rows[lane] = make_uint4(mine_lo, mine_hi, (uint32_t)left, 0u);
    wave_lds_fence();
    uint32_t run = (uint32_t)(sc - above + lt);
    uint16_t *row = tab + (size_t)(y0 + 1) * ld;
#pragma unroll 4
    for (int r = 0; r < nrows; ++r) {
        const uint4 f = rows[r];
        run = __builtin_amdgcn_mbcnt_hi(f.y, __builtin_amdgcn_mbcnt_lo(f.x, run)) + f.z;
        row[off] = (uint16_t)run;
        row += ld;
    }
}

}  // namespace arvx
