// views_kernels.h -- what arvx_set_views[_device] derives from the undistorted u8 masks:
// per view a 1-bit background plane (bit = all channel bytes zero, reference
// src/VoxelCarving.cpp:49-50) and the summed-area table of FOREGROUND pixels that the
// rectangle tests of the carve kernels query ((H+1) x (W+1) ints).
//
// Two launches for all views (images whose width is a multiple of 64; other widths take
// the general kernels in carve_kernels.h):
//   views_rows_kernel  one wave per image row: mask bytes -> background bits (ballot) and
//                      the row's running foreground count (popcount of the ballot below the
//                      lane -- no shuffles), written as the row of the table
//   views_cols_kernel  column sums: 16 columns x 32 row groups per workgroup; every thread
//                      loads its <= 16 rows once, the groups exchange their totals through
//                      LDS, offsets are added in registers and the rows written back -- one
//                      load round trip instead of a chain of H dependent adds
// HBM: W*H*C bytes in, W*H/8 + 4 (W+1)(H+1) bytes out per view, the table re-read once from
// L2.  (Round 1: memset + mask_to_bits + sat_rows + sat_cols, 79 us for 36 views of 640x480.)
#pragma once

#include "arvx_device.h"

namespace arvx {

template <int C>
__global__ __launch_bounds__(256) void views_rows_kernel(const uint8_t *__restrict__ masks, int W,
                                                         int H, uint32_t *__restrict__ bg,
                                                         int bgWords, int *__restrict__ sat,
                                                         int satStride) {
    const int v = blockIdx.y;
    const int yrow = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint32_t *b = bg + (size_t)v * bgWords;
    if (blockIdx.x == 0 && threadIdx.x == 0) b[bgWords - 1] = 0;  // the always-zero word
    if (yrow >= H) return;
    const uint8_t *row = masks + ((size_t)v * H + yrow) * (size_t)W * C;
    int *out = sat + (size_t)v * satStride + (size_t)(yrow + 1) * (W + 1);
    if (lane == 0) out[0] = 0;
    int carry = 0;
    for (int x0 = 0; x0 < W; x0 += 64) {  // W % 64 == 0
        const uint8_t *q = row + (size_t)(x0 + lane) * C;
        bool isbg = true;
#pragma unroll
        for (int c = 0; c < C; ++c) isbg = isbg && (q[c] == 0);
        const unsigned long long bgm = __ballot(isbg);
        const unsigned long long fgm = ~bgm;
        const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fgm >> 32),
                                                         __builtin_amdgcn_mbcnt_lo((uint32_t)fgm, 0u));
        out[x0 + lane + 1] = carry + below + (isbg ? 0 : 1);
        carry += __popcll(fgm);
        if (lane == 0) {
            const size_t w0 = ((size_t)yrow * W + x0) >> 5;
            b[w0] = (uint32_t)bgm;
            b[w0 + 1] = (uint32_t)(bgm >> 32);
        }
    }
}

// Column sums.  A workgroup takes 16 columns x kColGroups row groups; a thread holds its
// (at most 16) rows in registers: ONE load round trip, the group totals meet in LDS, the
// offsets are added and the rows written back.  Images taller than 16 * kColGroups rows walk
// their rows twice (kTall).
constexpr int kColGroups = 32, kColsPerWg = 16;

template <bool kTall>
__global__ __launch_bounds__(kColsPerWg * kColGroups) void views_cols_kernel(int W, int H,
                                                                             int *__restrict__ sat,
                                                                             int satStride) {
    __shared__ int part[kColGroups][kColsPerWg];
    const int v = blockIdx.y;
    const int c = threadIdx.x % kColsPerWg, g = threadIdx.x / kColsPerWg;
    const int col = blockIdx.x * kColsPerWg + c;  // 0..W
    const int R = (H + kColGroups - 1) / kColGroups;  // rows per group (<= 16 unless kTall)
    const int r0 = 1 + g * R, r1 = min(H + 1, r0 + R);  // table rows [r0, r1)
    const bool ok = col <= W;
    const size_t ld = (size_t)(W + 1);
    int *s = sat + (size_t)v * satStride + (ok ? col : 0);
    int t[16];
    int sum = 0;
    if (!kTall) {
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = (ok && r0 + k < r1) ? s[(size_t)(r0 + k) * ld] : 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) sum += t[k];
    } else {
        for (int y0 = r0; y0 < r1; y0 += 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = (ok && y0 + k < r1) ? s[(size_t)(y0 + k) * ld] : 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) sum += t[k];
        }
    }
    part[g][c] = sum;
    __syncthreads();
    int acc = 0;
    for (int k = 0; k < g; ++k) acc += part[k][c];
    if (!ok) return;
    if (g == 0) s[0] = 0;
    if (!kTall) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc += t[k];
            if (r0 + k < r1) s[(size_t)(r0 + k) * ld] = acc;
        }
    } else {
        for (int y0 = r0; y0 < r1; y0 += 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t[k] = (y0 + k < r1) ? s[(size_t)(y0 + k) * ld] : 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                acc += t[k];
                if (y0 + k < r1) s[(size_t)(y0 + k) * ld] = acc;
            }
        }
    }
}

}  // namespace arvx
