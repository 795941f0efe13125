// bitplane_kernels.h -- one bit per voxel: the neighbourhood tests of the colour
// pass (Model::isInner, reference src/Model.h:126-132) and of the closure
// (3x3x3 .. 9x9x9 box, src/Postprocessing3d.cpp:31-48) as word operations, and
// the ordered compaction of the voxels they select.
//
// Layout: a voxel row (fixed y,z) is XW = ceil(X/64) 64-bit words, voxel x at bit
// x%64 of word x/64; bits past X are zero.  Rows follow each other in (z, y)
// order, so ascending word/bit order is ascending flat index x + X*(y + Y*z).
// A 512^3 plane is 16 MiB and stays in L2; the byte plane is read once to build it.
#pragma once

#include "arvx_device.h"

namespace arvx {

struct BitGrid {
    int X, Y, Z;  // Z = planes held in this bit plane
    int XW;
};

enum BitPred {
    kBitOccupied = 0,  // state bit0
    // what the closure calls occupied: bit0, or painted UNSEEN_COLOR (bit2 from a host
    // Model; or never seen when handleUnseen ran before: w = 1, src/Model.cpp:42)
    kBitClosureOccupied = 1,
};

// predicate on 8 state bytes at once -> 0x01 in every byte that satisfies it
template <int PRED>
__device__ __forceinline__ unsigned long long bit_pred8(unsigned long long s, int apply_unseen) {
    const unsigned long long one = 0x0101010101010101ull;
    if (PRED == kBitOccupied) return s & one;
    unsigned long long m = s | (s >> 2);
    if (apply_unseen) m |= ~(s >> 1);
    return m & one;
}
template <int PRED>
__device__ __forceinline__ bool bit_pred1(uint8_t s, int apply_unseen) {
    if (PRED == kBitOccupied) return s & 1u;
    return (s & 1u) || (s & 4u) || (apply_unseen && !(s & 2u));
}

// X % 8 == 0: one thread turns 8 voxels into one byte of the plane (8-byte load)
template <int PRED>
__global__ __launch_bounds__(256) void bit_pack8_kernel(const uint8_t *__restrict__ state,
                                                        const BitGrid g, int apply_unseen,
                                                        unsigned long long *__restrict__ bits) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int rowBytes = g.XW * 8;
    const size_t nrows = (size_t)g.Y * g.Z;
    if (t >= nrows * rowBytes) return;
    const size_t row = t / rowBytes;
    const int b8 = (int)(t % rowBytes);
    uint8_t b = 0;
    if (b8 * 8 < g.X) {
        const unsigned long long s =
            *(const unsigned long long *)(state + row * g.X + (size_t)b8 * 8);
        // byte j's bit 0 -> bit j
        b = (uint8_t)((bit_pred8<PRED>(s, apply_unseen) * 0x0102040810204080ull) >> 56);
    }
    ((uint8_t *)bits)[t] = b;
}

// X % 32 == 0 and a 16-byte aligned plane: one thread turns 32 voxels into one 32-bit half of
// a word of the plane (two 16-byte loads)
template <int PRED>
__global__ __launch_bounds__(256) void bit_pack32_kernel(const uint8_t *__restrict__ state,
                                                         const BitGrid g, int apply_unseen,
                                                         unsigned long long *__restrict__ bits) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int rowHalves = g.XW * 2;
    const size_t nrows = (size_t)g.Y * g.Z;
    if (t >= nrows * rowHalves) return;
    const size_t row = t / rowHalves;
    const int h = (int)(t % rowHalves);
    uint32_t w = 0;
    if (h * 32 < g.X) {
        const ulonglong2 *src = (const ulonglong2 *)(state + row * g.X + (size_t)h * 32);
        const ulonglong2 a = src[0], b = src[1];
        const unsigned long long mul = 0x0102040810204080ull;
        w = (uint32_t)((bit_pred8<PRED>(a.x, apply_unseen) * mul) >> 56) |
            ((uint32_t)((bit_pred8<PRED>(a.y, apply_unseen) * mul) >> 56) << 8) |
            ((uint32_t)((bit_pred8<PRED>(b.x, apply_unseen) * mul) >> 56) << 16) |
            ((uint32_t)((bit_pred8<PRED>(b.y, apply_unseen) * mul) >> 56) << 24);
    }
    ((uint32_t *)bits)[t] = w;
}

// any X: one wave per word
template <int PRED>
__global__ __launch_bounds__(256) void bit_pack_kernel(const uint8_t *__restrict__ state,
                                                       const BitGrid g, int apply_unseen,
                                                       unsigned long long *__restrict__ bits) {
    const size_t wv = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t nrows = (size_t)g.Y * g.Z;
    if (wv >= nrows * g.XW) return;
    const size_t row = wv / g.XW;
    const int x = (int)(wv % g.XW) * 64 + (threadIdx.x & 63);
    const bool o = (x < g.X) && bit_pred1<PRED>(state[row * g.X + x], apply_unseen);
    const unsigned long long b = __ballot(o);
    if ((threadIdx.x & 63) == 0) bits[wv] = b;
}

__device__ __forceinline__ unsigned long long bit_word(const unsigned long long *bits,
                                                       const BitGrid &g, int xw, int y, int z) {
    if (xw < 0 || xw >= g.XW || y < 0 || y >= g.Y || z < 0 || z >= g.Z) return 0ull;
    return bits[((size_t)z * g.Y + y) * g.XW + xw];
}

// surf = occupied and not inner (all six neighbours occupied; outside the planes
// held = empty).  occ holds g.Z planes, the first owned plane is plane `halo_lo`;
// out holds Zown planes.
__global__ __launch_bounds__(256) void bit_surface_kernel(const unsigned long long *__restrict__ occ,
                                                          const BitGrid g, int halo_lo, int Zown,
                                                          unsigned long long *__restrict__ out) {
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= (size_t)g.XW * g.Y * Zown) return;
    const int xw = (int)(w % g.XW);
    const size_t row = w / g.XW;
    const int y = (int)(row % g.Y), z = (int)(row / g.Y) + halo_lo;
    const unsigned long long c = bit_word(occ, g, xw, y, z);
    unsigned long long inner = c;
    if (c) {
        inner &= (c << 1) | (bit_word(occ, g, xw - 1, y, z) >> 63);
        inner &= (c >> 1) | (bit_word(occ, g, xw + 1, y, z) << 63);
        inner &= bit_word(occ, g, xw, y - 1, z) & bit_word(occ, g, xw, y + 1, z);
        inner &= bit_word(occ, g, xw, y, z - 1) & bit_word(occ, g, xw, y, z + 1);
    }
    out[w] = c & ~inner;
}

// box dilation, one axis per launch (r = radius, 1..4)
__global__ __launch_bounds__(256) void bit_dilate_x_kernel(const unsigned long long *__restrict__ in,
                                                           const BitGrid g, int r,
                                                           unsigned long long *__restrict__ out) {
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= (size_t)g.XW * g.Y * g.Z) return;
    const int xw = (int)(w % g.XW);
    const unsigned long long c = in[w];
    const unsigned long long L = xw > 0 ? in[w - 1] : 0ull, R = xw + 1 < g.XW ? in[w + 1] : 0ull;
    unsigned long long acc = c;
    for (int k = 1; k <= r; ++k) acc |= (c << k) | (c >> k) | (L >> (64 - k)) | (R << (64 - k));
    const int nbits = min(64, g.X - xw * 64);  // keep the padding clear
    if (nbits < 64) acc &= (1ull << nbits) - 1ull;
    out[w] = acc;
}

// axis = 1: rows y-r..y+r ; axis = 2: planes z-r..z+r.  With `minus` the result is
// and-ed with ~minus (the closure fills EMPTY voxels only).
__global__ __launch_bounds__(256) void bit_dilate_yz_kernel(const unsigned long long *__restrict__ in,
                                                            const BitGrid g, int r, int axis,
                                                            const unsigned long long *__restrict__ minus,
                                                            unsigned long long *__restrict__ out) {
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= (size_t)g.XW * g.Y * g.Z) return;
    const int xw = (int)(w % g.XW);
    const size_t row = w / g.XW;
    const int y = (int)(row % g.Y), z = (int)(row / g.Y);
    unsigned long long acc = 0ull;
    for (int k = -r; k <= r; ++k)
        acc |= (axis == 1) ? bit_word(in, g, xw, y + k, z) : bit_word(in, g, xw, y, z + k);
    if (minus) acc &= ~minus[w];
    out[w] = acc;
}

// ---- ordered compaction of the set bits ------------------------------------------------

constexpr int kBitChunk = 4096;  // words per workgroup

__global__ __launch_bounds__(256) void bit_count_kernel(const unsigned long long *__restrict__ bits,
                                                        size_t nwords, int *__restrict__ counts) {
    __shared__ int wsum[4];
    const size_t base = (size_t)blockIdx.x * kBitChunk;
    int mine = 0;
    for (int it = 0; it < kBitChunk / 256; ++it) {
        const size_t w = base + (size_t)it * 256 + threadIdx.x;
        if (w < nwords) mine += __popcll(bits[w]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// index[k] = flat index (x + X*row) of the k-th set bit, ascending
__global__ __launch_bounds__(256) void bit_write_kernel(const unsigned long long *__restrict__ bits,
                                                        size_t nwords, const BitGrid g,
                                                        const long long *__restrict__ offsets,
                                                        int *__restrict__ index) {
    __shared__ int wtot[4];
    const size_t base = (size_t)blockIdx.x * kBitChunk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long run = offsets[blockIdx.x];
    for (int it = 0; it < kBitChunk / 256; ++it) {
        const size_t w = base + (size_t)it * 256 + threadIdx.x;
        unsigned long long b = (w < nwords) ? bits[w] : 0ull;
        const int n = __popcll(b);
        int sc = n;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(sc, d);
            if (lane >= d) sc += t;
        }
        if (lane == 63) wtot[wave] = sc;
        __syncthreads();
        long long slot = run + sc - n;
        for (int v = 0; v < wave; ++v) slot += wtot[v];
        if (b) {
            const size_t row = w / g.XW;
            const int x0 = (int)(w % g.XW) * 64;
            const int first = (int)(row * g.X) + x0;
            while (b) {
                index[slot++] = first + (__ffsll((long long)b) - 1);
                b &= b - 1ull;
            }
        }
        run += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
}

}  // namespace arvx
