// state_kernels.h -- conversions between the device's state records (arvx_device.h: 2 bits per
// voxel, one 256-byte record per 16 x 8 x 8 sub-tile) and the forms the C-ABI and the other
// stages exchange: the one-byte-per-voxel plane (bit0 occupied, bit1 seen; arvx_state_upload /
// _download / _device_ptr) and the flat packed occupancy (voxel i -> bit i % 32 of word i / 32;
// arvx_pack_occupancy[_global]).
#pragma once

#include "arvx_device.h"
#include "carve_kernels.h"

namespace arvx {

// every record "finished" (occ 0, seen 1): what the voxels outside the grid keep for ever
__global__ __launch_bounds__(256) void rec_init_kernel(uint32_t *__restrict__ rec32, size_t nwords) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride)
        rec32[i] = (i & 32) ? 0xffffffffu : 0u;  // 32 words occ, 32 words seen
}

// one workgroup per tile, one wave per sub-tile, lane = entry r = (z & 7) * 8 + (y & 7)
__global__ __launch_bounds__(256) void rec_from_bytes_kernel(const CarveParams p,
                                                             const uint8_t *__restrict__ bytes) {
    const int tx = blockIdx.x % p.tilesX, ty = (blockIdx.x / p.tilesX) % p.tilesY,
              tz = blockIdx.x / (p.tilesX * p.tilesY);
    const int wave = threadIdx.x >> 6, r = threadIdx.x & 63;
    const int x0 = tx * kTileX + wave * kSubX, y = ty * kTileY + (r & 7), z = tz * kTileZ + (r >> 3);
    uint32_t occ = 0, seen = 0xffffu;
    if (x0 < p.X && y < p.Y && z < p.Z) {
        const uint8_t *src = bytes + ((size_t)z * p.Y + y) * p.X + x0;
        uint8_t b[16];
        if ((p.X & 15) == 0 && ((uintptr_t)bytes & 15u) == 0) {
            *reinterpret_cast<uint4 *>(b) = *reinterpret_cast<const uint4 *>(src);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) b[j] = (x0 + j < p.X) ? src[j] : (uint8_t)2;
        }
        seen = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            occ |= (uint32_t)(b[j] & 1u) << j;
            seen |= (uint32_t)((b[j] >> 1) & 1u) << j;
        }
    }
    uint16_t *rec = p.rec + rec_index(p, tx, ty, tz, wave) * kRecU16;
    rec[r] = (uint16_t)occ;
    rec[64 + r] = (uint16_t)seen;
}

__global__ __launch_bounds__(256) void rec_to_bytes_kernel(const CarveParams p,
                                                           uint8_t *__restrict__ bytes) {
    const int tx = blockIdx.x % p.tilesX, ty = (blockIdx.x / p.tilesX) % p.tilesY,
              tz = blockIdx.x / (p.tilesX * p.tilesY);
    const int wave = threadIdx.x >> 6, r = threadIdx.x & 63;
    const int x0 = tx * kTileX + wave * kSubX, y = ty * kTileY + (r & 7), z = tz * kTileZ + (r >> 3);
    if (x0 >= p.X || y >= p.Y || z >= p.Z) return;
    const uint16_t *rec = p.rec + rec_index(p, tx, ty, tz, wave) * kRecU16;
    const uint32_t occ = rec[r], seen = rec[64 + r];
    uint8_t b[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) b[j] = (uint8_t)(((occ >> j) & 1u) | (((seen >> j) & 1u) << 1));
    uint8_t *dst = bytes + ((size_t)z * p.Y + y) * p.X + x0;
    if ((p.X & 15) == 0 && ((uintptr_t)bytes & 15u) == 0) {
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(b);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (x0 + j < p.X) dst[j] = b[j];
    }
}

// Flat packed occupancy of local planes [zl0, zl0 + nz) from the records; X % 32 == 0.
// Output word (z, y, k) covers x = 32 k .. 32 k + 31: two neighbouring sub-tiles' entries.
// global: the plane goes to its place in the WHOLE grid's word plane (global_z), else planes
// are written back to back starting at word 0.
__global__ __launch_bounds__(256) void pack_occupancy_rec_kernel(const CarveParams p, int zl0,
                                                                 int nz, int global,
                                                                 uint32_t *__restrict__ out) {
    const int wpr = p.X >> 5;  // words per row
    const size_t n = (size_t)wpr * p.Y * nz;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = (int)(i % wpr), y = (int)((i / wpr) % p.Y), zi = (int)(i / ((size_t)wpr * p.Y));
    const int z = zl0 + zi;
    const int r = (z & 7) * 8 + (y & 7);
    const uint16_t *rec = p.rec + rec_index(p, k >> 1, y >> 3, z >> 3, (k & 1) * 2) * kRecU16;
    const uint32_t w = (uint32_t)rec[r] | ((uint32_t)rec[kRecU16 + r] << 16);
    const size_t zo = global ? (size_t)global_z(p, z) : (size_t)zi;
    out[(zo * p.Y + y) * wpr + k] = w;
}

// The same for any X % 8 == 0 (e.g. the 648- and 816-wide grids of the 2- and 4-GPU weak
// scaling): a byte of the flat packing holds eight voxels of ONE row; a thread assembles one
// output word from its four bytes.  n_bytes = X * Y * nz / 8 (a multiple of 4 is not needed:
// the last word is padded with zeros).
__global__ __launch_bounds__(256) void pack_occupancy_rec8_kernel(const CarveParams p, int zl0,
                                                                  int nz, int global,
                                                                  uint32_t *__restrict__ out) {
    const size_t plane_bytes = (size_t)p.X * p.Y / 8;  // X % 8 == 0
    const size_t n_bytes = plane_bytes * nz;
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (w * 4 >= n_bytes) return;
    uint32_t word = 0;
    size_t dst = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t b = w * 4 + k;
        if (b >= n_bytes) break;
        const int zi = (int)(b / plane_bytes);
        const size_t in_plane = b % plane_bytes;
        const int y = (int)(in_plane / (p.X / 8)), x0 = (int)(in_plane % (p.X / 8)) * 8;
        const int z = zl0 + zi;
        const uint16_t *rec =
            p.rec + rec_index(p, x0 >> 6, y >> 3, z >> 3, (x0 >> 4) & 3) * kRecU16;
        const uint32_t e = rec[(z & 7) * 8 + (y & 7)];
        word |= ((e >> (x0 & 15)) & 0xffu) << (8 * k);
        if (k == 0) dst = global ? ((size_t)global_z(p, z) * plane_bytes + in_plane) : b;
    }
    // (global: planes start on word boundaries because X * Y % 64 == 0, and the four bytes of
    // a word never straddle a plane for the same reason)
    out[dst / 4] = word;
}

// Records -> one bit plane in the layout of bitplane_kernels.h (rows padded to 64-bit words,
// XW = ceil(X / 64) words per row): local planes [zl0, zl0 + nz).  closure_occupied: occupied
// OR never seen (what the closure calls occupied after handleUnseen), else occupied.
__global__ __launch_bounds__(256) void bitgrid_from_rec_kernel(const CarveParams p, int zl0, int nz,
                                                               int closure_occupied,
                                                               unsigned long long *__restrict__ bits) {
    const int XW = (p.X + 63) >> 6;
    const size_t n = (size_t)XW * p.Y * nz;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int xw = (int)(i % XW), y = (int)((i / XW) % p.Y), zi = (int)(i / ((size_t)XW * p.Y));
    const int z = zl0 + zi, r = (z & 7) * 8 + (y & 7);
    const uint16_t *rec = p.rec + rec_index(p, xw, y >> 3, z >> 3, 0) * kRecU16;
    unsigned long long w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // the tile's four sub-tiles: 16 voxels each
        if (64 * xw + 16 * k >= p.X) break;
        unsigned long long e = rec[k * kRecU16 + r];
        if (closure_occupied) e |= (unsigned long long)(uint16_t)~rec[k * kRecU16 + 64 + r];
        w |= e << (16 * k);
    }
    const int nx = p.X - 64 * xw;  // voxels behind the end of the row: zero
    if (nx < 64) w &= (1ull << nx) - 1ull;
    bits[i] = w;
}

// Model::handleUnseen on records: occ |= ~seen (voxels outside the grid are kept "seen")
__global__ __launch_bounds__(256) void rec_handle_unseen_kernel(uint32_t *__restrict__ rec32,
                                                                size_t nrec) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;  // record t / 32, word t % 32
    if (t >= nrec * 32) return;
    uint32_t *r = rec32 + (t >> 5) * 64 + (t & 31);
    r[0] |= ~r[32];
}

// Bit planes with rows padded to whole 32-bit words (wpr = ceil(X / 32) words per row; word
// (z, y, k) bit j = voxel x = 32 k + j): the form a host Model keeps (include/arvx/model.hpp).
// Planes [zl0, zl0 + nz) of the records <-> plane words starting at word 0.
__global__ __launch_bounds__(256) void planes_from_rec_kernel(const CarveParams p, int zl0, int nz,
                                                              uint32_t *__restrict__ occ,
                                                              uint32_t *__restrict__ seen) {
    const int wpr = (p.X + 31) >> 5;
    const size_t n = (size_t)wpr * p.Y * nz;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = (int)(i % wpr), y = (int)((i / wpr) % p.Y), zi = (int)(i / ((size_t)wpr * p.Y));
    const int z = zl0 + zi, r = (z & 7) * 8 + (y & 7);
    const uint16_t *rec = p.rec + rec_index(p, k >> 1, y >> 3, z >> 3, (k & 1) * 2) * kRecU16;
    uint32_t o = rec[r], sn = rec[64 + r];
    if (32 * k + 16 < p.X) {  // the word's upper half lies in the next sub-tile
        o |= (uint32_t)rec[kRecU16 + r] << 16;
        sn |= (uint32_t)rec[kRecU16 + 64 + r] << 16;
    }
    // voxels behind the end of the row: records hold (occ 0, seen 1) there, the planes zeros
    const int nx = p.X - 32 * k;
    const uint32_t valid = nx >= 32 ? 0xffffffffu : ((1u << nx) - 1u);
    occ[i] = o & valid;
    seen[i] = sn & valid;
}

__global__ __launch_bounds__(256) void rec_from_planes_kernel(const CarveParams p, int zl0, int nz,
                                                              const uint32_t *__restrict__ occ,
                                                              const uint32_t *__restrict__ seen) {
    const int wpr = (p.X + 31) >> 5;
    const size_t n = (size_t)wpr * p.Y * nz;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = (int)(i % wpr), y = (int)((i / wpr) % p.Y), zi = (int)(i / ((size_t)wpr * p.Y));
    const int z = zl0 + zi, r = (z & 7) * 8 + (y & 7);
    uint16_t *rec = p.rec + rec_index(p, k >> 1, y >> 3, z >> 3, (k & 1) * 2) * kRecU16;
    const int nx = p.X - 32 * k;
    const uint32_t valid = nx >= 32 ? 0xffffffffu : ((1u << nx) - 1u);
    const uint32_t o = occ[i] & valid, sn = seen[i] | ~valid;  // outside the grid: occ 0, seen 1
    rec[r] = (uint16_t)o;
    rec[64 + r] = (uint16_t)sn;
    if (32 * k + 16 < p.X) {
        rec[kRecU16 + r] = (uint16_t)(o >> 16);
        rec[kRecU16 + 64 + r] = (uint16_t)(sn >> 16);
    }
}

}  // namespace arvx
