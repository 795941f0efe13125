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

// one workgroup per tile, one wave per sub-tile, lane = entry r = (z & 7) * 8 + (y & 7).
// `bytes` holds the context's planes (plane 0 = local plane 0); only local planes
// [zl0, zl0 + nz) are converted (an upload of the owned planes leaves the halo planes alone,
// arvx_state_upload_halo converts one plane).  bit2 of a byte (painted UNSEEN_COLOR by a host
// Model) has no place in a record: it goes to the `paint` bit plane (layout of
// bitplane_kernels.h over the context's planes, XW = ceil(X / 64) words per row), and *any
// is set when one was found.
__global__ __launch_bounds__(256) void rec_from_bytes_kernel(const CarveParams p,
                                                             const uint8_t *__restrict__ bytes,
                                                             int zl0, int nz,
                                                             unsigned long long *__restrict__ paint,
                                                             int *__restrict__ any) {
    const int tx = blockIdx.x % p.tilesX, ty = (blockIdx.x / p.tilesX) % p.tilesY,
              tz = blockIdx.x / (p.tilesX * p.tilesY);
    const int wave = threadIdx.x >> 6, r = threadIdx.x & 63;
    const int x0 = tx * kTileX + wave * kSubX, y = ty * kTileY + (r & 7), z = tz * kTileZ + (r >> 3);
    if (y >= p.Y || z < zl0 || z >= zl0 + nz || z >= p.Z) return;  // (outside the grid: kept "finished")
    uint32_t occ = 0, seen = 0xffffu, pnt = 0;
    if (x0 < p.X) {
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
            pnt |= (uint32_t)((b[j] >> 2) & 1u) << j;
        }
        uint16_t *rec = p.rec + rec_index(p, tx, ty, tz, wave) * kRecU16;
        rec[r] = (uint16_t)occ;
        rec[64 + r] = (uint16_t)seen;
        if (pnt) atomicOr(any, 1);
    }
    // the sub-tile's 16 bits of the row's word tx (zeros behind the end of the row)
    const int XW = (p.X + 63) >> 6;
    reinterpret_cast<uint16_t *>(paint + ((size_t)z * p.Y + y) * XW + tx)[wave] = (uint16_t)pnt;
}

// records (+ paint plane, may be null) -> bytes, local planes [zl0, zl0 + nz)
__global__ __launch_bounds__(256) void rec_to_bytes_kernel(const CarveParams p,
                                                           uint8_t *__restrict__ bytes, int zl0,
                                                           int nz,
                                                           const unsigned long long *__restrict__ paint) {
    const int tx = blockIdx.x % p.tilesX, ty = (blockIdx.x / p.tilesX) % p.tilesY,
              tz = blockIdx.x / (p.tilesX * p.tilesY);
    const int wave = threadIdx.x >> 6, r = threadIdx.x & 63;
    const int x0 = tx * kTileX + wave * kSubX, y = ty * kTileY + (r & 7), z = tz * kTileZ + (r >> 3);
    if (x0 >= p.X || y >= p.Y || z >= p.Z || z < zl0 || z >= zl0 + nz) return;
    const uint16_t *rec = p.rec + rec_index(p, tx, ty, tz, wave) * kRecU16;
    const int code = lazy_code(p, tx, ty, tz);
    const uint32_t occ = code ? lazy_occ(p, code, tx, ty, tz, wave, r) : rec[r];
    const uint32_t seen = code ? lazy_seen(p, code, tx, ty, tz, wave, r) : rec[64 + r];
    uint32_t pnt = 0;
    if (paint) {
        const int XW = (p.X + 63) >> 6;
        pnt = reinterpret_cast<const uint16_t *>(paint + ((size_t)z * p.Y + y) * XW + tx)[wave];
    }
    uint8_t b[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
        b[j] = (uint8_t)(((occ >> j) & 1u) | (((seen >> j) & 1u) << 1) | (((pnt >> j) & 1u) << 2));
    uint8_t *dst = bytes + ((size_t)z * p.Y + y) * p.X + x0;
    if ((p.X & 15) == 0 && ((uintptr_t)bytes & 15u) == 0) {
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(b);
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (x0 + j < p.X) dst[j] = b[j];
    }
}

// one voxel's state bits (bit0 occupied, bit1 seen) straight from its record; lz = local plane
__device__ __forceinline__ uint32_t rec_state(const CarveParams &p, int x, int y, int lz) {
    const uint16_t *rec = p.rec + rec_index(p, x >> 6, y >> 3, lz >> 3, (x >> 4) & 3) * kRecU16;
    const int r = (lz & 7) * 8 + (y & 7), b = x & 15;
    return ((rec[r] >> b) & 1u) | (((rec[64 + r] >> b) & 1u) << 1);
}
__device__ __forceinline__ bool rec_seen(const CarveParams &p, int x, int y, int lz) {
    const int code = lazy_code(p, x >> 6, y >> 3, lz >> 3);
    if (code)  // (a coarse tile that exists only as its code, arvx_device.h)
        return (lazy_seen(p, code, x >> 6, y >> 3, lz >> 3, (x >> 4) & 3, (lz & 7) * 8 + (y & 7)) >> (x & 15)) & 1u;
    const uint16_t *rec = p.rec + rec_index(p, x >> 6, y >> 3, lz >> 3, (x >> 4) & 3) * kRecU16;
    return (rec[64 + (lz & 7) * 8 + (y & 7)] >> (x & 15)) & 1u;
}
// one bit of a plane in the layout of bitplane_kernels.h (null plane: 0)
__device__ __forceinline__ uint32_t plane_bit(const unsigned long long *__restrict__ bits, int X,
                                              int Y, int x, int y, int lz) {
    if (!bits) return 0u;
    return (uint32_t)((bits[((size_t)lz * Y + y) * ((X + 63) >> 6) + (x >> 6)] >> (x & 63)) & 1ull);
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
    const int code = lazy_code(p, k >> 1, y >> 3, z >> 3);
    const uint32_t w = code ? (lazy_occ(p, code, k >> 1, y >> 3, z >> 3, (k & 1) * 2, r) |
                               (lazy_occ(p, code, k >> 1, y >> 3, z >> 3, (k & 1) * 2 + 1, r) << 16))
                            : ((uint32_t)rec[r] | ((uint32_t)rec[kRecU16 + r] << 16));
    const size_t zo = global ? (size_t)global_z(p, z) : (size_t)zi;
    out[(zo * p.Y + y) * wpr + k] = w;
}

// X % 64 == 0, 8-byte aligned output: one WAVE per tile of 64 x 8 x 8 voxels, lane = one of the
// tile's 64 rows (y, z).  The four sub-tiles' occupancy entries of a row are four 2-byte loads
// that a wave issues as four contiguous 128-byte pieces (the kernel above reads the same bytes
// two per thread, 256 bytes apart, and does the tile's index arithmetic once per output word);
// a lane writes its row's 64 voxels as one 8-byte word; four tiles along x per workgroup, so
// that their pieces of a row meet in one line.  A tile of a settled coarse tile (lazy code) is
// not read at all.
__global__ __launch_bounds__(256) void pack_occupancy_tile_kernel(const CarveParams p, int zl0,
                                                                  int nz, int global,
                                                                  unsigned long long *__restrict__ out) {
    const int tz0 = zl0 >> 3, ntz = ((zl0 + nz - 1) >> 3) - tz0 + 1;
    const long long t = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= (long long)p.tilesX * p.tilesY * ntz) return;
    const int tx = (int)(t % p.tilesX), ty = (int)((t / p.tilesX) % p.tilesY);
    const int tz = tz0 + (int)(t / ((long long)p.tilesX * p.tilesY));
    const int lane = threadIdx.x & 63;  // row r = zrow * 8 + yrow, as in the records
    const int y = ty * kTileY + (lane & 7), z = tz * kTileZ + (lane >> 3);
    const int code = lazy_code(p, tx, ty, tz);  // (wave-uniform)
    unsigned long long w = 0;
#pragma unroll
    for (int sw = 0; sw < 4; ++sw) {
        uint32_t e;
        if (code) {
            e = lazy_occ(p, code, tx, ty, tz, sw, lane);
        } else {
            e = p.rec[rec_index(p, tx, ty, tz, sw) * kRecU16 + lane];
        }
        w |= (unsigned long long)(e & 0xffffu) << (16 * sw);
    }
    if (y >= p.Y || z < zl0 || z >= zl0 + nz) return;
    const size_t zo = global ? (size_t)global_z(p, z) : (size_t)(z - zl0);
    out[(zo * p.Y + y) * (size_t)(p.X >> 6) + tx] = w;
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
        const int code = lazy_code(p, x0 >> 6, y >> 3, z >> 3);
        const uint32_t e = code ? lazy_occ(p, code, x0 >> 6, y >> 3, z >> 3, (x0 >> 4) & 3,
                                           (z & 7) * 8 + (y & 7))
                                : rec[(z & 7) * 8 + (y & 7)];
        word |= ((e >> (x0 & 15)) & 0xffu) << (8 * k);
        if (k == 0) dst = global ? ((size_t)global_z(p, z) * plane_bytes + in_plane) : b;
    }
    // (global: planes start on word boundaries because X * Y % 64 == 0, and the four bytes of
    // a word never straddle a plane for the same reason)
    out[dst / 4] = word;
}

// Records -> bit planes in the layout of bitplane_kernels.h (rows padded to 64-bit words,
// XW = ceil(X / 64) words per row): local planes [zl0, zl0 + nz).
//   occ_out     occupied; with closure_occupied: what the closure calls occupied -- occupied,
//               or painted UNSEEN_COLOR by the host (`paint`, may be null; indexed by the
//               context's planes), or, with apply_unseen, never seen (handleUnseen gives
//               those w = 1, src/Model.cpp:42)
//   unseen_out  (may be null) the voxels whose colour is UNSEEN_COLOR: painted, or with
//               apply_unseen never seen
__global__ __launch_bounds__(256) void bitgrid_from_rec_kernel(
    const CarveParams p, int zl0, int nz, int closure_occupied, int apply_unseen,
    const unsigned long long *__restrict__ paint, unsigned long long *__restrict__ occ_out,
    unsigned long long *__restrict__ unseen_out) {
    const int XW = (p.X + 63) >> 6;
    const size_t n = (size_t)XW * p.Y * nz;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int xw = (int)(i % XW), y = (int)((i / XW) % p.Y), zi = (int)(i / ((size_t)XW * p.Y));
    const int z = zl0 + zi, r = (z & 7) * 8 + (y & 7);
    const uint16_t *rec = p.rec + rec_index(p, xw, y >> 3, z >> 3, 0) * kRecU16;
    const int code = lazy_code(p, xw, y >> 3, z >> 3);
    unsigned long long o = 0, u = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // the tile's four sub-tiles: 16 voxels each
        if (64 * xw + 16 * k >= p.X) break;
        const uint32_t eo = code ? lazy_occ(p, code, xw, y >> 3, z >> 3, k, r) : rec[k * kRecU16 + r];
        const uint32_t es = code ? lazy_seen(p, code, xw, y >> 3, z >> 3, k, r)
                                 : rec[k * kRecU16 + 64 + r];
        o |= (unsigned long long)eo << (16 * k);
        if (apply_unseen) u |= (unsigned long long)(uint16_t)~es << (16 * k);
    }
    if (paint) u |= paint[((size_t)z * p.Y + y) * XW + xw];
    if (closure_occupied) o |= u;
    const int nx = p.X - 64 * xw;  // voxels behind the end of the row: zero
    if (nx < 64) {
        o &= (1ull << nx) - 1ull;
        u &= (1ull << nx) - 1ull;
    }
    occ_out[i] = o;
    if (unseen_out) unseen_out[i] = u;
}

// occ |= bits for local planes [zl0, zl0 + nz): the closure's filled voxels become occupied
// (bits: layout of bitplane_kernels.h over those planes)
__global__ __launch_bounds__(256) void rec_or_bitgrid_kernel(const CarveParams p, int zl0, int nz,
                                                             const unsigned long long *__restrict__ bits) {
    const int XW = (p.X + 63) >> 6;
    const size_t n = (size_t)XW * p.Y * nz;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned long long w = bits[i];
    if (!w) return;
    const int xw = (int)(i % XW), y = (int)((i / XW) % p.Y), zi = (int)(i / ((size_t)XW * p.Y));
    const int z = zl0 + zi, r = (z & 7) * 8 + (y & 7);
    uint16_t *rec = p.rec + rec_index(p, xw, y >> 3, z >> 3, 0) * kRecU16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint16_t e = (uint16_t)(w >> (16 * k));
        if (e) rec[k * kRecU16 + r] |= e;
    }
}

// rec_or_bitgrid_kernel where coarse tiles may exist only as their code (lazy state, arvx_device.h;
// Y and the planes held are multiples of 8: every row of a tile has a word of the plane).  The
// plane's producer has marked the coded tiles that receive a bit (bit 7 of the code,
// bit_dilate_z_count_kernel): every word's thread of such a tile writes its row's eight entries --
// the code's constants with the bits on top --, and from this launch on the marked code reads as
// "records hold the state" (lazy_code).  Tiles without a code get the bits OR-ed in as above; coded
// tiles without a mark receive nothing and stay codes.
__global__ __launch_bounds__(256) void rec_or_bitgrid_lazy_kernel(const CarveParams p, int nz,
                                                                  const unsigned long long *__restrict__ bits,
                                                                  const uint8_t *__restrict__ ccode) {
    const int XW = (p.X + 63) >> 6;
    const size_t n = (size_t)XW * p.Y * nz;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int xw = (int)(i % XW), y = (int)((i / XW) % p.Y), z = (int)(i / ((size_t)XW * p.Y));
    const int c = (int)ccode[coarse_of(p, xw, y >> 3, z >> 3)];
    const unsigned long long w = bits[i];
    const int r = (z & 7) * 8 + (y & 7);
    uint16_t *rec = p.rec + rec_index(p, xw, y >> 3, z >> 3, 0) * kRecU16;
    if (!(c & 0x7f)) {  // records hold the state
        if (!w) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint16_t e = (uint16_t)(w >> (16 * k));
            if (e) rec[k * kRecU16 + r] |= e;
        }
    } else if (c & kCodeWritten) {  // a coded tile that receives voxels: written out, row by row
        const int code = c & 0x7f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            rec[k * kRecU16 + r] =
                (uint16_t)(lazy_occ(p, code, xw, y >> 3, z >> 3, k, r) | (uint32_t)(uint16_t)(w >> (16 * k)));
            rec[k * kRecU16 + 64 + r] = (uint16_t)lazy_seen(p, code, xw, y >> 3, z >> 3, k, r);
        }
    }
}

// Model::handleUnseen on records: occ |= ~seen (voxels outside the grid are kept "seen").  The
// records of a coarse tile that exists only as its code (ccode, may be null; 2^rec_shift records
// per coarse tile) are nobody's to read: skipped -- "carved and seen", "untouched and seen" and
// "untouched, not seen" all stay what they are under occ |= ~seen.
__global__ __launch_bounds__(256) void rec_handle_unseen_kernel(uint32_t *__restrict__ rec32, size_t nrec,
                                                                const uint8_t *__restrict__ ccode,
                                                                int rec_shift) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;  // record t / 32, word t % 32
    if (t >= nrec * 32) return;
    if (ccode) {
        const int c = ccode[(t >> 5) >> rec_shift];
        if (c && !(c & kCodeWritten)) return;
    }
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
    const int code = lazy_code(p, k >> 1, y >> 3, z >> 3);
    const int w0 = (k & 1) * 2;
    uint32_t o = code ? lazy_occ(p, code, k >> 1, y >> 3, z >> 3, w0, r) : rec[r];
    uint32_t sn = code ? lazy_seen(p, code, k >> 1, y >> 3, z >> 3, w0, r) : rec[64 + r];
    if (32 * k + 16 < p.X) {  // the word's upper half lies in the next sub-tile
        o |= (code ? lazy_occ(p, code, k >> 1, y >> 3, z >> 3, w0 + 1, r) : (uint32_t)rec[kRecU16 + r]) << 16;
        sn |= (code ? lazy_seen(p, code, k >> 1, y >> 3, z >> 3, w0 + 1, r)
                    : (uint32_t)rec[kRecU16 + 64 + r]) << 16;
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

// Model::voxels for owned voxels [i0, i0+n) (flat index over the owned planes; zown = local
// plane of owned plane 0): MODEL_COLOR where occupied, zero where carved (reference
// src/Model.cpp:9-14, src/VoxelCarving.cpp:52), UNSEEN_COLOR where the host painted (`paint`,
// may be null) and, with apply_unseen, where never seen (src/Model.cpp:36-47).
__global__ __launch_bounds__(256) void export_fill_kernel(const CarveParams p, int zown, size_t i0,
                                                          size_t n,
                                                          const unsigned long long *__restrict__ paint,
                                                          float4 *__restrict__ out,
                                                          int apply_unseen) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n; k += stride) {
        const size_t i = i0 + k;
        const int x = (int)(i % p.X), y = (int)((i / p.X) % p.Y),
                  lz = zown + (int)(i / ((size_t)p.X * p.Y));
        const uint32_t st = rec_state(p, x, y, lz);
        float4 v = (st & 1u) ? make_float4(50.f, 168.f, 141.f, 1.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (plane_bit(paint, p.X, p.Y, x, y, lz) || (apply_unseen && !(st & 2u)))
            v = make_float4(204.f, 0.f, 0.f, 1.f);
        out[k] = v;
    }
}

// an uploaded colour list (3 floats per voxel) in the device's form (r, g, b, has = 1)
__global__ __launch_bounds__(256) void rgb_to_rgba_kernel(const float *__restrict__ rgb, long long n,
                                                          float4 *__restrict__ rgba) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k < n) rgba[k] = make_float4(rgb[3 * k], rgb[3 * k + 1], rgb[3 * k + 2], 1.f);
}

__global__ __launch_bounds__(256) void export_scatter_kernel(
    const int *__restrict__ index, const float4 *__restrict__ rgba, long long first,
    long long last, const CarveParams p, int zown,
    const unsigned long long *__restrict__ paint, size_t i0, float4 *__restrict__ out,
    int apply_unseen) {
    const long long e = first + (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= last) return;
    const float4 c = rgba[e];
    if (c.w == 0.f) return;  // no sample: the voxel keeps what the state says
    // (list indices run over the context's planes; `out` over the owned ones)
    const size_t j = (size_t)index[e];
    const int x = (int)(j % p.X), y = (int)((j / p.X) % p.Y), lz = (int)(j / ((size_t)p.X * p.Y));
    const size_t i = j - (size_t)zown * p.X * p.Y;
    // handleUnseen runs after colouring
    if (plane_bit(paint, p.X, p.Y, x, y, lz) || (apply_unseen && !(rec_state(p, x, y, lz) & 2u)))
        return;
    out[i - i0] = make_float4(c.x, c.y, c.z, 1.f);
}

}  // namespace arvx
