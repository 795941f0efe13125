// mc_kernels.h -- marching-cubes hand-off for gfx950: which cells produce
// triangles, with their cube index, in the order the reference visits them.
//
// Reference: marchingCubes() walks every cell (x,y,z) in [-1,X) x [-1,Y) x [-1,Z),
// x outermost and z innermost (src/MarchingCubes.cpp:12-18); ProcessVoxel reads the
// eight corners in the order of src/MarchingCubes.h:537-552, Polygonise sets bit i
// of the cube index when corner i has w < threshold (:479-484) and returns at once
// when edgeTable[index] == 0 (:486-488), which holds for index 0 and 255 only.
// With w in {0,1} and the threshold 0.5 every caller passes (src/main.cpp:303,
// src/VoxelCarving.cpp:67) "w < threshold" is "not occupied".
//
// The occupancy is first packed ALONG Z: one 64-bit word holds 64 consecutive planes
// of one voxel column (mc_zpack_rec_kernel: the state records are read once, whole records,
// and transposed through LDS).
// One lane then owns one (x,y) column of cells and walks it 64 cells at a time: the
// eight corner bits of 64 cells are eight words (four voxel columns, each shifted by
// 0 and 1 plane), "triangulates" is any & ~all, counting is a popcount.  Lanes run
// along x (coalesced rows).  Column counts are written transposed (x-major), scanned,
// and the second walk writes each column's cells at its offset: the list comes out
// in the reference's emission order without a sort.
#pragma once

#include "arvx/mc_tables.hpp"
#include "arvx_device.h"

namespace arvx {

// Bourke's triangle table as the kernels read it
struct McTriTable {
    alignas(16) int8_t e[256][16];  // edge numbers, three per triangle, -1 terminated
    int8_t n[256];      // triangles per cube index
};

constexpr McTriTable make_mc_tri_table() {
    McTriTable t{};
    for (int i = 0; i < 256; ++i) {
        int k = 0;
        for (const char *s = mc::kTriangles[i]; *s; ++s) t.e[i][k++] = (int8_t)mc::hex_digit(*s);
        t.n[i] = (int8_t)(k / 3);
        for (; k < 16; ++k) t.e[i][k] = -1;
    }
    return t;
}

__constant__ McTriTable kMcTri = make_mc_tri_table();


struct McParams {
    int X, Y, Z;
    int ze0, ze1;  // planes the context's records hold
    int cz0, cz1;  // cells with z in [cz0, cz1) are listed by this context
    // z-packed occupancy: word (w, y, x), bit b = voxel (x, y, cz0 + 64 w + b) occupied;
    // planes outside the grid are empty.  ZW words per column cover planes cz0 .. cz1.
    unsigned long long *zbits;
    int ZW;
};

// One workgroup per tile column (64 x 8 voxel columns) and word w: the occ halves of the (up
// to) nine tiles that the word's 64 planes touch go to LDS -- whole 128-byte loads, stored
// with the plane index innermost --, then every thread builds the words of two columns: per
// tile ONE 16-byte LDS read brings the entries of its eight planes, the column's bit of each
// is picked two planes at a time.  Lanes of a wave run along x: the stores are whole rows.
__global__ __launch_bounds__(256) void mc_zpack_rec_kernel(const CarveParams g, const McParams p) {
    // [tile along z][sub-tile][y (padded: 144-byte sub-tile stride, no bank conflicts)][plane]
    __shared__ __attribute__((aligned(16))) uint16_t occ[9][4][9][8];
    const int tx = blockIdx.x % g.tilesX, ty = (blockIdx.x / g.tilesX) % g.tilesY,
              w = blockIdx.x / (g.tilesX * g.tilesY);
    const int zbase = p.cz0 + 64 * w;  // global plane of bit 0
    const int lbase = zbase - p.ze0;   // ... as a local plane of the records (may be negative)
    const int tz_lo = (lbase >= 0 ? lbase : lbase - 7) / 8;  // floor
    const int off = lbase - 8 * tz_lo;                        // 0..7
    {
        const int sub = threadIdx.x >> 6, r = threadIdx.x & 63;
        uint16_t e[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int tz = tz_lo + j;
            // (a coarse tile that exists only as its code: arvx_device.h, lazy state.  Its records are
            // NOT read: requested together with the code -- to save the dependent round trip -- they
            // came cold from HBM, 29 -> 35 us)
            const int code = (tz >= 0 && tz < g.tilesZ) ? lazy_code(g, tx, ty, tz) : 0;
            e[j] = (tz >= 0 && tz < g.tilesZ)
                       ? (code ? (uint16_t)lazy_occ(g, code, tx, ty, tz, sub, r)
                               : g.rec[rec_index(g, tx, ty, tz, sub) * kRecU16 + r])
                       : (uint16_t)0;
        }
#pragma unroll
        for (int j = 0; j < 9; ++j) occ[j][sub][r & 7][r >> 3] = e[j];
    }
    __syncthreads();
    // planes of this word that exist for this context: inside the grid and not above cz1
    unsigned long long valid = ~0ull;
    {
        const int lo = zbase < 0 ? -zbase : 0;  // first valid bit
        const int top = (p.Z - 1 < p.cz1 ? p.Z - 1 : p.cz1) - zbase;  // last valid bit
        if (top < 0 || lo > 63) valid = 0ull;
        else {
            valid = (lo ? (~0ull << lo) : ~0ull);
            if (top < 63) valid &= (1ull << (top + 1)) - 1ull;
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = threadIdx.x + 256 * k;
        const int xl = c & 63, yl = c >> 6;
        const int x = tx * kTileX + xl, y = ty * kTileY + yl;
        if (x >= p.X || y >= p.Y) continue;
        const int sub = xl >> 4, bit = xl & 15;
        uint32_t B[9];  // the column's bit in the eight planes of tile j
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const uint4 q = *reinterpret_cast<const uint4 *>(&occ[j][sub][yl][0]);
            const uint32_t d[4] = {q.x, q.y, q.z, q.w};
            uint32_t byte = 0;
#pragma unroll
            for (int h = 0; h < 4; ++h) {  // dword h: planes 2h (low half) and 2h + 1 (high half)
                const uint32_t t = (d[h] >> bit) & 0x00010001u;
                byte |= ((t | (t >> 15)) & 3u) << (2 * h);
            }
            B[j] = byte;
        }
        const uint32_t lo32 = B[0] | (B[1] << 8) | (B[2] << 16) | (B[3] << 24);
        const uint32_t hi32 = B[4] | (B[5] << 8) | (B[6] << 16) | (B[7] << 24);
        unsigned long long word = ((unsigned long long)hi32 << 32) | lo32;  // planes 8 tz_lo + 0..63
        if (off) word = (word >> off) | ((unsigned long long)B[8] << (64 - off));
        p.zbits[((size_t)w * p.Y + y) * p.X + x] = word & valid;
    }
}

__device__ __forceinline__ bool mc_column_of_thread(const McParams &p, int &cx, int &cy) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cols = p.X + 1;
    if (t >= (long long)cols * (p.Y + 1)) return false;
    cy = (int)(t / cols) - 1;
    cx = (int)(t % cols) - 1;
    return true;
}

// walks the column and calls emit(cz, cubeIndex) for every cell that triangulates,
// cz ascending.  Corner order of the reference (src/MarchingCubes.h:537-552):
// (cx+1,cy) (cx,cy) (cx,cy+1) (cx+1,cy+1) on plane cz, then the same on plane cz+1.
// word by word: visit(w, act, lo, hi) -- bit b of act = cell p.cz0 + 64 w + b triangulates; lo[c] /
// hi[c]: corner column c on the cell's lower / upper plane
template <class Visit>
__device__ __forceinline__ void mc_walk_words(const McParams &p, int cx, int cy, Visit visit) {
    const bool x0 = cx >= 0, x1 = cx + 1 < p.X, y0 = cy >= 0, y1 = cy + 1 < p.Y;
    const bool in[4] = {x1 && y0, x0 && y0, x0 && y1, x1 && y1};
    const int dx[4] = {1, 0, 0, 1}, dy[4] = {0, 0, 1, 1};
    const unsigned long long *colp[4];
    const size_t wstride = (size_t)p.X * p.Y;
#pragma unroll
    for (int c = 0; c < 4; ++c)  // columns outside the grid read column 0 and are masked
        colp[c] = p.zbits + (in[c] ? (size_t)(cy + dy[c]) * p.X + (cx + dx[c]) : 0);
    const int ncell = p.cz1 - p.cz0;  // cells of this column; cell q uses planes q, q+1
    unsigned long long cur[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) cur[c] = in[c] ? colp[c][0] : 0ull;
#pragma clang loop vectorize(disable) unroll(disable)
    for (int w = 0; w * 64 < ncell; ++w) {
        unsigned long long nxt[4], lo[4], hi[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            nxt[c] = (in[c] && w + 1 < p.ZW) ? colp[c][(size_t)(w + 1) * wstride] : 0ull;
            lo[c] = cur[c];
            hi[c] = (cur[c] >> 1) | (nxt[c] << 63);
        }
        const unsigned long long any = lo[0] | lo[1] | lo[2] | lo[3] | hi[0] | hi[1] | hi[2] | hi[3];
        const unsigned long long all = lo[0] & lo[1] & lo[2] & lo[3] & hi[0] & hi[1] & hi[2] & hi[3];
        unsigned long long act = any & ~all;
        const int left = ncell - w * 64;
        if (left < 64) act &= (1ull << left) - 1ull;
        visit(w, act, lo, hi);
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = nxt[c];
    }
}
template <class Emit>
__device__ __forceinline__ void mc_walk(const McParams &p, int cx, int cy, Emit emit) {
    mc_walk_words(p, cx, cy, [&](int w, unsigned long long act, const unsigned long long *lo,
                                 const unsigned long long *hi) {
        while (act) {
            const int b = __ffsll((long long)act) - 1;
            act &= act - 1ull;
            unsigned idx = 0;  // bit i set = corner i EMPTY (src/MarchingCubes.h:479-484)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                idx |= (unsigned)((~lo[c] >> b) & 1ull) << c;
                idx |= (unsigned)((~hi[c] >> b) & 1ull) << (4 + c);
            }
            emit(p.cz0 + w * 64 + b, idx);
        }
    });
}

// the cells of a column that triangulate: the walk of mc_walk without visiting the cells
__global__ __launch_bounds__(256) void mc_count_kernel(const McParams p, int *__restrict__ counts) {
    int cx, cy;
    if (!mc_column_of_thread(p, cx, cy)) return;
    int n = 0;
    mc_walk_words(p, cx, cy, [&](int, unsigned long long act, const unsigned long long *, const unsigned long long *) {
        n += __popcll(act);
    });
    counts[(size_t)(cx + 1) * (p.Y + 1) + (cy + 1)] = n;
}

// (The columns' counts are scanned by scan2_lookback_kernel, bitplane_kernels.h; what follows serves
// the occupancy compression of exchange_kernels.h.)
// inclusive scan of one value per thread across the 256-thread workgroup;
// returns the exclusive prefix, *total gets the workgroup sum
__device__ __forceinline__ long long wg_exclusive_scan(long long mine, long long *wtot,
                                                       long long *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long sc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long t = __shfl_up(sc, d);
        if (lane >= d) sc += t;
    }
    if (lane == 63) wtot[wave] = sc;
    __syncthreads();
    long long pre = sc - mine;
    for (int w = 0; w < wave; ++w) pre += wtot[w];
    *total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    return pre;
}

// scans nb <= 2^31 block sums with ONE workgroup; block_off[nb] = total
__global__ __launch_bounds__(256) void mc_scan_blocks_kernel(const int *__restrict__ block_sum,
                                                             int nb,
                                                             long long *__restrict__ block_off) {
    __shared__ long long wtot[4];
    long long carry = 0;
    for (int b0 = 0; b0 < nb; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const long long v = (i < nb) ? block_sum[i] : 0;
        long long total;
        const long long pre = wg_exclusive_scan(v, wtot, &total);
        if (i < nb) block_off[i] = carry + pre;
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) block_off[nb] = carry;
}

// cap: room in `cells` (the list's length is not known when the buffers are sized: the caller
// reads the true total at its synchronisation and repeats the launch if it was larger)
__global__ __launch_bounds__(256) void mc_write_kernel(const McParams p,
                                                       const int *__restrict__ offsets,
                                                       long long cap, int4 *__restrict__ cells) {
    int cx, cy;
    if (!mc_column_of_thread(p, cx, cy)) return;
    long long at = offsets[(size_t)(cx + 1) * (p.Y + 1) + (cy + 1)];
    mc_walk(p, cx, cy, [&](int cz, unsigned idx) {
        if (at < cap) cells[at] = make_int4(cx, cy, cz, (int)idx);
        ++at;
    });
}

}  // namespace arvx
