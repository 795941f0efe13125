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
// One lane owns one (x,y) column of cells and walks it along z, so the four corner
// bytes of a plane are read once and reused as the lower face of the next cell;
// lanes run along x (coalesced rows).  Column counts are written transposed
// (x-major), scanned, and the second walk writes each column's cells at its offset:
// the list comes out in the reference's emission order without a sort.
#pragma once

#include "arvx_device.h"

namespace arvx {

struct McParams {
    const uint8_t *state;  // planes ze0 .. ze1-1 of the grid
    int X, Y, Z;
    int ze0, ze1;
    int cz0, cz1;  // cells with z in [cz0, cz1) are listed by this context
};

// One (cx,cy) column of cells.  The four corner bytes of a plane sit at fixed
// offsets inside the plane; corners outside the grid (Model::get returns zero
// there, src/Model.h:119-122 => empty) are read from offset 0 and masked, so the
// loads carry no branches and kMcBatch planes are in flight at once -- the walk is
// a chain of dependent steps only through the 4 bits handed from cell to cell.
constexpr int kMcBatch = 16;

struct McColumn {
    unsigned off[4];   // (cx+1,cy) (cx,cy) (cx,cy+1) (cx+1,cy+1): the reference's corner order
    unsigned keep[4];  // 1 if that corner is inside the grid in x and y
};

__device__ __forceinline__ McColumn mc_column(const McParams &p, int cx, int cy) {
    const bool x0 = cx >= 0, x1 = cx + 1 < p.X, y0 = cy >= 0, y1 = cy + 1 < p.Y;
    const bool in[4] = {x1 && y0, x0 && y0, x0 && y1, x1 && y1};
    const int dx[4] = {1, 0, 0, 1}, dy[4] = {0, 0, 1, 1};
    McColumn c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        c.keep[k] = in[k] ? 1u : 0u;
        c.off[k] = in[k] ? (unsigned)((cy + dy[k]) * p.X + (cx + dx[k])) : 0u;
    }
    return c;
}

// bit set = corner EMPTY, for plane zg (any integer: planes outside the grid are empty)
__device__ __forceinline__ unsigned mc_face(const McParams &p, const McColumn &c, int zg) {
    const bool inz = zg >= 0 && zg < p.Z;
    const int zc = inz ? zg : p.ze0;  // any plane that is held
    const uint8_t *pl = p.state + (size_t)(zc - p.ze0) * p.X * p.Y;
    unsigned o = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) o |= (pl[c.off[k]] & c.keep[k]) << k;
    return inz ? (o ^ 0xFu) : 0xFu;
}

__device__ __forceinline__ bool mc_column_of_thread(const McParams &p, int &cx, int &cy) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cols = p.X + 1;
    if (t >= (long long)cols * (p.Y + 1)) return false;
    cy = (int)(t / cols) - 1;
    cx = (int)(t % cols) - 1;
    return true;
}

// walks the column and calls emit(cz, cubeIndex) for every cell that triangulates
template <class Emit>
__device__ __forceinline__ void mc_walk(const McParams &p, int cx, int cy, Emit emit) {
    const McColumn c = mc_column(p, cx, cy);
    unsigned lo = mc_face(p, c, p.cz0);
#pragma clang loop vectorize(disable) unroll(disable)
    for (int cz = p.cz0; cz < p.cz1; cz += kMcBatch) {
        unsigned f[kMcBatch];
#pragma unroll
        for (int k = 0; k < kMcBatch; ++k) f[k] = mc_face(p, c, min(cz + k, p.cz1 - 1) + 1);
#pragma unroll
        for (int k = 0; k < kMcBatch; ++k) {
            const unsigned idx = lo | (f[k] << 4);
            if (cz + k < p.cz1 && idx != 0u && idx != 255u) emit(cz + k, idx);
            lo = f[k];
        }
    }
}

__global__ __launch_bounds__(256) void mc_count_kernel(const McParams p, int *__restrict__ counts) {
    int cx, cy;
    if (!mc_column_of_thread(p, cx, cy)) return;
    int n = 0;
    mc_walk(p, cx, cy, [&](int, unsigned) { ++n; });
    counts[(size_t)(cx + 1) * (p.Y + 1) + (cy + 1)] = n;
}

// Exclusive scan of the column counts in three launches: every workgroup sums its
// kScanBlock entries; one workgroup scans those sums (block_off[nb] = total); every
// workgroup scans its own entries on top of its block offset.
constexpr int kScanPerThread = 16;
constexpr int kScanBlock = 256 * kScanPerThread;

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

__global__ __launch_bounds__(256) void mc_block_sum_kernel(const int *__restrict__ counts, int n,
                                                           int *__restrict__ block_sum) {
    __shared__ long long wtot[4];
    const int i0 = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    long long mine = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) mine += (i0 + k < n) ? counts[i0 + k] : 0;
    long long total;
    (void)wg_exclusive_scan(mine, wtot, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = (int)total;  // <= kScanBlock * (Z+1)
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

__global__ __launch_bounds__(256) void mc_block_scan_kernel(const int *__restrict__ counts, int n,
                                                            const long long *__restrict__ block_off,
                                                            long long *__restrict__ offsets) {
    __shared__ long long wtot[4];
    const int i0 = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    int v[kScanPerThread];
    long long mine = 0;
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        v[k] = (i0 + k < n) ? counts[i0 + k] : 0;
        mine += v[k];
    }
    long long total;
    long long pre = block_off[blockIdx.x] + wg_exclusive_scan(mine, wtot, &total);
#pragma unroll
    for (int k = 0; k < kScanPerThread; ++k) {
        if (i0 + k < n) offsets[i0 + k] = pre;
        pre += v[k];
    }
}

__global__ __launch_bounds__(256) void mc_write_kernel(const McParams p,
                                                       const long long *__restrict__ offsets,
                                                       int4 *__restrict__ cells) {
    int cx, cy;
    if (!mc_column_of_thread(p, cx, cy)) return;
    long long at = offsets[(size_t)(cx + 1) * (p.Y + 1) + (cy + 1)];
    mc_walk(p, cx, cy,
            [&](int cz, unsigned idx) { cells[at++] = make_int4(cx, cy, cz, (int)idx); });
}

}  // namespace arvx
