// closure_kernels.h -- the reference's morphological "closure", applyClosure()
// (src/Postprocessing3d.cpp:4-100), on gfx950.
//
// What the reference computes: with thresh = 0 its erosion pass can
// never fail (it tests w < 0), so the whole function is ONE dilation with a
// (2r+1)^3 box: an empty voxel that has occupied voxels in its neighbourhood
// becomes their mean RGBA (sum in the order x-offset, y-offset, z-offset, all
// fp32, then an IEEE divide by float(count)); occupied voxels keep their value.
// The model a neighbour's colour is read from is the one main.cpp has at that
// point (src/main.cpp:291-299): carve -> colour pass -> handleUnseen.
//
// The voxels that get filled are found on bit planes (bitplane_kernels.h: pack
// "occupied", dilate along x, y, z, remove the occupied ones) and compacted in
// ascending index order; one thread per filled voxel then gathers the colours.
#pragma once

#include "arvx_device.h"
#include "color_kernels.h"

namespace arvx {

struct ClosureParams {
    const uint8_t *state;  // whole grid
    int X, Y, Z;
    int radius;        // (kernelSize - 1) / 2
    int apply_unseen;  // treat never-seen voxels as painted UNSEEN_COLOR (handleUnseen ran)
    const int *col_index;  // sparse colours, ascending flat index
    const float *col_rgb;
    const uint8_t *col_has;
    long long ncol;
    const int *row_start;  // [Y*Z + 1] first colour-list entry of each voxel row (x run)
};

// bit2 = painted with UNSEEN_COLOR by the host Model (include/arvx/model.hpp)
__device__ __forceinline__ bool cl_unseen(const ClosureParams &p, uint8_t st) {
    return (st & 4u) || (p.apply_unseen && !(st & 2u));
}
__device__ __forceinline__ bool cl_occupied(const ClosureParams &p, uint8_t st) {
    return (st & 1u) || cl_unseen(p, st);  // handleUnseen gives w = 1 (src/Model.cpp:42)
}

__device__ inline float4 cl_color(const ClosureParams &p, size_t i, uint8_t st) {
    if (cl_unseen(p, st)) return make_float4(204.f, 0.f, 0.f, 1.f);
    // the colour list is sorted by flat index, so one voxel row is one short run of it
    const size_t row = i / p.X;
    int lo = p.row_start[row], hi = p.row_start[row + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((size_t)p.col_index[mid] < i) lo = mid + 1; else hi = mid;
    }
    if (lo < p.row_start[row + 1] && (size_t)p.col_index[lo] == i && p.col_has[lo])
        return make_float4(p.col_rgb[3 * lo], p.col_rgb[3 * lo + 1], p.col_rgb[3 * lo + 2], 1.f);
    return make_float4(50.f, 168.f, 141.f, 1.f);
}

// row_start[r] = first list entry with index >= r*X  (r = 0..rows; row_start[rows] = ncol)
__global__ __launch_bounds__(256) void closure_rows_kernel(const int *__restrict__ col_index,
                                                           long long ncol, int X, long long rows,
                                                           int *__restrict__ row_start) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r > rows) return;
    const long long key = r * X;
    long long lo = 0, hi = ncol;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((long long)col_index[mid] < key) lo = mid + 1; else hi = mid;
    }
    row_start[r] = (int)lo;
}

// returns the number of occupied neighbours and their colour sum
__device__ inline int cl_gather(const ClosureParams &p, size_t i, float4 &sum) {
    const int x = (int)(i % p.X);
    const size_t t = i / p.X;
    const int y = (int)(t % p.Y), z = (int)(t / p.Y);
    int count = 0;
    sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = -p.radius; a <= p.radius; ++a) {  // src/Postprocessing3d.cpp:31-48
        const int xn = x + a;
        if (xn < 0 || xn >= p.X) continue;
        for (int b = -p.radius; b <= p.radius; ++b) {
            const int yn = y + b;
            if (yn < 0 || yn >= p.Y) continue;
            for (int c = -p.radius; c <= p.radius; ++c) {
                const int zn = z + c;
                if (zn < 0 || zn >= p.Z) continue;
                const size_t q = (size_t)xn + (size_t)p.X * ((size_t)yn + (size_t)p.Y * zn);
                const uint8_t st = p.state[q];
                if (!cl_occupied(p, st)) continue;
                ++count;
                const float4 v = cl_color(p, q, st);
                sum.x = sum.x + v.x;
                sum.y = sum.y + v.y;
                sum.z = sum.z + v.z;
                sum.w = sum.w + v.w;
            }
        }
    }
    return count;
}

// one thread per filled voxel (index ascending): mean RGBA of its occupied neighbours
__global__ __launch_bounds__(256) void closure_fill_kernel(const ClosureParams p,
                                                           const int *__restrict__ index,
                                                           long long n,
                                                           float4 *__restrict__ rgba) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float4 sum;
    const int count = cl_gather(p, (size_t)index[e], sum);
    const float fc = (float)count;  // Eigen `sum /= count`, src/Postprocessing3d.cpp:49-51
    rgba[e] = make_float4(sum.x / fc, sum.y / fc, sum.z / fc, sum.w / fc);
}

// the filled voxels are occupied from now on (their w is count/count = 1)
__global__ __launch_bounds__(256) void closure_mark_kernel(uint8_t *__restrict__ state,
                                                           const int *__restrict__ index,
                                                           long long n) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < n) state[index[e]] |= 1u;
}

__global__ __launch_bounds__(256) void export_overlay_kernel(const int *__restrict__ index,
                                                             const float4 *__restrict__ rgba,
                                                             long long first, long long last,
                                                             size_t i0, float4 *__restrict__ out) {
    const long long e = first + (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < last) out[(size_t)index[e] - i0] = rgba[e];
}

}  // namespace arvx
