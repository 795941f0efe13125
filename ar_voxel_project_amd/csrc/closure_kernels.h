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
// Everything runs on bit planes built from the state records (bitplane_kernels.h,
// state_kernels.h): "occupied" as the closure sees it, dilated along x, y, z, minus itself =
// the voxels that get filled, compacted in ascending index order; one thread per filled voxel
// then gathers its neighbours' colours: occupancy and UNSEEN paint are bit tests, a
// neighbour's explicit colour is one rank lookup in the colour pass's sparse list.
#pragma once

#include "arvx_device.h"
#include "bitplane_kernels.h"
#include "color_kernels.h"

namespace arvx {

struct ClosureParams {
    BitGrid g;                         // the whole grid
    const unsigned long long *occ;     // what the closure calls occupied
    const unsigned long long *unseen;  // voxels whose colour is UNSEEN_COLOR (null: none)
    int radius;                        // (kernelSize - 1) / 2
    SparseList col;                    // the colour pass's list: plane + rank ...
    const float *col_rgb;              // ... rgb and has-sample flag per entry
    const uint8_t *col_has;
};

// colour of an occupied voxel as Model::get returns it at src/main.cpp:297
__device__ inline float4 cl_color(const ClosureParams &p, int x, size_t row) {
    if (p.unseen && ((p.unseen[row * p.g.XW + (x >> 6)] >> (x & 63)) & 1ull))
        return make_float4(204.f, 0.f, 0.f, 1.f);
    const int k = sparse_find(p.col, p.g.XW, x, row);
    if (k >= 0 && p.col_has[k])
        return make_float4(p.col_rgb[3 * k], p.col_rgb[3 * k + 1], p.col_rgb[3 * k + 2], 1.f);
    return make_float4(50.f, 168.f, 141.f, 1.f);
}

// returns the number of occupied neighbours and their colour sum
__device__ inline int cl_gather(const ClosureParams &p, size_t i, float4 &sum) {
    const int X = p.g.X, Y = p.g.Y, Z = p.g.Z;
    const int x = (int)(i % X);
    const size_t t = i / X;
    const int y = (int)(t % Y), z = (int)(t / Y);
    int count = 0;
    sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int a = -p.radius; a <= p.radius; ++a) {  // src/Postprocessing3d.cpp:31-48
        const int xn = x + a;
        if (xn < 0 || xn >= X) continue;
        for (int b = -p.radius; b <= p.radius; ++b) {
            const int yn = y + b;
            if (yn < 0 || yn >= Y) continue;
            for (int c = -p.radius; c <= p.radius; ++c) {
                const int zn = z + c;
                if (zn < 0 || zn >= Z) continue;
                const size_t row = (size_t)zn * Y + yn;
                if (!((p.occ[row * p.g.XW + (xn >> 6)] >> (xn & 63)) & 1ull)) continue;
                ++count;
                const float4 v = cl_color(p, xn, row);
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

__global__ __launch_bounds__(256) void export_overlay_kernel(const int *__restrict__ index,
                                                             const float4 *__restrict__ rgba,
                                                             long long first, long long last,
                                                             size_t i0, float4 *__restrict__ out) {
    const long long e = first + (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < last) out[(size_t)index[e] - i0] = rgba[e];
}

}  // namespace arvx
