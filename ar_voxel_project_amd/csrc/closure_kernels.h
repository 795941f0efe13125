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
    BitGrid g;                         // the context's planes (the whole grid, or slab + halo)
    const unsigned long long *occ;     // what the closure calls occupied
    const unsigned long long *unseen;  // voxels whose colour is UNSEEN_COLOR (null: none)
    int radius;                        // (kernelSize - 1) / 2
    SparseList col;                    // the colour pass's list: its index ...
    const float4 *col_rgba;            // ... and r, g, b, has-sample flag per entry
};

// colour of an occupied voxel as Model::get returns it at src/main.cpp:297
__device__ inline float4 cl_color(const ClosureParams &p, int x, size_t row) {
    if (p.unseen && ((p.unseen[row * p.g.XW + (x >> 6)] >> (x & 63)) & 1ull))
        return make_float4(204.f, 0.f, 0.f, 1.f);
    const int k = sparse_find(p.col, p.g.XW, x, row);
    if (k >= 0) {
        const float4 c = p.col_rgba[k];
        if (c.w != 0.f) return make_float4(c.x, c.y, c.z, 1.f);
    }
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

// The 3x3x3 box (kernelSize 3, what src/main.cpp:298 passes) without a dependent load per
// neighbour: the nine voxel rows around the voxel are read as words first -- occupancy, UNSEEN
// paint, the colour list's index words: 27 independent loads --, then the colours of nine
// neighbours at a time (one x offset) are fetched together and added in the reference's order
// (x offset outermost, then y, then z, src/Postprocessing3d.cpp:31-48).
__device__ inline int cl_gather3(const ClosureParams &p, size_t i, float4 &sum) {
    const int X = p.g.X, Y = p.g.Y, Z = p.g.Z, XW = p.g.XW;
    const int x = (int)(i % X);
    const size_t t = i / X;
    const int y = (int)(t % Y), z = (int)(t / Y);
    const int xw = x >> 6, xb = x & 63;
    // 3-bit windows of the nine rows (y + b, z + c), row r = 3 (b + 1) + (c + 1) at bits 3 r .. 3 r + 2,
    // bit a = voxel x + a - 1: occupancy, UNSEEN paint, membership in the colour list (the bits of
    // the voxel's own word; a neighbour in the next / previous word is looked up on its own below).
    // Packed, and with ONE rank per row (of the window's first voxel), so that the kernel stays
    // within 64 registers: eight waves per SIMD hide what is a chain of dependent gathers.
    unsigned occ = 0, uns = 0, col = 0;
    int kbase[9];
    auto window = [&](const unsigned long long *plane, size_t at) -> unsigned {
        const unsigned long long w = plane[at];
        unsigned v = xb ? (unsigned)((w >> (xb - 1)) & 7ull) : (unsigned)((w << 1) & 7ull);
        if (xb == 0 && xw > 0) v |= (unsigned)(plane[at - 1] >> 63);
        if (xb == 63 && xw + 1 < XW) v |= (unsigned)(plane[at + 1] & 1ull) << 2;
        return v;
    };
    const int lo = xb ? xb - 1 : 0;  // first bit of the window inside the word
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const int yn = y + r / 3 - 1, zn = z + r % 3 - 1;
        kbase[r] = 0;
        if (yn < 0 || yn >= Y || zn < 0 || zn >= Z) continue;
        const size_t at = ((size_t)zn * Y + yn) * XW + xw;
        occ |= window(p.occ, at) << (3 * r);
        if (p.unseen) uns |= window(p.unseen, at) << (3 * r);
        if (p.col.w) {
            const SparseWord e = sparse_word(p.col, at);
            col |= (xb ? (unsigned)((e.bits >> (xb - 1)) & 7ull) : (unsigned)((e.bits << 1) & 7ull)) << (3 * r);
            kbase[r] = e.rank + __popcll(e.bits & ((1ull << lo) - 1ull));
        }
    }
    int count = 0;
    sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // x offset a - 1
        const int xn = x + a - 1;
        const bool other_word = (xn >> 6) != xw;  // (xb = 0, a = 0 or xb = 63, a = 2)
        float4 rgb[9];  // w = has-sample flag; (0, 0, 0, 0): not in the list
#pragma unroll
        for (int r = 0; r < 9; ++r) {  // the nine lookups of this x offset, all in flight
            int k = -1;
            if ((occ >> (3 * r + a)) & 1u) {
                if (!other_word) {
                    const unsigned w3 = (col >> (3 * r)) & 7u;
                    if ((w3 >> a) & 1u) k = kbase[r] + __popc(w3 & ((1u << a) - 1u));
                } else if (p.col.w && xn >= 0 && xn < X) {
                    const int yn = y + r / 3 - 1, zn = z + r % 3 - 1;
                    k = sparse_find(p.col, XW, xn, (size_t)zn * Y + yn);
                }
            }
            rgb[r] = (k >= 0) ? p.col_rgba[k] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (!((occ >> (3 * r + a)) & 1u)) continue;
            ++count;
            float4 v = make_float4(50.f, 168.f, 141.f, 1.f);
            if (rgb[r].w != 0.f) v = make_float4(rgb[r].x, rgb[r].y, rgb[r].z, 1.f);
            if ((uns >> (3 * r + a)) & 1u) v = make_float4(204.f, 0.f, 0.f, 1.f);
            sum.x = sum.x + v.x;
            sum.y = sum.y + v.y;
            sum.z = sum.z + v.z;
            sum.w = sum.w + v.w;
        }
    }
    return count;
}

// one thread per filled voxel (index ascending): mean RGBA of its occupied neighbours
template <bool BOX3>
__global__ __launch_bounds__(256) void closure_fill_kernel(const ClosureParams p,
                                                           const int *__restrict__ index,
                                                           long long n,
                                                           const long long *__restrict__ n_dev,
                                                           float4 *__restrict__ rgba) {
    // n: the list's capacity; n_dev: its length as the compaction left it on the device
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n || e >= *n_dev) return;
    float4 sum;
    const int count = BOX3 ? cl_gather3(p, (size_t)index[e], sum) : cl_gather(p, (size_t)index[e], sum);
    const float fc = (float)count;  // Eigen `sum /= count`, src/Postprocessing3d.cpp:49-51
    rgba[e] = make_float4(sum.x / fc, sum.y / fc, sum.z / fc, sum.w / fc);
}

// (base: flat index of the first owned voxel in the list's numbering over the context's planes)
__global__ __launch_bounds__(256) void export_overlay_kernel(const int *__restrict__ index,
                                                             const float4 *__restrict__ rgba,
                                                             long long first, long long last,
                                                             size_t base, size_t i0,
                                                             float4 *__restrict__ out) {
    const long long e = first + (long long)blockIdx.x * 256 + threadIdx.x;
    if (e < last) out[(size_t)index[e] - base - i0] = rgba[e];
}

}  // namespace arvx
