// arvx_device.h -- shared device-side definitions for the gfx950 kernels.
//
// Arithmetic contract (must match what the reference computes per voxel,
// reference src/VoxelCarving.cpp:18-21,41-54 and src/Model.h:134-140):
//   w    = (float(y)*s, float(x)*s, float(-z)*s, 1)              fp32
//   p_k  = double(M[r][k]) * double(w[k])                        exact in fp64
//   a_r  = float(((p0 + p1) + p2) + p3)                          cv::gemm generic path
//   u,v  = a_0 / a_2, a_1 / a_2                                  IEEE fp32 divide
//   px   = (int)roundf(u), py = (int)roundf(v); inside iff 0<=px<W, 0<=py<H
// The library is built with -ffp-contract=off so only explicit fma() fuses.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace arvx {

constexpr int kTileX = 64, kTileY = 8, kTileZ = 8;  // voxels per workgroup
constexpr int kSubX = 16;                           // x extent of one wave's sub-tile
constexpr int kCoarseX = 64;  // pre-pass tile: 64 x (8 << cyShift) x (8 << czShift) voxels
constexpr uint32_t kDone4 = 0x02020202u;            // 4 voxels carved+seen
constexpr int kMaxImageDim = 16384;

enum : int { kClsOut = 0, kClsFg = 1, kClsCarved = 2, kClsMixed = 3 };

struct CarveParams {
    uint8_t *state;         // slab state plane
    const float *M;         // V x 12
    const uint32_t *bg;     // V x bgWords, bit = 1 where the mask pixel is background
    const int *sat;         // V x satStride, summed-area table of foreground pixels
    unsigned long long *stats;
    int X, Y, Z;            // slab extent in voxels (Z = planes held)
    int zoff;               // global z of slab plane 0 (contiguous slabs)
    int zstride, zphase;    // striped slabs: local 8-plane group g is global group
                            // g*zstride + zphase (contiguous: 1, 0)
    float s;                // voxel edge
    int W, H;
    int bgWords, satStride;
    int v0, v1;             // view range [v0, v1)
    unsigned flags;         // bit0 no cull, bit1 stats, bit2 state is fresh (skip the load)
    int tilesX, tilesY, tilesZ;
    // coarse pre-pass results
    int coarseX, coarseY, coarseZ, nchunks;
    int cyShift, czShift;   // coarse tile = 64 x (8 << cyShift) x (8 << czShift)
    unsigned long long *coarseMixed;  // [ncoarse][nchunks] views to re-classify per sub-tile
    unsigned long long *coarseFg;     // [ncoarse][nchunks] views that see only foreground
    uint8_t *coarseCarved;            // [ncoarse] some view carves the whole coarse tile
};

// global z of local plane lz
__device__ __forceinline__ int global_z(const CarveParams &p, int lz) {
    return p.zoff + (((lz >> 3) * p.zstride + p.zphase) << 3) + (lz & 7);
}

// Rounded pixel of one projected voxel.  a0,a1,a2 are the fp32 row results.
// Returns false (outside) for non-finite quotients as x86's cvttss2si does.
__device__ __forceinline__ bool pixel_of(float a0, float a1, float a2, int W, int H,
                                         int &pix) {
    float u = a0 / a2;
    float v = a1 / a2;
    float ru = roundf(u);
    float rv = roundf(v);
    // W,H <= 16384 are exact in fp32; ru,rv are integral, NaN fails every test.
    bool in = (ru >= 0.f) && (ru < (float)W) && (rv >= 0.f) && (rv < (float)H);
    pix = in ? (int)rv * W + (int)ru : 0;
    return in;
}

__device__ __forceinline__ float row_sum(double p01, double p2, double p3) {
    return (float)((p01 + p2) + p3);
}

}  // namespace arvx
