// color_kernels.h -- per-voxel colour vote for gfx950 (model export: state_kernels.h).
//
// Replaces the reference's colour pass: the `voxel_pass` loops of
// src/ColorReconstruction.h:34-74 plus the bodies of reconstructClosestColor
// (src/ColorReconstruction.cpp:26-42) and reconstructAvgColor (:52-66), and
// Model::handleUnseen (src/Model.cpp:36-47) at export time.
//
// The reference walks all N voxels and votes only on occupied voxels that are
// not "inner" (all six neighbours occupied, src/Model.h:126-132).  Here the
// surface is first found on a bit plane and compacted into an index list in
// ascending flat-index order (bitplane_kernels.h), then one thread per surface
// voxel gathers its samples from the BGR images.
#pragma once

#include "arvx_device.h"

namespace arvx {

// exclusive scan of `counts` (n entries) by ONE workgroup; offsets[n] = total
__global__ __launch_bounds__(256) void surface_scan_kernel(const int *__restrict__ counts, int n,
                                                           long long *__restrict__ offsets) {
    __shared__ long long wtot[4];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int b0 = 0; b0 < n; b0 += 256) {
        const int i = b0 + threadIdx.x;
        const long long v = (i < n) ? counts[i] : 0;
        long long sc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long t = __shfl_up(sc, d);
            if (lane >= d) sc += t;
        }
        if (lane == 63) wtot[wave] = sc;
        __syncthreads();
        long long pre = carry_s;
        for (int w = 0; w < wave; ++w) pre += wtot[w];
        if (i < n) offsets[i] = pre + sc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = carry_s;
}

struct VoteParams {
    const int *index;  // surface voxels, slab-local flat index over owned planes
    long long n;       // entries (with n_dev: the list's capacity)
    const long long *n_dev;  // the list's length as the compaction left it on the device (or null)
    int X, Y;
    int zglob0;  // global z of owned plane 0
    float s;
    int V, W, H;
    const float *M;       // V x 12
    const float *campos;  // V x 3
    const uint8_t *images;  // V x H x W x 3 BGR
    int mode;
    float4 *rgba;  // n: r, g, b and 1 if the voxel received >= 1 sample, else 0
    float *depth;  // n   (minimum sample depth)
    uint8_t *has;  // n   (1 if >= 1 sample)
};

// (float)cv::norm(cameras[i] - word_coord), src/ColorReconstruction.h:59: fp32 differences of
// the two Vec4f (the fourth is 1 - 1), then the square root of an fp64 sum of squares in the
// order ((d0^2 + d1^2) + d2^2) + d3^2 (normL2Sqr<float, double>, one 4-way step)
__device__ __forceinline__ float sample_depth(const float *__restrict__ c, float w0, float w1,
                                              float w2) {
    const double e0 = (double)(c[0] - w0), e1 = (double)(c[1] - w1), e2 = (double)(c[2] - w2);
    const double e3 = (double)(1.f - 1.f);
    const double acc = ((e0 * e0 + e1 * e1) + e2 * e2) + e3 * e3;
    return (float)sqrt(acc);
}

// one row triple of M * world for voxel (x, y, z) (global z), reference src/Model.h:134-140 and
// src/VoxelCarving.cpp:18-21
template <bool LEFT>
__device__ __forceinline__ void project_rows(const float *__restrict__ Mv, float s, int x, int y,
                                             int z, float a[3]) {
    const float w0 = (float)y * s, w1 = (float)x * s, w2 = (float)(-z) * s;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const double p0 = (double)Mv[4 * r] * (double)w0;
        const double p1 = (double)Mv[4 * r + 1] * (double)w1;
        const double p2 = (double)Mv[4 * r + 2] * (double)w2;
        const double p3 = (double)Mv[4 * r + 3];
        a[r] = row_sum<LEFT>(p0, p1, p2, p3);
    }
}

// self-test hooks (arvx_selftest_project / _depth): the raw values the kernels compute, for a
// host that has OpenCV to compare with cv::gemm / cv::norm (include/arvx/opencv_dropin.hpp)
struct SelftestMatrix {
    float m[12];
};
template <bool LEFT>
__global__ __launch_bounds__(256) void selftest_project_kernel(const SelftestMatrix M, float s,
                                                               const int *__restrict__ xyz,
                                                               long long n, float *__restrict__ rows,
                                                               float *__restrict__ uv) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a[3];
    project_rows<LEFT>(M.m, s, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], a);
    rows[3 * i] = a[0];
    rows[3 * i + 1] = a[1];
    rows[3 * i + 2] = a[2];
    uv[2 * i] = a[0] / a[2];
    uv[2 * i + 1] = a[1] / a[2];
}
__global__ __launch_bounds__(256) void selftest_depth_kernel(const SelftestMatrix cam, float s,
                                                             const int *__restrict__ xyz,
                                                             long long n, float *__restrict__ depth) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float w0 = (float)xyz[3 * i + 1] * s, w1 = (float)xyz[3 * i] * s,
                w2 = (float)(-xyz[3 * i + 2]) * s;
    depth[i] = sample_depth(cam.m, w0, w1, w2);
}

// the samples themselves (arvx_color_samples): one thread per (voxel, view); 8 bytes out:
// r, g, b, valid, depth -- the DCLR entries Model::addColor would get, in view order
template <bool LEFT>
__global__ __launch_bounds__(256) void color_samples_kernel(const VoteParams p,
                                                            const long long *__restrict__ index,
                                                            uint2 *__restrict__ out) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.n * p.V) return;
    const long long k = t / p.V;
    const int v = (int)(t % p.V);
    const long long i = index[k];
    const int x = (int)(i % p.X);
    const int y = (int)((i / p.X) % p.Y);
    const int z = p.zglob0 + (int)(i / ((long long)p.X * p.Y));
    float a[3];
    project_rows<LEFT>(p.M + 12 * v, p.s, x, y, z, a);
    int pix;
    uint2 o = make_uint2(0u, 0u);
    if (pixel_of(a[0], a[1], a[2], p.W, p.H, pix)) {  // ColorReconstruction.h:54-57
        const uint8_t *q = p.images + ((size_t)v * p.W * p.H + pix) * 3;  // Vec3b is BGR
        const float w0 = (float)y * p.s, w1 = (float)x * p.s, w2 = (float)(-z) * p.s;
        o.x = (unsigned)q[2] | ((unsigned)q[1] << 8) | ((unsigned)q[0] << 16) | (1u << 24);
        o.y = __float_as_uint(sample_depth(p.campos + 3 * v, w0, w1, w2));
    }
    out[t] = o;
}

// One thread per surface voxel, all views.  What a (voxel, view) pair costs is vector instructions
// (36 exact projections per voxel), so the loop is kept lean:
//  * the views' matrices are widened to fp64 ONCE per workgroup, into LDS (twelve conversions per
//    view and lane otherwise; up to kVoteLdsViews views, more take the plain loop);
//  * both quotients share one reciprocal where divide2_shared_rcp is the IEEE quotient (arvx_device.h:
//    2^-60 <= |a2| <= 2^60, |a0|, |a1| <= 2^60 in every lane of the wave), `/` otherwise;
//  * the square root of a sample's depth is taken only when its fp64 sum of squares is below the
//    smallest so far: depth = (float)sqrt(sum) is monotone in the sum, so a sample with a sum that
//    is not smaller cannot be strictly closer (and the first view wins ties, .cpp:33-40).
constexpr int kVoteLdsViews = 256;
template <bool LEFT>
__global__ __launch_bounds__(256) void color_vote_kernel(const VoteParams p) {
    __shared__ double s_M[kVoteLdsViews * 12];
    const bool lds = p.V <= kVoteLdsViews;  // (uniform)
    if (lds) {
        for (int k = threadIdx.x; k < p.V * 12; k += 256) s_M[k] = (double)p.M[k];
        __syncthreads();
    }
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= p.n || (p.n_dev && t >= *p.n_dev)) return;
    const int i = p.index[t];
    const int x = i % p.X;
    const int y = (i / p.X) % p.Y;
    const int z = p.zglob0 + i / (p.X * p.Y);
    // Model::toWord, reference src/Model.h:134-140
    const float w0 = (float)y * p.s, w1 = (float)x * p.s, w2 = (float)(-z) * p.s;
    const double d0w = (double)w0, d1w = (double)w1, d2w = (double)w2;
    const float wlim = (float)p.W - 0.5f, hlim = (float)p.H - 0.5f;
    unsigned sr = 0, sg = 0, sb = 0, n = 0;
    float best = 0.f, br = 0.f, bgc = 0.f, bb = 0.f;
    double best_sum = 0.0;
    for (int v = 0; v < p.V; ++v) {
        float a[3];
        if (lds) {
            const double *Md = s_M + 12 * v;
#pragma unroll
            for (int r = 0; r < 3; ++r)
                a[r] = row_sum<LEFT>(Md[4 * r] * d0w, Md[4 * r + 1] * d1w, Md[4 * r + 2] * d2w, Md[4 * r + 3]);
        } else {
            const float *__restrict__ Mv = p.M + 12 * v;
#pragma unroll
            for (int r = 0; r < 3; ++r)
                a[r] = row_sum<LEFT>((double)Mv[4 * r] * d0w, (double)Mv[4 * r + 1] * d1w,
                                     (double)Mv[4 * r + 2] * d2w, (double)Mv[4 * r + 3]);
        }
        float qu, qv;
        const bool tame = fabsf(a[2]) >= 0x1p-60f && fabsf(a[2]) <= 0x1p60f && fabsf(a[0]) <= 0x1p60f &&
                          fabsf(a[1]) <= 0x1p60f;
        if (__all(tame)) {
            divide2_shared_rcp(a[0], a[1], a[2], qu, qv);
        } else {
            qu = a[0] / a[2];
            qv = a[1] / a[2];
        }
        int pix;
        if (!pixel_from_quotients(qu, qv, p.W, wlim, hlim, pix)) continue;  // ColorReconstruction.h:54-57
        const uint8_t *q = p.images + ((size_t)v * p.W * p.H + pix) * 3;
        const unsigned b = q[0], g = q[1], r = q[2];  // Vec3b is BGR; colour = (R,G,B,1), :59
        // (float)cv::norm(cameras[v] - word_coord): sample_depth, with the root taken when it matters
        const float *__restrict__ c = p.campos + 3 * v;
        const double e0 = (double)(c[0] - w0), e1 = (double)(c[1] - w1), e2 = (double)(c[2] - w2);
        const double e3 = (double)(1.f - 1.f);
        const double sum = ((e0 * e0 + e1 * e1) + e2 * e2) + e3 * e3;
        if (n == 0 || sum < best_sum) {
            const float depth = (float)sqrt(sum);
            if (n == 0 || depth < best) {  // strict <: the first view wins ties, .cpp:33-40
                best = depth;
                br = (float)r;
                bgc = (float)g;
                bb = (float)b;
            }
            // (a smaller sum with the same rounded depth: later samples are compared with the
            // smaller sum, whose depth is the same or smaller -- still exact)
            best_sum = sum;
        }
        sr += r;
        sg += g;
        sb += b;
        ++n;
    }
    p.has[t] = n ? 1 : 0;
    p.depth[t] = best;
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    if (n) {
        if (p.mode == 0) {
            o0 = br;
            o1 = bgc;
            o2 = bb;
        } else {
            const float fn = (float)n;  // sums <= 255*V are exact in fp32
            o0 = roundf((float)sr / fn);  // .cpp:64-65
            o1 = roundf((float)sg / fn);
            o2 = roundf((float)sb / fn);
        }
    }
    p.rgba[t] = make_float4(o0, o1, o2, n ? 1.f : 0.f);
}

}  // namespace arvx
