// carve_kernels.h -- dense silhouette carve for gfx950 (MI355X).
//
// Replaces the voxel loop of the reference's carve(): src/VoxelCarving.cpp:38-55
// (one view) and :60-72 (all views).  One launch covers all requested views:
// the state plane is read once, every view is applied in registers, and the
// plane is written once.
//
// Work decomposition
//   workgroup (256 threads) = tile of 64 x 8 x 8 voxels (x fastest in memory)
//   wave                    = sub-tile of 16 x 8 x 8 voxels
//   lane                    = 4 consecutive x (one dword of state) at one y,
//                             for 4 consecutive z: 16 voxels in 4 registers
// so the four waves of a workgroup together touch whole 64-byte runs.
//
// Per sub-tile, before any voxel is projected, lane i classifies view i: the
// eight corners of the sub-tile's world box are projected in fp64, a rigorous
// error margin is added, and the resulting pixel rectangle is looked up in
// the view's summed-area table of foreground pixels:
//   rectangle outside the image            -> no voxel is seen by this view
//   inside, no foreground pixel            -> every voxel is carved: sub-tile done
//   inside, only foreground pixels         -> every voxel is seen, none carved
//   anything else                          -> evaluate the 1024 voxels exactly
// A wave ballot over the lanes turns this into three 64-bit view masks; the
// "carved" mask ends the sub-tile at once (carved implies seen, reference
// src/VoxelCarving.cpp:50-54), the "mixed" mask drives the exact per-voxel
// loop, which itself stops as soon as a ballot finds all 1024 voxels carved.
#pragma once

#include "arvx_device.h"

namespace arvx {

// Conservative classification of the voxel box [x0,x1]x[y0,y1]x[z0,z1]
// (inclusive, z global) against one view.  Rigour: every voxel's w = fl32(i*s)
// lies between the corner values (rounding is monotone), the row values a_r are
// affine in w, and u = a_0/a_2 is linear-fractional, so over a box on which
// a_2 keeps its sign the extremes of u,v sit at the eight corners.  The
// computed a_r differ from the real ones by <= 2^-24*E_r (+fp64 dust), E_r the
// sum of |terms|; 2^-22*E_r is used.  The fp32 divide adds 2^-24*|u|; the
// corner quotients are taken in fp32 (3 ulp); 2^-20*|u| + 2^-12 covers both.
__device__ inline int classify_box(const float *__restrict__ M, float s, int x0, int x1,
                                   int y0, int y1, int z0, int z1, int W, int H,
                                   const int *__restrict__ sat) {
    const double wy[2] = {(double)((float)y0 * s), (double)((float)y1 * s)};
    const double wx[2] = {(double)((float)x0 * s), (double)((float)x1 * s)};
    const double wz[2] = {(double)((float)(-z0) * s), (double)((float)(-z1) * s)};
    const double ay = fmax(fabs(wy[0]), fabs(wy[1]));
    const double ax = fmax(fabs(wx[0]), fabs(wx[1]));
    const double az = fmax(fabs(wz[0]), fabs(wz[1]));
    double m[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = (double)M[i];
    double E[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        E[r] = fabs(m[4 * r]) * ay + fabs(m[4 * r + 1]) * ax + fabs(m[4 * r + 2]) * az +
               fabs(m[4 * r + 3]);
    float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
    double cmin = INFINITY, cmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const double y = wy[c & 1], x = wx[(c >> 1) & 1], z = wz[c >> 2];
        const double a0 = fma(m[0], y, fma(m[1], x, fma(m[2], z, m[3])));
        const double a1 = fma(m[4], y, fma(m[5], x, fma(m[6], z, m[7])));
        const double a2 = fma(m[8], y, fma(m[9], x, fma(m[10], z, m[11])));
        cmin = fmin(cmin, a2);
        cmax = fmax(cmax, a2);
        const float fu = (float)a0 / (float)a2;
        const float fv = (float)a1 / (float)a2;
        umin = fminf(umin, fu);
        umax = fmaxf(umax, fu);
        vmin = fminf(vmin, fv);
        vmax = fmaxf(vmax, fv);
    }
    const double k22 = 2.384185791015625e-07;  // 2^-22
    const double k20 = 9.5367431640625e-07;    // 2^-20
    const double k12 = 2.44140625e-04;         // 2^-12
    const double eps2 = E[2] * k22;
    const double cabs = (cmin > 0.0) ? cmin : ((cmax < 0.0) ? -cmax : 0.0);
    if (!(cabs > 8.0 * eps2 + 1e-30)) return kClsMixed;  // denominator may vanish
    const double cden = cabs - eps2;
    const double Ua = fmax(fabs((double)umin), fabs((double)umax));
    const double Va = fmax(fabs((double)vmin), fabs((double)vmax));
    if (!(Ua < 1.0e6 && Va < 1.0e6)) return kClsMixed;  // also NaN
    const double mu = (E[0] * k22 + Ua * eps2) / cden + Ua * k20 + k12;
    const double mv = (E[1] * k22 + Va * eps2) / cden + Va * k20 + k12;
    // roundf(t) lies in [t-0.5, t+0.5]
    const int pxlo = (int)ceil((double)umin - mu - 0.5);
    const int pxhi = (int)floor((double)umax + mu + 0.5);
    const int pylo = (int)ceil((double)vmin - mv - 0.5);
    const int pyhi = (int)floor((double)vmax + mv + 0.5);
    if (pxhi < 0 || pxlo >= W || pyhi < 0 || pylo >= H) return kClsOut;
    if (pxlo < 0 || pxhi >= W || pylo < 0 || pyhi >= H) return kClsMixed;
    const int S = W + 1;
    const int cnt = sat[(pyhi + 1) * S + pxhi + 1] - sat[pylo * S + pxhi + 1] -
                    sat[(pyhi + 1) * S + pxlo] + sat[pylo * S + pxlo];
    if (cnt == 0) return kClsCarved;
    const int area = (pxhi - pxlo + 1) * (pyhi - pylo + 1);
    return (cnt == area) ? kClsFg : kClsMixed;
}

template <bool kAligned4>
__global__ __launch_bounds__(256) void carve_fused_kernel(const CarveParams p) {
    // Blocks b, b+8, b+16.. share an XCD (and its L2): hand each XCD one
    // contiguous run of tiles so neighbouring 64-byte runs meet in one L2.
    const unsigned ntiles = (unsigned)p.tilesX * p.tilesY * p.tilesZ;
    const unsigned per = gridDim.x >> 3;
    const unsigned tile = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int tx = tile % p.tilesX;
    const int ty = (tile / p.tilesX) % p.tilesY;
    const int tz = tile / (p.tilesX * p.tilesY);

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int sx0 = tx * kTileX + wave * kSubX;
    if (sx0 >= p.X) return;  // wave-uniform
    const int sy0 = ty * kTileY;
    const int sz0 = tz * kTileZ;
    const int sx1 = min(sx0 + kSubX - 1, p.X - 1);
    const int sy1 = min(sy0 + kTileY - 1, p.Y - 1);
    const int sz1 = min(sz0 + kTileZ - 1, p.Z - 1);

    const int x = sx0 + 4 * (lane & 3);
    const int y = sy0 + ((lane >> 2) & 7);
    const int zb = sz0 + 4 * (lane >> 5);
    const bool lane_ok = (x < p.X) && (y < p.Y);

    // per-lane world coordinates, reference src/Model.h:134-140
    const double dwy = (double)((float)y * p.s);
    double dwx[4], dwz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dwx[j] = (double)((float)(x + j) * p.s);
#pragma unroll
    for (int k = 0; k < 4; ++k) dwz[k] = (double)((float)(-(p.zoff + zb + k)) * p.s);

    uint32_t st[4] = {kDone4, kDone4, kDone4, kDone4};
    bool loaded = false, all_carved = false, all_done = false;
    const size_t row = (size_t)p.X;
    const size_t plane = (size_t)p.X * p.Y;

    for (int vc = p.v0; vc < p.v1 && !all_done; vc += 64) {
        const int myv = vc + lane;
        int cls = kClsOut;
        if (myv < p.v1) {
            cls = (p.flags & 1u)
                      ? kClsMixed
                      : classify_box(p.M + 12 * myv, p.s, sx0, sx1, sy0, sy1, p.zoff + sz0,
                                     p.zoff + sz1, p.W, p.H,
                                     p.sat + (size_t)myv * p.satStride);
        }
        const unsigned long long carved = __ballot(cls == kClsCarved);
        unsigned long long mixed = __ballot(cls == kClsMixed);
        const unsigned long long infg = __ballot(cls == kClsFg);
        if (p.flags & 2u) {
            if (lane == 0) {
                if (vc == p.v0) atomicAdd(&p.stats[0], 1ull);
                if (carved) atomicAdd(&p.stats[1], 1ull);
                atomicAdd(&p.stats[2], (unsigned long long)__popcll(mixed));
                atomicAdd(&p.stats[3], (unsigned long long)min(64, p.v1 - vc));
            }
        }
        if (carved) {
            all_carved = true;
            break;
        }
        if (!loaded) {
            loaded = true;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int z = zb + k;
                if (lane_ok && z < p.Z) {
                    const uint8_t *src = p.state + (size_t)z * plane + (size_t)y * row + x;
                    if (kAligned4) {
                        st[k] = *reinterpret_cast<const uint32_t *>(src);
                    } else {
                        uint32_t w = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            w |= (uint32_t)((x + j < p.X) ? src[j] : (uint8_t)2) << (8 * j);
                        st[k] = w;
                    }
                }
            }
        }
        if (infg) {
#pragma unroll
            for (int k = 0; k < 4; ++k) st[k] |= kDone4;  // seen, src/VoxelCarving.cpp:54
        }
        while (mixed) {
            const int b = __ffsll((long long)mixed) - 1;
            mixed &= mixed - 1;
            const int view = __builtin_amdgcn_readfirstlane(vc + b);
            const float *__restrict__ Mv = p.M + 12 * view;
            const uint32_t *__restrict__ bgv = p.bg + (size_t)view * p.bgWords;
            double p01[3][4], p3[3], m2[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double p0 = (double)Mv[4 * r] * dwy;
                const double m1 = (double)Mv[4 * r + 1];
                m2[r] = (double)Mv[4 * r + 2];
                p3[r] = (double)Mv[4 * r + 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) p01[r][j] = p0 + m1 * dwx[j];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!__any(st[k] != kDone4)) continue;  // these 256 voxels are finished
                const double p20 = m2[0] * dwz[k], p21 = m2[1] * dwz[k], p22 = m2[2] * dwz[k];
                uint32_t w = st[k];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a0 = row_sum(p01[0][j], p20, p3[0]);
                    const float a1 = row_sum(p01[1][j], p21, p3[1]);
                    const float a2 = row_sum(p01[2][j], p22, p3[2]);
                    int pix;
                    const bool in = pixel_of(a0, a1, a2, p.W, p.H, pix);
                    const uint32_t word = in ? bgv[pix >> 5] : 0u;
                    const uint32_t isbg = (word >> (pix & 31)) & 1u;
                    const uint32_t seen = in ? (2u << (8 * j)) : 0u;
                    w = (w | seen) & ~(isbg << (8 * j));
                }
                st[k] = w;
            }
            if (__all(st[0] == kDone4 && st[1] == kDone4 && st[2] == kDone4 &&
                      st[3] == kDone4)) {
                all_done = true;
                break;
            }
        }
    }

    if (all_carved) {
#pragma unroll
        for (int k = 0; k < 4; ++k) st[k] = kDone4;
    } else if (!loaded) {
        return;  // empty view range: nothing changed
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int z = zb + k;
        if (lane_ok && z < p.Z) {
            uint8_t *dst = p.state + (size_t)z * plane + (size_t)y * row + x;
            if (kAligned4) {
                *reinterpret_cast<uint32_t *>(dst) = st[k];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x + j < p.X) dst[j] = (uint8_t)(st[k] >> (8 * j));
            }
        }
    }
}

// ---- view pre-processing ---------------------------------------------------

// bit i of plane v = 1 iff all C channel bytes of pixel i are zero
// (reference src/VoxelCarving.cpp:49-50).
__global__ __launch_bounds__(256) void mask_to_bits_kernel(const uint8_t *__restrict__ masks,
                                                           int C, int npix,
                                                           uint32_t *__restrict__ bg,
                                                           int bgWords) {
    const int v = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    bool isbg = false;
    if (pix < npix) {
        const uint8_t *q = masks + ((size_t)v * npix + pix) * C;
        isbg = true;
        for (int c = 0; c < C; ++c) isbg = isbg && (q[c] == 0);
    }
    const unsigned long long b = __ballot(isbg);
    if ((threadIdx.x & 63) == 0) {
        const int w0 = pix >> 5;
        uint32_t *dst = bg + (size_t)v * bgWords;
        if (w0 < bgWords) dst[w0] = (uint32_t)b;
        if (w0 + 1 < bgWords) dst[w0 + 1] = (uint32_t)(b >> 32);
    }
}

// Summed-area table of FOREGROUND pixels, (H+1) x (W+1) ints per view.
// One wave per image row: 64-wide inclusive scans with a running carry.
__global__ __launch_bounds__(64) void sat_rows_kernel(const uint32_t *__restrict__ bg,
                                                      int bgWords, int W, int H,
                                                      int *__restrict__ sat, int satStride) {
    const int v = blockIdx.y;
    const int yrow = blockIdx.x;  // 0..H-1
    const int lane = threadIdx.x;
    const uint32_t *b = bg + (size_t)v * bgWords;
    int *out = sat + (size_t)v * satStride + (size_t)(yrow + 1) * (W + 1);
    if (lane == 0) out[0] = 0;
    int carry = 0;
    for (int x0 = 0; x0 < W; x0 += 64) {
        const int xx = x0 + lane;
        int fg = 0;
        if (xx < W) {
            const int pix = yrow * W + xx;
            fg = 1 - (int)((b[pix >> 5] >> (pix & 31)) & 1u);
        }
        int sc = fg;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(sc, d);
            if (lane >= d) sc += t;
        }
        if (xx < W) out[xx + 1] = carry + sc;
        carry += __shfl(sc, 63);
    }
}

__global__ __launch_bounds__(256) void sat_cols_kernel(int W, int H, int *__restrict__ sat,
                                                       int satStride) {
    const int v = blockIdx.y;
    const int xcol = blockIdx.x * 256 + threadIdx.x;  // 0..W
    if (xcol > W) return;
    int *s = sat + (size_t)v * satStride;
    int acc = 0;
    s[xcol] = 0;
    for (int yy = 1; yy <= H; ++yy) {
        acc += s[(size_t)yy * (W + 1) + xcol];
        s[(size_t)yy * (W + 1) + xcol] = acc;
    }
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *__restrict__ dst, size_t n4,
                                                       uint32_t value, uint8_t *__restrict__ tail,
                                                       int ntail) {
    // n4 = number of 16-byte groups
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 v4 = make_uint4(value, value, value, value);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<uint4 *>(dst)[i] = v4;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = (uint8_t)value;
}

// occupancy bit-plane: voxel i -> bit i%32 of word i/32
__global__ __launch_bounds__(256) void pack_occupancy_kernel(const uint8_t *__restrict__ state,
                                                             size_t n,
                                                             uint32_t *__restrict__ words) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t nround = (n + 63) & ~(size_t)63;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += stride) {
        const bool occ = (i < n) && (state[i] & 1u);
        const unsigned long long b = __ballot(occ);
        if ((threadIdx.x & 63) == 0) {
            const size_t w0 = i >> 5;
            const size_t nw = (n + 31) >> 5;
            if (w0 < nw) words[w0] = (uint32_t)b;
            if (w0 + 1 < nw) words[w0 + 1] = (uint32_t)(b >> 32);
        }
    }
}

}  // namespace arvx
