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
// eight corners of the sub-tile's world box are projected, a rigorous error
// margin is added, and the resulting pixel rectangle is looked up in the view's
// summed-area table of foreground pixels (a pre-pass does the same for coarse
// 64x32x32 tiles first, so most sub-tiles inherit their answer):
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

// World-space box of a block of voxels: the fl32 products Model::toWord forms at
// its two faces per axis (x/y swapped, z negated; reference src/Model.h:134-140).
struct BoxW {
    float wy0, wy1, wx0, wx1, wz0, wz1;
};

__device__ __forceinline__ BoxW make_box(float s, int x0, int x1, int y0, int y1, int z0, int z1) {
    BoxW b;
    b.wy0 = (float)y0 * s;
    b.wy1 = (float)y1 * s;
    b.wx0 = (float)x0 * s;
    b.wx1 = (float)x1 * s;
    b.wz0 = (float)(-z0) * s;
    b.wz1 = (float)(-z1) * s;
    return b;
}

// Conservative classification of a voxel box against one view, all in fp32.
//
// Rigour.  Every voxel's w = fl32(i*s) lies between the corner values (rounding
// is monotone); the rows a_r are affine in w and u = a_0/a_2, v = a_1/a_2 are
// linear-fractional, so over a box on which a_2 keeps its sign the real-valued
// extremes of u,v sit at the eight corners.  Error budget per row, in units of
// E_r = sum_k |M[r][k]|*max|w_k| and u = 2^-24:
//   * the value the exact path computes (fp64 sum rounded to fp32) differs from
//     the real a_r by <= 1u*E_r (+ fp64 dust);
//   * a corner evaluated here as base + deltas (3 fma, <= 3 mul of a rounded
//     difference: 4u each, <= 3 add: 1u each) differs from the real corner value by
//     <= 18u*E_r;
//   total <= 19u*E_r; eps_r = 2^-19*E_r = 32u*E_r is used.
// The quotient: |u_computed - u_real| <= (eps_0 + |u| eps_2)/(|a_2| - eps_2)
// plus the roundings of the exact path's divide (1u|u|), of rcp+mul here (<3u|u|)
// and of the bound arithmetic below (<2u|u|): 2^-20|u| = 16u|u| and an absolute
// 2^-12 cover them.  roundf(t) lies in [t-0.5, t+0.5].
__device__ inline int classify_box(const float *__restrict__ M, const BoxW b, int W, int H,
                                   const int *__restrict__ sat) {
    const float dy = b.wy1 - b.wy0, dx = b.wx1 - b.wx0, dz = b.wz1 - b.wz0;
    const float ay = fmaxf(fabsf(b.wy0), fabsf(b.wy1));
    const float ax = fmaxf(fabsf(b.wx0), fabsf(b.wx1));
    const float az = fmaxf(fabsf(b.wz0), fabsf(b.wz1));
    float a[3][8], E[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float m0 = M[4 * r], m1 = M[4 * r + 1], m2 = M[4 * r + 2], m3 = M[4 * r + 3];
        E[r] = fabsf(m0) * ay + fabsf(m1) * ax + fabsf(m2) * az + fabsf(m3);
        const float base = fmaf(m0, b.wy0, fmaf(m1, b.wx0, fmaf(m2, b.wz0, m3)));
        const float ey = m0 * dy, ex = m1 * dx, ez = m2 * dz;
        a[r][0] = base;
        a[r][1] = base + ey;
        a[r][2] = base + ex;
        a[r][3] = a[r][1] + ex;
#pragma unroll
        for (int c = 0; c < 4; ++c) a[r][4 + c] = a[r][c] + ez;
    }
    float umin = INFINITY, umax = -INFINITY, vmin = INFINITY, vmax = -INFINITY;
    float cmin = INFINITY, cmax = -INFINITY;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        cmin = fminf(cmin, a[2][c]);
        cmax = fmaxf(cmax, a[2][c]);
        const float rc = __builtin_amdgcn_rcpf(a[2][c]);
        const float fu = a[0][c] * rc, fv = a[1][c] * rc;
        umin = fminf(umin, fu);
        umax = fmaxf(umax, fu);
        vmin = fminf(vmin, fv);
        vmax = fmaxf(vmax, fv);
    }
    const float k19 = 1.9073486328125e-06f;  // 2^-19
    const float k20 = 9.5367431640625e-07f;  // 2^-20
    const float k12 = 2.44140625e-04f;       // 2^-12
    const float eps0 = E[0] * k19, eps1 = E[1] * k19, eps2 = E[2] * k19;
    const float cabs = (cmin > 0.f) ? cmin : ((cmax < 0.f) ? -cmax : 0.f);
    if (!(cabs > 8.f * eps2 + 1e-30f)) return kClsMixed;  // the denominator may vanish
    // every voxel of the box then has |a2| >= cabs - eps2 > 0; kFastDiv: all row values
    // lie in the range where divide2_shared_rcp equals the IEEE quotient (or the
    // difference cannot matter): 2^-59 <= |a2|, and |a_r| <= 2^59
    const int fast = (cabs >= 1.8e-18f && E[0] <= 5.7e17f && E[1] <= 5.7e17f && E[2] <= 5.7e17f)
                         ? kFastDiv
                         : 0;
    const float Ua = fmaxf(fabsf(umin), fabsf(umax));
    const float Va = fmaxf(fabsf(vmin), fabsf(vmax));
    if (!(Ua < 1.0e6f && Va < 1.0e6f)) return kClsMixed | fast;  // also NaN
    const float rden = 1.0001f / (cabs - eps2);
    const float mu = (eps0 + Ua * eps2) * rden + Ua * k20 + k12;
    const float mv = (eps1 + Va * eps2) * rden + Va * k20 + k12;
    const int pxlo = (int)ceilf(umin - mu - 0.5f);
    const int pxhi = (int)floorf(umax + mu + 0.5f);
    const int pylo = (int)ceilf(vmin - mv - 0.5f);
    const int pyhi = (int)floorf(vmax + mv + 0.5f);
    if (pxhi < 0 || pxlo >= W || pyhi < 0 || pylo >= H) return kClsOut;
    if (pxlo < 0 || pxhi >= W || pylo < 0 || pyhi >= H) return kClsMixed | fast;
    const int S = W + 1;
    const int cnt = sat[(pyhi + 1) * S + pxhi + 1] - sat[pylo * S + pxhi + 1] -
                    sat[(pyhi + 1) * S + pxlo] + sat[pylo * S + pxlo];
    if (cnt == 0) return kClsCarved;
    const int area = (pxhi - pxlo + 1) * (pyhi - pylo + 1);
    return (cnt == area) ? kClsFg : (kClsMixed | fast);
}

// Pre-pass over coarse tiles of 64 x 32 x 32 voxels (64 sub-tiles each; striped
// slabs use 64 x 64 x 8 so that a coarse tile stays inside one stripe): one wave
// per coarse tile, lane i = view i.  A view that sees the whole coarse box as
// background decides all 64 sub-tiles at once; views that are "outside" or "all
// foreground" for the coarse box are that for every sub-tile too, so the main
// kernel re-classifies only the views left in the coarse "mixed" mask.
__global__ __launch_bounds__(256) void carve_coarse_kernel(const CarveParams p) {
    const int ct = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int ncoarse = p.coarseX * p.coarseY * p.coarseZ;
    if (ct >= ncoarse) return;
    const int cx = ct % p.coarseX;
    const int cy = (ct / p.coarseX) % p.coarseY;
    const int cz = ct / (p.coarseX * p.coarseY);
    const int cyN = 8 << p.cyShift, czN = 8 << p.czShift;
    const int x0 = cx * kCoarseX, y0 = cy * cyN, z0 = cz * czN;
    // (striped slabs: the box spans the foreign planes in between as well -- conservative)
    const BoxW box = make_box(p.s, x0, min(x0 + kCoarseX - 1, p.X - 1), y0,
                              min(y0 + cyN - 1, p.Y - 1), global_z(p, z0),
                              global_z(p, min(z0 + czN - 1, p.Z - 1)));
    bool any_carved = false, any_mixed = false, any_fg = false;
    int chunk = 0;
    for (int vc = p.v0; vc < p.v1; vc += 64, ++chunk) {
        const int myv = vc + lane;
        int cls = kClsOut;
        if (myv < p.v1)
            cls = classify_box(p.M + 12 * myv, box, p.W, p.H, p.sat + (size_t)myv * p.satStride) &
                  3;
        const unsigned long long carved = __ballot(cls == kClsCarved);
        const unsigned long long mixed = __ballot(cls == kClsMixed);
        const unsigned long long fg = __ballot(cls == kClsFg);
        any_carved = any_carved || carved;
        any_mixed = any_mixed || mixed;
        any_fg = any_fg || fg;
        if (lane == 0) {
            p.coarseMixed[(size_t)ct * p.nchunks + chunk] = mixed;
            p.coarseFg[(size_t)ct * p.nchunks + chunk] = fg;
        }
    }
    // 1: some view carves the whole tile.  2 / 3: no view needs a closer look and none
    // carves -- every voxel keeps its occupancy and is seen (2) or not even seen (3).
    if (lane == 0)
        p.coarseCarved[ct] = any_carved ? 1 : (any_mixed ? 0 : (any_fg ? 2 : 3));
}

#ifdef ARVX_TIMELINE  // diagnostic build only (tools/timeline.py): per-workgroup start/end
struct TimelineScope {
    unsigned long long *slot;
    __device__ explicit TimelineScope(unsigned long long *base) : slot(nullptr) {
        if (threadIdx.x == 0 && base) {
            slot = base + 4ull * blockIdx.x;
            slot[0] = __builtin_amdgcn_s_memrealtime();
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            slot[2] = xcc & 0xf;
        }
    }
    __device__ ~TimelineScope() {
        if (slot) slot[1] = __builtin_amdgcn_s_memrealtime();
    }
};
#endif

template <bool kAligned4>
__global__ __launch_bounds__(256, 4) void carve_fused_kernel(const CarveParams p) {
#ifdef ARVX_TIMELINE
    TimelineScope timeline_scope(p.timeline);
#endif
    // Blocks b, b+8, b+16.. share an XCD (and its L2).  A row of tiles along x (one
    // 8x8 bundle of voxel rows) stays on one XCD, so neighbouring 64-byte runs meet
    // in one L2; rows are dealt to the 8 XCDs cyclically, which spreads the
    // expensive surface tiles evenly (contiguous z ranges per XCD left the XCDs that
    // own the empty top and bottom of the grid idle: +40 % on the sphere scene).
    const unsigned k = blockIdx.x >> 3;
    const unsigned trow = (k / p.tilesX) * 8u + (blockIdx.x & 7u);
    if (trow >= (unsigned)(p.tilesY * p.tilesZ)) return;
    const int tx = k % p.tilesX;
    const int ty = trow % p.tilesY;
    const int tz = trow / p.tilesY;

    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const bool cull = !(p.flags & 1u);
    const int ct =
        cull ? tx + p.coarseX * ((ty >> p.cyShift) + p.coarseY * (tz >> p.czShift)) : 0;
    const int code = cull ? p.coarseCarved[ct] : 0;  // workgroup-uniform (scalar load)
    const bool coarse_carved = code == 1;
    // Pure fill of a 64x8x8 tile that the pre-pass decided: carved+seen (code 1), or,
    // for a fresh model, untouched occupancy with (2) / without (3) the seen bit.
    if (kAligned4 && (coarse_carved || (code >= 2 && (p.flags & 4u))) && (p.X & 15) == 0 &&
        (tx + 1) * kTileX <= p.X) {
        // One 16-byte store per thread, 4 lanes per 64-byte row, instead of the
        // per-sub-tile layout's four dword stores per lane.
        const uint32_t v4 = code == 1 ? kDone4 : (code == 2 ? 0x03030303u : 0x01010101u);
        const int yy = ty * kTileY + ((threadIdx.x >> 2) & 7);
        const int zz = tz * kTileZ + (threadIdx.x >> 5);
        if (yy < p.Y && zz < p.Z) {
            uint8_t *dst = p.state + ((size_t)zz * p.Y + yy) * p.X + tx * kTileX +
                           16 * (threadIdx.x & 3);
            *reinterpret_cast<uint4 *>(dst) = make_uint4(v4, v4, v4, v4);
        }
        if ((p.flags & 2u) && lane == 0) {
            atomicAdd(&p.stats[0], 1ull);
            if (coarse_carved) atomicAdd(&p.stats[1], 1ull);
        }
        return;
    }
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
    for (int k = 0; k < 4; ++k) dwz[k] = (double)((float)(-global_z(p, zb + k)) * p.s);

    uint32_t st[4] = {kDone4, kDone4, kDone4, kDone4};
    bool loaded = false, all_carved = false, all_done = false;
    const size_t row = (size_t)p.X;
    const size_t plane = (size_t)p.X * p.Y;
    if (coarse_carved) {
        all_carved = true;
        if ((p.flags & 2u) && lane == 0) {
            atomicAdd(&p.stats[0], 1ull);
            atomicAdd(&p.stats[1], 1ull);
        }
    }
    const BoxW box = make_box(p.s, sx0, sx1, sy0, sy1, global_z(p, sz0), global_z(p, sz1));

    int chunk = 0;
    for (int vc = p.v0; vc < p.v1 && !all_done && !all_carved; vc += 64, ++chunk) {
        const int myv = vc + lane;
        int cls = kClsOut;
        if (myv < p.v1) {
            if (!cull) {
                cls = kClsMixed;
            } else {
                const unsigned long long cm = p.coarseMixed[(size_t)ct * p.nchunks + chunk];
                const unsigned long long cf = p.coarseFg[(size_t)ct * p.nchunks + chunk];
                if ((cf >> lane) & 1ull)
                    cls = kClsFg;  // inherited: the coarse rectangle contains this one
                else if ((cm >> lane) & 1ull)
                    cls = classify_box(p.M + 12 * myv, box, p.W, p.H,
                                       p.sat + (size_t)myv * p.satStride);
            }
        }
        const unsigned long long fastdiv = __ballot((cls & kFastDiv) != 0);
        cls &= 3;
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
                    if (p.flags & 4u) {  // fresh model: all occupied, none seen; no load
                        uint32_t w = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            w |= (uint32_t)((x + j < p.X) ? 1u : 2u) << (8 * j);
                        st[k] = w;
                    } else if (kAligned4) {
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
            const bool fast = (fastdiv >> b) & 1ull;  // wave-uniform
            const float wlim = (float)p.W - 0.5f, hlim = (float)p.H - 0.5f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (!__any(st[k] != kDone4)) continue;  // these 256 voxels are finished
                if (p.flags & 2u) {  // how many of the 256 evaluations were still open
                    const uint32_t x4 = st[k] ^ kDone4;
                    int open = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) open += ((x4 >> (8 * j)) & 0xffu) ? 1 : 0;
#pragma unroll
                    for (int d = 32; d >= 1; d >>= 1) open += __shfl_xor(open, d);
                    if (lane == 0) {
                        atomicAdd(&p.stats[5], 1ull);
                        atomicAdd(&p.stats[6], (unsigned long long)open);
                    }
                }
                const double p20 = m2[0] * dwz[k], p21 = m2[1] * dwz[k], p22 = m2[2] * dwz[k];
                uint32_t w = st[k];
                // project the four voxels first, then issue the four table reads together:
                // the loop is bound by the latency of these dependent reads, not by VALU
                int pix[4];
                bool in[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a0 = row_sum(p01[0][j], p20, p3[0]);
                    const float a1 = row_sum(p01[1][j], p21, p3[1]);
                    const float a2 = row_sum(p01[2][j], p22, p3[2]);
                    float u, v;
                    if (fast) {
                        divide2_shared_rcp(a0, a1, a2, u, v);
                    } else {
                        u = a0 / a2;
                        v = a1 / a2;
                    }
                    in[j] = pixel_from_quotients(u, v, p.W, wlim, hlim, pix[j]);
                }
                uint32_t word[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) word[j] = bgv[(unsigned)pix[j] >> 5];  // pix = 0 outside
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t isbg = in[j] ? ((word[j] >> (pix[j] & 31)) & 1u) : 0u;
                    const uint32_t seen = in[j] ? (2u << (8 * j)) : 0u;
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

// self-test support: both division forms on caller-supplied operands
__global__ __launch_bounds__(256) void selftest_divide_kernel(const float *__restrict__ a0,
                                                              const float *__restrict__ a1,
                                                              const float *__restrict__ b, size_t n,
                                                              float *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float u, v;
    divide2_shared_rcp(a0[i], a1[i], b[i], u, v);
    out[4 * i] = u;
    out[4 * i + 1] = v;
    out[4 * i + 2] = a0[i] / b[i];
    out[4 * i + 3] = a1[i] / b[i];
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
    int *s = sat + (size_t)v * satStride + xcol;
    const size_t ld = (size_t)(W + 1);
    int acc = 0;
    s[0] = 0;
    // 16 rows at a time: the loads of a chunk are independent of each other, so their
    // latency overlaps (a load-add-store chain per row was 120 us for 36 views)
    for (int y0 = 1; y0 <= H; y0 += 16) {
        int t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = (y0 + k <= H) ? s[(size_t)(y0 + k) * ld] : 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc += t[k];
            if (y0 + k <= H) s[(size_t)(y0 + k) * ld] = acc;
        }
    }
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

// Same, written at each stripe's place in the GLOBAL word plane (striped or
// contiguous slabs): local 8-plane group g lands at global group g*zstride+zphase.
__global__ __launch_bounds__(256) void pack_occupancy_global_kernel(
    const uint8_t *__restrict__ state, size_t plane, int Zloc, int zoff, int zstride, int zphase,
    uint32_t *__restrict__ words) {
    // plane (= X*Y) is a multiple of 32: every plane starts on a word
    const size_t n = plane * (size_t)Zloc;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const bool occ = state[i] & 1u;  // n is a multiple of 64 when plane % 64 == 0
        const unsigned long long b = __ballot(occ);
        if ((threadIdx.x & 63) == 0) {
            const int lz = (int)(i / plane);
            const size_t in_plane = i % plane;
            const int gz = zoff + (((lz >> 3) * zstride + zphase) << 3) + (lz & 7);
            uint32_t *dst = words + ((size_t)gz * plane + in_plane) / 32;
            dst[0] = (uint32_t)b;
            dst[1] = (uint32_t)(b >> 32);
        }
    }
}

}  // namespace arvx
