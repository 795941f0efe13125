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
    // the work-list and pool counters of the kernels that follow start at zero (this
    // saves a memset launch in front of every carve)
    if (blockIdx.x == 0 && p.workCount)
        for (int i = threadIdx.x; i < (kWorkLists + kPoolCounters) * kCounterStride; i += 256)
            p.workCount[i] = 0;  // poolNext follows workCount in the same allocation
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

// persistent kernels: one record per wave {start, end, items | ticks to the end of the
// first item << 32, views | ticks to the end of the first view << 32}
struct WaveTimeline {
    unsigned long long *slot;
    unsigned long long t0, items, views, first_item, first_view;
    // ticks: waiting for / reading the item, set-up until the first view, the views,
    // write-back after the last view
    unsigned long long ph[4] = {0, 0, 0, 0};
    unsigned long long mark = 0;
    __device__ void tick(int k) {
        if (slot) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (k >= 0) ph[k] += now - mark;
            mark = now;
        }
    }
    __device__ explicit WaveTimeline(unsigned long long *base)
        : slot(nullptr), t0(0), items(0), views(0), first_item(0), first_view(0) {
        if ((threadIdx.x & 63) == 0 && base) {
            slot = base + 8ull * (blockIdx.x * 4 + (threadIdx.x >> 6));
            t0 = __builtin_amdgcn_s_memrealtime();
            mark = t0;
            slot[0] = t0;
        }
    }
    __device__ void view_done() {
        if (slot && views++ == 0) first_view = __builtin_amdgcn_s_memrealtime() - t0;
    }
    __device__ void item_done() {
        if (slot && items++ == 0) first_item = __builtin_amdgcn_s_memrealtime() - t0;
    }
    __device__ ~WaveTimeline() {
        if (slot) {
            slot[1] = __builtin_amdgcn_s_memrealtime();
            slot[2] = items | (first_item << 32);
            slot[3] = views | (first_view << 32);
            slot[4] = ph[0] | (ph[1] << 32);
            slot[5] = ph[2] | (ph[3] << 32);
        }
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
            // (row-wise lane map: 4 x, one y, 4 z per lane.  The products are hoisted; the
            // sums follow row_sum's grouping, which the compiler hoists where it can)
            double p0[3], p1[3][4], p3[3], m2[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                p0[r] = (double)Mv[4 * r] * dwy;
                const double m1 = (double)Mv[4 * r + 1];
                m2[r] = (double)Mv[4 * r + 2];
                p3[r] = (double)Mv[4 * r + 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) p1[r][j] = m1 * dwx[j];
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
                    const float a0 = row_sum(p0[0], p1[0][j], p20, p3[0]);
                    const float a1 = row_sum(p0[1], p1[1][j], p21, p3[1]);
                    const float a2 = row_sum(p0[2], p1[2][j], p22, p3[2]);
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

// ---------------------------------------------------------------------------------
// Split form of the same algorithm (the default when culling is on):
//   carve_classify_kernel   every sub-tile: rectangle tests, and the whole job for the
//                           sub-tiles they decide (fill / inherit / store).  Short
//                           dependent-read chains, few registers -> 8 waves per SIMD.
//   carve_exact_kernel      only the sub-tiles with "mixed" views, pulled from work
//                           lists by persistent waves: uniform VALU-bound work.
// In the fused kernel one wave goes through both kinds of phase, and with 4 waves per
// SIMD the VALUs idle two thirds of the time while waves sit in the latency-bound
// phases (profiles/r1_kernel_v5: 34 % VALU-active at 512^3).
// ---------------------------------------------------------------------------------

struct SubTile {
    int sx0, sy0, sz0, sx1, sy1, sz1;  // voxel box (slab-local z)
    int x, y, zb;                      // this lane's 4 x-voxels, its y, its first z
    bool lane_ok;
};

__device__ __forceinline__ SubTile subtile_of(const CarveParams &p, int tx, int ty, int tz,
                                              int wave, int lane) {
    SubTile t;
    t.sx0 = tx * kTileX + wave * kSubX;
    t.sy0 = ty * kTileY;
    t.sz0 = tz * kTileZ;
    t.sx1 = min(t.sx0 + kSubX - 1, p.X - 1);
    t.sy1 = min(t.sy0 + kTileY - 1, p.Y - 1);
    t.sz1 = min(t.sz0 + kTileZ - 1, p.Z - 1);
    t.x = t.sx0 + 4 * (lane & 3);
    t.y = t.sy0 + ((lane >> 2) & 7);
    t.zb = t.sz0 + 4 * (lane >> 5);
    t.lane_ok = (t.x < p.X) && (t.y < p.Y);
    return t;
}

template <bool kAligned4>
__device__ __forceinline__ void subtile_load(const CarveParams &p, const SubTile &t,
                                             uint32_t st[4]) {
    const size_t row = (size_t)p.X, plane = (size_t)p.X * p.Y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        st[k] = kDone4;  // lanes / voxels outside the grid count as finished
        const int z = t.zb + k;
        if (t.lane_ok && z < p.Z) {
            const uint8_t *src = p.state + (size_t)z * plane + (size_t)t.y * row + t.x;
            if (p.flags & 4u) {  // fresh model: all occupied, none seen; no load
                uint32_t w = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w |= (uint32_t)((t.x + j < p.X) ? 1u : 2u) << (8 * j);
                st[k] = w;
            } else if (kAligned4) {
                st[k] = *reinterpret_cast<const uint32_t *>(src);
            } else {
                uint32_t w = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    w |= (uint32_t)((t.x + j < p.X) ? src[j] : (uint8_t)2) << (8 * j);
                st[k] = w;
            }
        }
    }
}

template <bool kAligned4>
__device__ __forceinline__ void subtile_store(const CarveParams &p, const SubTile &t,
                                              const uint32_t st[4]) {
    const size_t row = (size_t)p.X, plane = (size_t)p.X * p.Y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int z = t.zb + k;
        if (t.lane_ok && z < p.Z) {
            uint8_t *dst = p.state + (size_t)z * plane + (size_t)t.y * row + t.x;
            if (kAligned4) {
                *reinterpret_cast<uint32_t *>(dst) = st[k];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (t.x + j < p.X) dst[j] = (uint8_t)(st[k] >> (8 * j));
            }
        }
    }
}

// (Tried and dropped, 1024^3: the pure fill as a kernel of its own with a few fat
// workgroups -- alone it reaches twice the write rate of the fill branch below, but after
// it the classification still takes 190 us by itself (it is bound by arithmetic, and the
// branch's stores hide under it: 311 us together); run beside the classification on a
// second stream, both slow down (388 us): the classification lives on memory latency too.
// A fixed grid walking a compacted list of the undecided tiles classifies them in 127 us,
// but fill (227 us as a row-walking kernel) + list + classification is still more than
// the 311 us of this kernel, in which the stores hide under the classification.  Four tiles
// along x per workgroup (a quarter of the launches, whole 256-byte lines per fill store):
// 399 us -- the kernel is not bound by its launches but by the ~5 us chain of dependent reads
// of every undecided sub-tile at 8 waves per SIMD, and fatter workgroups lengthen that chain.)
template <bool kAligned4>
__global__ __launch_bounds__(256, 8) void carve_classify_kernel(const CarveParams p) {
    // Rows of tiles (along x) are dealt to the XCDs as in carve_fused_kernel, but the tile
    // comes straight from a 3-D block index: grid (8 * tilesX, ceil(tilesY / 8), tilesZ), the
    // dispatcher walks x fastest and hands consecutive workgroups to consecutive XCDs, so
    // blockIdx.x & 7 is the XCD and the eight rows ty = 8 * blockIdx.y + 0..7 run side by side.
    // (With a flat index every wave spent ~100 scalar instructions on two integer divisions;
    // a million one-store fill waves at 1024^3 kept the scalar units busy for longer than
    // the stores take.)
    const int tx = (int)(blockIdx.x >> 3);
    const int ty = (int)(blockIdx.y * 8u + (blockIdx.x & 7u));
    const int tz = (int)blockIdx.z;
    if (ty >= p.tilesY) return;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int ct = tx + p.coarseX * ((ty >> p.cyShift) + p.coarseY * (tz >> p.czShift));
    const int code = p.coarseCarved[ct];
    if (kAligned4 && (code == 1 || (code >= 2 && (p.flags & 4u))) && (p.X & 15) == 0 &&
        (tx + 1) * kTileX <= p.X) {  // pure fill, as in carve_fused_kernel
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
            if (code == 1) atomicAdd(&p.stats[1], 1ull);
        }
        return;
    }
    if (tx * kTileX + wave * kSubX >= p.X) return;  // wave-uniform
    const SubTile t = subtile_of(p, tx, ty, tz, wave, lane);
    const BoxW box = make_box(p.s, t.sx0, t.sx1, t.sy0, t.sy1, global_z(p, t.sz0),
                              global_z(p, t.sz1));
    bool any_carved = code == 1, any_fg = false, any_mixed = false;
    unsigned long long mixed_c[kMaxChunks], fast_c[kMaxChunks];
#pragma unroll
    for (int chunk = 0; chunk < kMaxChunks; ++chunk) {  // unrolled: the arrays stay in SGPRs
        mixed_c[chunk] = fast_c[chunk] = 0;
        const int vc = p.v0 + 64 * chunk;
        if (vc >= p.v1 || any_carved) continue;
        const int myv = vc + lane;
        int cls = kClsOut;
        if (myv < p.v1) {
            const unsigned long long cm = p.coarseMixed[(size_t)ct * p.nchunks + chunk];
            const unsigned long long cf = p.coarseFg[(size_t)ct * p.nchunks + chunk];
            // (one value from both words: their loads go out together, not one per branch)
            unsigned sel = (unsigned)((cf >> lane) & 1ull) | ((unsigned)((cm >> lane) & 1ull) << 1);
            // the lane's matrix is requested together with the masks, not after them
            float Mr[12];
            const float4 *Mp = reinterpret_cast<const float4 *>(p.M + 12 * myv);
            float4 m0 = Mp[0], m1 = Mp[1], m2 = Mp[2];
            // (or the compiler moves every load back behind the branch that needs it)
            asm volatile("" : "+v"(sel), "+v"(m0.x), "+v"(m1.x), "+v"(m2.x));
            Mr[0] = m0.x; Mr[1] = m0.y; Mr[2] = m0.z; Mr[3] = m0.w;
            Mr[4] = m1.x; Mr[5] = m1.y; Mr[6] = m1.z; Mr[7] = m1.w;
            Mr[8] = m2.x; Mr[9] = m2.y; Mr[10] = m2.z; Mr[11] = m2.w;
            if (sel & 1u)
                cls = kClsFg;  // inherited: the coarse rectangle contains this one
            else if (sel & 2u)
                cls = classify_box(Mr, box, p.W, p.H, p.sat + (size_t)myv * p.satStride);
        }
        fast_c[chunk] = __ballot((cls & kFastDiv) != 0);
        cls &= 3;
        mixed_c[chunk] = __ballot(cls == kClsMixed);
        any_carved = __ballot(cls == kClsCarved) != 0;
        any_fg = any_fg || __ballot(cls == kClsFg) != 0;
        any_mixed = any_mixed || mixed_c[chunk] != 0;
        if ((p.flags & 2u) && lane == 0) {
            atomicAdd(&p.stats[2], (unsigned long long)__popcll(mixed_c[chunk]));
            atomicAdd(&p.stats[3], (unsigned long long)min(64, p.v1 - vc));
        }
    }
    if ((p.flags & 2u) && lane == 0) {
        atomicAdd(&p.stats[0], 1ull);
        if (any_carved) atomicAdd(&p.stats[1], 1ull);
    }
    uint32_t st[4];
    if (any_carved) {  // carved implies seen (src/VoxelCarving.cpp:50-54): no load needed
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) st[kk] = kDone4;
        subtile_store<kAligned4>(p, t, st);
    } else if (!any_mixed) {
        subtile_load<kAligned4>(p, t, st);
        if (any_fg) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) st[kk] |= kDone4;  // seen, src/VoxelCarving.cpp:54
        }
        subtile_store<kAligned4>(p, t, st);
    } else if ((p.flags & 12u) == 12u) {
        // the exact kernel may hand this sub-tile to several waves that merge their results
        // with atomics: the plane must hold the sub-tile's initial state (a fresh model exists
        // only as a flag until now)
        subtile_load<kAligned4>(p, t, st);
        subtile_store<kAligned4>(p, t, st);
    }
    if (!any_carved && any_mixed && lane == 0) {
        // hand the sub-tile to carve_exact_kernel: where it is, whether some view sees
        // all of it, and per chunk of 64 views which ones to evaluate (and how to divide)
        // spread the sub-tiles evenly: with the list taken from the block index the lists
        // of some tile rows hold most of the surface and the others are empty from the start
        // (consecutive sub-tiles go to consecutive lists: no list can get more than its
        // share of all sub-tiles, which is what the host sizes the lists for)
        // Eight weight classes of eight lists, the longest items (most views to evaluate)
        // first: the waves start on those, and what the kernel ends on are the short ones.
        int nmixed = 0;
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c)
            if (c < p.nchunks) nmixed += __popcll(mixed_c[c]);
        const int wclass = 7 - min(7, nmixed * 8 / (p.v1 - p.v0 + 1));
        const int cls = wclass * 8 + (((((tz * p.tilesY + ty) * p.tilesX + tx) << 2) + wave) & 7);
        const int pos = atomicAdd(&p.workCount[cls * kCounterStride], 1);
        const size_t it = (size_t)cls * p.workCap + pos;
        p.itemInfo[it] = (unsigned long long)tx | ((unsigned long long)ty << 16) |
                         ((unsigned long long)tz << 32) | ((unsigned long long)wave << 48) |
                         ((unsigned long long)(any_fg ? 1 : 0) << 50);
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c)
            if (c < p.nchunks) {
                p.itemMasks[(it * p.nchunks + c) * 2] = mixed_c[c];
                p.itemMasks[(it * p.nchunks + c) * 2 + 1] = fast_c[c];
            }
    }
}

// One view applied exactly to the 16 voxels of every lane.  Returns true when all
// 1024 voxels of the sub-tile are carved and seen.
__device__ __forceinline__ bool exact_view(const CarveParams &p, const int view, const bool fast,
                                           const double dwy, const double (&dwx)[4],
                                           const double (&dwz)[4], uint32_t (&st)[4],
                                           const int lane) {
    const float *__restrict__ Mv = p.M + 12 * view;
    const uint32_t *__restrict__ bgv = p.bg + (size_t)view * p.bgWords;
    double p0[3], p1[3][4], p3[3], m2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        p0[r] = (double)Mv[4 * r] * dwy;
        const double m1 = (double)Mv[4 * r + 1];
        m2[r] = (double)Mv[4 * r + 2];
        p3[r] = (double)Mv[4 * r + 3];
#pragma unroll
        for (int j = 0; j < 4; ++j) p1[r][j] = m1 * dwx[j];
    }
    const float wlim = (float)p.W - 0.5f, hlim = (float)p.H - 0.5f;
    bool saw_bg = false, saw_other = false;  // ARVX_CARVE_STATS only
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!__any(st[k] != kDone4)) continue;  // these 256 voxels are finished
        if (p.flags & 2u) {
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
        int pix[4];
        bool in[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a0 = row_sum(p0[0], p1[0][j], p20, p3[0]);
            const float a1 = row_sum(p0[1], p1[1][j], p21, p3[1]);
            const float a2 = row_sum(p0[2], p1[2][j], p22, p3[2]);
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
            if (p.flags & 2u) {
                const bool open = ((st[k] >> (8 * j)) & 0xffu) != 2u;  // not carved-and-seen yet
                saw_bg = saw_bg || (open && isbg);
                saw_other = saw_other || (open && !isbg);
            }
        }
        st[k] = w;
    }
    if (p.flags & 2u) {  // pairs whose evaluated voxels all got the same answer
        const bool uniform = !(__any(saw_bg) && __any(saw_other));
        if (lane == 0 && uniform) atomicAdd(&p.stats[7], 1ull);
    }
    return __all(st[0] == kDone4 && st[1] == kDone4 && st[2] == kDone4 && st[3] == kDone4);
}

// Work distribution of the persistent exact kernels.  carve_classify_kernel appends the
// sub-tiles that need exact work to kWorkLists lists (one padded counter each: a single
// atomic word sustains only ~90 appends per microsecond, and counters sharing a line
// serialise; the lists form eight weight classes, long items first, and sub-tile i goes
// to list i % 8 of its class).  Every wave first takes ONE item of that concatenation by
// its own index -- the start of the kernel needs no atomic at all -- and then draws the
// rest one by one from a shared pool (ticket counters): the waves come back at different
// times, so the counters are not crowded; when its counters run past the end it leaves.
// (Measured before this, with waves pulling EVERY item with atomics from shared lists, in
// several arrangements -- a walk over the lists, snapshots of all counters, weight
// classes: all 4096 waves queue on the same few lines for ~20 us at the start, and spend
// tens of microseconds at the end finding out that nothing is left; a failed pull costs
// ~10 us when a thousand waves go for the same list.  A fully static split by the
// number of views per item removed that but left the waves unevenly loaded: the cost of
// an item is not known before it has run.)
// kSplit: when there are few items per wave the kernel is bound by its longest item (the
// views of an item run one after the other), so every item is handed out as 2, 4 or 8 units,
// each taking every 2nd / 4th / 8th of its views; body gets (item, part, list, log2 parts) and
// merges what the parts find (flags bit3: the caller allows it).
// srank / nstatic: this wave's rank among the nstatic waves that take an item by index
// (srank < 0: none for this wave).
template <bool kSplit, class Body>
__device__ __forceinline__ void for_each_work_item(const CarveParams &p, const int lane,
                                                   const int srank, const int nstatic,
                                                   Body body) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    // inclusive prefix of the list fill counts, list l in lane l
    int incl = (lane < kWorkLists) ? p.workCount[lane * kCounterStride] : 0;
#pragma unroll
    for (int d = 1; d < kWorkLists; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    const int items = __builtin_amdgcn_readlane(incl, kWorkLists - 1);
    int shift = 0;
    if (kSplit && (p.flags & 8u)) {
        // only while every unit still gets a wave of its own: the parts of an item do not see
        // each other's carving, so together they evaluate more voxels than one wave would.
        // (Sphere scene, 36 views, whole / 2 / 4 / 8 parts: 64^3 0.109 / 0.068 / 0.045 /
        // 0.035 ms, 128^3 0.103 / 0.065 / 0.047 / 0.048, 192^3 0.093 / 0.068 / 0.058 / 0.115,
        // 256^3 0.094 / 0.076 / 0.114, 320^3 0.097 / 0.118: from there on there are more
        // items than half the waves.)
        if (16 * items <= p.nwaves)
            shift = 3;
        else if (4 * items <= p.nwaves)
            shift = 2;
        else if (2 * items <= p.nwaves)
            shift = 1;
    }
    const int T = items << shift;  // units
    auto run = [&](int u) {  // flat unit u -> item f -> its place in the lists
        const int f = u >> shift;
        const int l = __popcll(__ballot(lane < kWorkLists && incl <= f));
        const int start = l ? __builtin_amdgcn_readlane(incl, l - 1) : 0;
        body((size_t)l * p.workCap + (f - start), u & ((1 << shift) - 1), l, shift);
    };
    // the pool: flat items pool0 + k + kPoolCounters * ticket, counter k
    const int pool0 = min(T, nstatic);
    const int P = T - pool0;  // units in the pool
    // A pool that outlasts the first items (4 P > waves): the waves come back at different
    // times, draw from their home counter until it is empty, try ONE more and leave.  (Every
    // counter is drained by its own 512 home waves whatever the others do, and the items are
    // dealt to the counters round-robin, so they run dry together.  Trying all eight counters
    // before leaving -- eight failed draws per wave on lines that all waves want -- cost 6 %
    // at 512^3, 9 % at 448^3, 3 % at 1024^3; trying 1, 2 or 4 measures the same.)
    // A small pool (slabs, grids around 400^3) or none (smaller grids): nearly every wave
    // comes back to find nothing left, and finding that out by drawing from eight counters
    // is eight returning atomics on lines every wave wants -- 4096 waves x 8 failed draws
    // took 45 us, during which the waves still at work wait behind them.  There a wave LOOKS
    // at all eight counters with one load first and draws only from one that still holds
    // tickets; without any it leaves.  (For the large pools the same scheme is slower:
    // 512^3 +16 %, 1024^3 +3 %.)
    const bool walk = 4 * P > p.nwaves;
    int u = (srank >= 0 && srank < T) ? srank : -1;  // the first unit: by index, no atomic
    int k = w & (kPoolCounters - 1), tried = 0;
    for (;;) {  // (one loop, so that the body exists once: the kernel is at its register budget)
        if (u < 0) {
            if (P <= 0) break;
            if (walk) {
                while (tried < 2) {
                    int ticket = 0;
                    if (lane == 0) ticket = atomicAdd(&p.poolNext[k * kCounterStride], 1);
                    ticket = __builtin_amdgcn_readfirstlane(ticket);
                    const long long f = (long long)pool0 + k + (long long)kPoolCounters * ticket;
                    if (f < T) {
                        u = (int)f;
                        break;
                    }
                    k = (k + 1) & (kPoolCounters - 1);
                    ++tried;
                }
                if (u < 0) break;
            } else {
                const int drawn = lane < kPoolCounters
                                      ? __hip_atomic_load(&p.poolNext[lane * kCounterStride],
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : 0;
                // counter l holds tickets [0, nk)
                const int nk = (P - lane + kPoolCounters - 1) / kPoolCounters;
                const unsigned avail = (unsigned)__ballot(lane < kPoolCounters && drawn < nk);
                if (!avail) break;
                // the first counter with tickets at or after k, cyclically
                const unsigned rot = ((avail >> k) | (avail << (kPoolCounters - k))) &
                                     ((1u << kPoolCounters) - 1u);
                k = (k + __ffs((int)rot) - 1) & (kPoolCounters - 1);
                int ticket = 0;
                if (lane == 0) ticket = atomicAdd(&p.poolNext[k * kCounterStride], 1);
                ticket = __builtin_amdgcn_readfirstlane(ticket);
                const long long f = (long long)pool0 + k + (long long)kPoolCounters * ticket;
                if (f >= T) continue;  // somebody was faster: look again
                u = (int)f;
            }
        }
        run(u);
        u = -1;
    }
}

template <bool kAligned4>
__global__ __launch_bounds__(256, 4) void carve_exact_kernel(const CarveParams p) {
#ifdef ARVX_TIMELINE
    WaveTimeline wave_timeline(p.timeline);
#endif
    const int lane = threadIdx.x & 63;
    for_each_work_item<false>(p, lane, blockIdx.x * 4 + (threadIdx.x >> 6), p.nwaves,
                          [&](const size_t it, const int, const int, const int) {
            const unsigned long long info = p.itemInfo[it];
            const int tx = (int)(info & 0xffffu), ty = (int)((info >> 16) & 0xffffu);
            const int tz = (int)((info >> 32) & 0xffffu), wave = (int)((info >> 48) & 3u);
            const SubTile t = subtile_of(p, tx, ty, tz, wave, lane);
            const double dwy = (double)((float)t.y * p.s);
            double dwx[4], dwz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) dwx[j] = (double)((float)(t.x + j) * p.s);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                dwz[k] = (double)((float)(-global_z(p, t.zb + k)) * p.s);
            uint32_t st[4];
            subtile_load<kAligned4>(p, t, st);
            if ((info >> 50) & 1ull) {
#pragma unroll
                for (int k = 0; k < 4; ++k) st[k] |= kDone4;  // seen by an all-foreground view
            }
            bool done = false;
            for (int c = 0; c < p.nchunks && !done; ++c) {
                unsigned long long mixed = p.itemMasks[(it * p.nchunks + c) * 2];
                const unsigned long long fastdiv = p.itemMasks[(it * p.nchunks + c) * 2 + 1];
                while (mixed && !done) {
                    const int b = __ffsll((long long)mixed) - 1;
                    mixed &= mixed - 1;
                    done = exact_view(p, p.v0 + 64 * c + b, (fastdiv >> b) & 1ull, dwy, dwx, dwz,
                                      st, lane);
#ifdef ARVX_TIMELINE
                    wave_timeline.view_done();
#endif
                }
            }
#ifdef ARVX_TIMELINE
            wave_timeline.item_done();
#endif
            subtile_store<kAligned4>(p, t, st);
    });
}

// ---- exact kernel, block mapping ---------------------------------------------------------
//
// Same work lists, same arithmetic, another lane <-> voxel map.  Above, one pass of the
// wave covers 256 voxels spread over two whole z planes of the sub-tile, and a pass can
// be skipped only when all of them are finished; around the hull about half of the
// voxels in the passes that do run are finished ones (carved and seen by an earlier
// view).  Here the 16 x 8 x 8 sub-tile is cut into sixteen 4 x 4 x 4 blocks and a pass
// covers ONE block, one voxel per lane: a compact block is far more often finished as a
// whole (512^3 sphere scene: 31 % fewer voxels go through the projection).
// The state plane is still read and written with the row-wise map (4-byte accesses);
// the bytes change lanes through 1 KB of LDS per wave, once per sub-tile each way.

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// view `view` on the blocks of one sub-tile.  st[byi * 2 + bzi] byte bxi = state of
// voxel (4 bxi + lx, 4 byi + ly, 4 bzi + lz) of the sub-tile.
// (Tried and dropped: consuming the table reads of one group while the next group is
// projected -- no faster, and the extra live registers spill inside the loop;
// handing out half sub-tiles (eight blocks) as the unit of work -- the tail gets shorter
// but the per-view set-up is paid twice: 27 % more view evaluations, 4 % slower.)
__device__ __forceinline__ bool exact_view_blocks(const CarveParams &p, const int view,
                                                  const bool fast, const float (&wy)[2],
                                                  const float (&wx)[4], const float (&wz)[2],
                                                  uint32_t (&st)[4]) {
    const uint32_t *__restrict__ bgv = p.bg + (size_t)view * p.bgWords;
    // The matrix of a view is the same for every lane and never written by a kernel:
    // it is fetched through the scalar cache into scalar registers (the compiler itself
    // issues vector loads here, a full memory round trip at the head of every view), stays
    // there as floats and is widened where it is used -- twelve doubles per lane would not
    // fit next to the hoisted row sums.
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 row0, row1, row2;
    const float *Mv = p.M + 12 * view;
    asm volatile(
        "s_load_dwordx4 %0, %3, 0x0\n\t"
        "s_load_dwordx4 %1, %3, 0x10\n\t"
        "s_load_dwordx4 %2, %3, 0x20\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(row0), "=&s"(row1), "=&s"(row2)
        : "s"(Mv)
        : "memory");
    const float mf[3][3] = {{row0.x, row0.y, row0.z}, {row1.x, row1.y, row1.z},
                            {row2.x, row2.y, row2.z}};
    const double p3[3] = {(double)row0.w, (double)row1.w, (double)row2.w};
    const float wlim = (float)p.W - 0.5f, hlim = (float)p.H - 0.5f;
    // a voxel outside the image reads the always-zero bit behind the view's plane: "not
    // background" needs no separate masking below
    const int zero_pix = 32 * (p.bgWords - 1);
    // All sixteen blocks are projected before the table is read: ONE wait per view.  (The
    // sub-tiles that stay fully occupied pay every wait in every view, and they are what
    // the kernel ends on.)
    int pix[4][4];
    bool in[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            in[m][j] = false;
            pix[m][j] = zero_pix;
        }
    auto project = [&](const int m, const int j, const double s0, const double s1,
                       const double s2) {
        const float a0 = (float)s0, a1 = (float)s1, a2 = (float)s2;
        float u, v;
        if (fast) {
            divide2_shared_rcp(a0, a1, a2, u, v);
        } else {
            u = a0 / a2;
            v = a1 / a2;
        }
        in[m][j] = pixel_from_quotients(u, v, p.W, wlim, hlim, pix[m][j], zero_pix);
    };
#ifndef ARVX_ASSOC_LEFT
    // a_r = p0[y] + ((p1[x] + p2[z]) + p3) (row_sum): the inner sum q depends on x and z
    // only, so it is formed once per (x, z) of the lane -- 4 x 2 values per row -- and a voxel
    // costs ONE fp64 add per row.  p1 is an exact product, so fma(m1, wx, p2) IS
    // round(p1 + p2).
    double p0[2][3];
#pragma unroll
    for (int byi = 0; byi < 2; ++byi)
#pragma unroll
        for (int r = 0; r < 3; ++r) p0[byi][r] = (double)mf[r][0] * (double)wy[byi];
#pragma unroll
    for (int bzi = 0; bzi < 2; ++bzi) {
        if (!__any(st[bzi] != kDone4 || st[2 + bzi] != kDone4)) continue;
        double q[3][4];
        {
            const double dwz = (double)wz[bzi];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double p2 = (double)mf[r][2] * dwz;
                const double m1 = (double)mf[r][1];
#pragma unroll
                for (int j = 0; j < 4; ++j) q[r][j] = fma(m1, (double)wx[j], p2) + p3[r];
            }
        }
#pragma unroll
        for (int byi = 0; byi < 2; ++byi) {
            const int m = 2 * byi + bzi;
            const uint32_t w = st[m];
            if (!__any(w != kDone4)) continue;  // these four blocks are finished
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!__any(((w >> (8 * j)) & 0xffu) != 2u)) continue;  // block j is finished
                project(m, j, p0[byi][0] + q[0][j], p0[byi][1] + q[1][j], p0[byi][2] + q[2][j]);
            }
        }
    }
#else
    // a_r = ((p0[y] + p1[x]) + p2[z]) + p3: the inner sum depends on y and x
#pragma unroll
    for (int byi = 0; byi < 2; ++byi) {
        if (!__any(st[2 * byi] != kDone4 || st[2 * byi + 1] != kDone4)) continue;
        double p01[3][4];
        {
            const double dwy = (double)wy[byi];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double p0 = (double)mf[r][0] * dwy;
                const double m1 = (double)mf[r][1];
#pragma unroll
                for (int j = 0; j < 4; ++j) p01[r][j] = fma(m1, (double)wx[j], p0);
            }
        }
#pragma unroll
        for (int bzi = 0; bzi < 2; ++bzi) {
            const int m = 2 * byi + bzi;
            const uint32_t w = st[m];
            if (!__any(w != kDone4)) continue;  // these four blocks are finished
            const double dwz = (double)wz[bzi];
            const double p20 = (double)mf[0][2] * dwz, p21 = (double)mf[1][2] * dwz,
                         p22 = (double)mf[2][2] * dwz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!__any(((w >> (8 * j)) & 0xffu) != 2u)) continue;  // block j is finished
                project(m, j, (p01[0][j] + p20) + p3[0], (p01[1][j] + p21) + p3[1],
                        (p01[2][j] + p22) + p3[2]);
            }
        }
    }
#endif
    uint32_t word[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j) word[m][j] = bgv[(unsigned)pix[m][j] >> 5];  // pix = 0 if skipped
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        uint32_t w = st[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t isbg = __builtin_amdgcn_ubfe(word[m][j], (uint32_t)pix[m][j], 1u);
            const uint32_t seen = in[m][j] ? (2u << (8 * j)) : 0u;
            w = (w | seen) & ~(isbg << (8 * j));
        }
        st[m] = w;
    }
    return __all(st[0] == kDone4 && st[1] == kDone4 && st[2] == kDone4 && st[3] == kDone4);
}

template <bool kAligned4>
__global__ __launch_bounds__(256, 4) void carve_exact_blocks_kernel(const CarveParams p) {
#ifdef ARVX_TIMELINE
    WaveTimeline wave_timeline(p.timeline);
#endif
    __shared__ uint32_t xpose[4][256];  // one sub-tile of state bytes per wave, [z][y][x]
    const int lane = threadIdx.x & 63;
    uint32_t *buf = xpose[threadIdx.x >> 6];
    uint8_t *buf8 = reinterpret_cast<uint8_t *>(buf);
    // row-wise map (subtile_of): 4 x-voxels, one y, 4 z per lane
    const int rowSlot = ((4 * (lane >> 5)) * 8 + ((lane >> 2) & 7)) * 4 + (lane & 3);  // + 32 k
    // block map: one voxel per block
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    // every fourth workgroup starts with the fill and joins the exact work afterwards;
    // the others end with whatever is left of the fill
    // (Tried and dropped: leaving the pure fill of the decided tiles to a quarter of these
    // workgroups so that it overlaps the exact work -- the fill saturates HBM and the
    // exact waves, which live on memory latency, slow down by more than the fill costs.)
    for_each_work_item<kAligned4>(p, lane, blockIdx.x * 4 + (threadIdx.x >> 6), p.nwaves,
                          [&](const size_t it, const int part, const int list, const int pshift) {
            // the kernel ends on its longest items (an item's views run one after the other):
            // the items of the heavy weight classes get the SIMD's issue slots first
            // (512^3: -1.2 %, 1024^3: no change; the waves of a SIMD mostly hold items of
            // similar weight, and a view is bound by the SIMD's issue rate either way)
            switch (list >> 4) {
                case 0: __builtin_amdgcn_s_setprio(3); break;
                case 1: __builtin_amdgcn_s_setprio(2); break;
                case 2: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
            }
            const unsigned long long info = p.itemInfo[it];
            // (the first chunk's view masks are requested with the item, not after its set-up)
            unsigned long long mixed0 = p.itemMasks[it * p.nchunks * 2];
            unsigned long long fast0 = p.itemMasks[it * p.nchunks * 2 + 1];
#ifdef ARVX_TIMELINE
            { volatile unsigned long long sink = info; (void)sink; }
            wave_timeline.tick(0);  // since the end of the previous item: the pull + this read
#endif
            const int tx = (int)(info & 0xffffu), ty = (int)((info >> 16) & 0xffffu);
            const int tz = (int)((info >> 32) & 0xffffu), wave = (int)((info >> 48) & 3u);
            const SubTile t = subtile_of(p, tx, ty, tz, wave, lane);
            uint32_t st[4];
            subtile_load<kAligned4>(p, t, st);
            if ((info >> 50) & 1ull) {
#pragma unroll
                for (int k = 0; k < 4; ++k) st[k] |= kDone4;  // seen by an all-foreground view
            }
            // rows -> blocks
#pragma unroll
            for (int k = 0; k < 4; ++k) buf[rowSlot + 32 * k] = st[k];
            wave_lds_sync();
#pragma unroll
            for (int byi = 0; byi < 2; ++byi)
#pragma unroll
                for (int bzi = 0; bzi < 2; ++bzi) {
                    uint32_t w = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        w |= (uint32_t)buf8[((4 * bzi + lz) * 8 + 4 * byi + ly) * 16 + 4 * j + lx]
                             << (8 * j);
                    st[2 * byi + bzi] = w;
                }
            // world coordinates as the reference's toWord gives them (fp32); widened to
            // double where the products are formed
            float wx[4], wy[2], wz[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) wx[j] = (float)(t.sx0 + 4 * j + lx) * p.s;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                wy[b] = (float)(t.sy0 + 4 * b + ly) * p.s;
                wz[b] = (float)(-global_z(p, t.sz0 + 4 * b + lz)) * p.s;
            }
            bool done = false;
#ifdef ARVX_TIMELINE
            wave_timeline.tick(1);
#endif
            for (int c = 0; c < p.nchunks && !done; ++c) {
                // the same words in every lane: keep them, and the view loop, scalar
                unsigned long long mixed =
                    uniform64(c ? p.itemMasks[(it * p.nchunks + c) * 2] : mixed0);
                const unsigned long long fastdiv =
                    uniform64(c ? p.itemMasks[(it * p.nchunks + c) * 2 + 1] : fast0);
                for (int nth = 0; mixed && !done; ++nth) {
                    const int b = __ffsll((long long)mixed) - 1;
                    mixed &= mixed - 1;
                    if ((nth & ((1 << pshift) - 1)) != part) continue;  // another part's view
                    done = exact_view_blocks(p, __builtin_amdgcn_readfirstlane(p.v0 + 64 * c + b),
                                             (fastdiv >> b) & 1ull, wy, wx, wz, st);
#ifdef ARVX_TIMELINE
                    wave_timeline.view_done();
#endif
                }
            }
#ifdef ARVX_TIMELINE
            wave_timeline.item_done();
            wave_timeline.tick(2);
#endif
            // blocks -> rows
            wave_lds_sync();
#pragma unroll
            for (int byi = 0; byi < 2; ++byi)
#pragma unroll
                for (int bzi = 0; bzi < 2; ++bzi)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        buf8[((4 * bzi + lz) * 8 + 4 * byi + ly) * 16 + 4 * j + lx] =
                            (uint8_t)(st[2 * byi + bzi] >> (8 * j));
            wave_lds_sync();
#pragma unroll
            for (int k = 0; k < 4; ++k) st[k] = buf[rowSlot + 32 * k];
            wave_lds_sync();
            if (kAligned4 && pshift) {
                // the parts of an item carve (clear bit0) and see (set bit1) independently:
                // both are monotone, so the order in which they reach the plane is irrelevant
                const size_t row = (size_t)p.X, plane = (size_t)p.X * p.Y;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int z = t.zb + k;
                    if (t.lane_ok && z < p.Z) {
                        uint32_t *dst = reinterpret_cast<uint32_t *>(
                            p.state + (size_t)z * plane + (size_t)t.y * row + t.x);
                        (void)__hip_atomic_fetch_and(dst, st[k] | 0xfefefefeu, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                        (void)__hip_atomic_fetch_or(dst, st[k] & 0x02020202u, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else {
                subtile_store<kAligned4>(p, t, st);
            }
#ifdef ARVX_TIMELINE
            wave_timeline.tick(3);
#endif
    });
}

// self-test support: round_pixel against std::round on every float of a bit range
__global__ __launch_bounds__(256) void selftest_round_kernel(unsigned lo, unsigned hi,
                                                             unsigned long long *__restrict__ nbad) {
    const unsigned long long i = lo + (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (i > hi) return;
    const float x = __uint_as_float((unsigned)i);
    const int want = (int)roundf(x);
    const int alt = (int)floorf(x) + (__builtin_amdgcn_fractf(x) >= 0.5f ? 1 : 0);
    if (round_pixel(x) != want || alt != want) atomicAdd(nbad, 1ull);
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

// 8 voxels per thread (8-byte load, one byte of the plane out: voxel i is bit i%8 of
// byte i/8 in the little-endian words).  Needs an 8-aligned plane and n % 8 == 0.
__global__ __launch_bounds__(256) void pack_occupancy8_kernel(const uint8_t *__restrict__ state,
                                                              size_t nbytes,
                                                              uint8_t *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nbytes) return;
    const unsigned long long s = ((const unsigned long long *)state)[t];
    out[t] = (uint8_t)(((s & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
}

// 32 voxels per thread: two 16-byte loads in, one 32-bit word of the plane out.  Needs a
// 16-aligned plane and n % 32 == 0.
__global__ __launch_bounds__(256) void pack_occupancy32_kernel(const uint8_t *__restrict__ state,
                                                               size_t nwords,
                                                               uint32_t *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nwords) return;
    const ulonglong2 a = ((const ulonglong2 *)state)[2 * t], b = ((const ulonglong2 *)state)[2 * t + 1];
    const unsigned long long one = 0x0101010101010101ull, mul = 0x0102040810204080ull;
    out[t] = (uint32_t)(((a.x & one) * mul) >> 56) | ((uint32_t)(((a.y & one) * mul) >> 56) << 8) |
             ((uint32_t)(((b.x & one) * mul) >> 56) << 16) |
             ((uint32_t)(((b.y & one) * mul) >> 56) << 24);
}

// the global (slab / striped) form of the same; plane % 64 == 0
__global__ __launch_bounds__(256) void pack_occupancy_global8_kernel(
    const uint8_t *__restrict__ state, size_t plane, int Zloc, int zoff, int zstride, int zphase,
    uint8_t *__restrict__ out) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t pbytes = plane / 8;
    if (t >= pbytes * (size_t)Zloc) return;
    const int lz = (int)(t / pbytes);
    const size_t in_plane = t % pbytes;
    const int gz = zoff + (((lz >> 3) * zstride + zphase) << 3) + (lz & 7);
    const unsigned long long s = ((const unsigned long long *)state)[t];
    out[(size_t)gz * pbytes + in_plane] =
        (uint8_t)(((s & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56);
}

}  // namespace arvx
