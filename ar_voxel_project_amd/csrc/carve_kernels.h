// carve_kernels.h -- dense silhouette carve for gfx950 (MI355X).
//
// Replaces the voxel loop of the reference's carve(): src/VoxelCarving.cpp:38-55
// (one view) and :60-72 (all views).  One arvx_carve call applies all requested views; a voxel's
// state (2 bits, records of 16 x 8 x 8 voxels: arvx_device.h) is touched at most once.
//
// Three launches, each settling what it can from RECTANGLES and handing the rest on
// (classify_box: the eight corners of a voxel box are projected, a rigorous error margin is added,
// and the pixel rectangle is looked up in the view's summed-area table of foreground pixels):
//   rectangle outside the image            -> no voxel of the box is seen by this view
//   inside, no foreground pixel            -> every voxel is carved: the box is finished
//                                             (carved implies seen, src/VoxelCarving.cpp:50-54)
//   inside, only foreground pixels         -> every voxel is seen, none carved
//   anything else                          -> "mixed": look closer
//   carve_coarse_kernel          coarse tiles of 64 x 32 x 32 voxels, a lane per (tile, view) pair; a
//                                fresh model's decided tiles are not even written (their code is
//                                their state), the undecided ones are listed
//   carve_classify_dense_kernel  the 64 sub-tiles (16 x 8 x 8) of a listed coarse tile against its
//                                mixed views, lane = (sub-tile, view slot); settled records are
//                                written, mixed sub-tiles queued as items in eight weight classes
//                                (carve_classify_kernel: a wave per sub-tile, for small grids / stats)
//   carve_exact_blocks_kernel    persistent waves take the items; per item block-level rectangle
//                                tests (4 x 4 x 4 voxels) first, then the blocks still mixed are
//                                projected voxel by voxel, one voxel of each block per lane, in the
//                                reference's own arithmetic (arvx_device.h: row sums, division,
//                                rounding), until a ballot finds all 1024 voxels carved and seen
// carve_fused_kernel is the brute-force form (every voxel in every view: ARVX_CARVE_NO_CULL, > 256
// views) and the yardstick of tests/test_carve_gpu.py; carve_coarse_fill_kernel / carve_fill_kernel
// write whole coarse tiles for models that are not fresh and for the stages that need every record.
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
// sat: the view's summed-area table of foreground pixels, satW entries per row
// (views_kernels.h): entry (Y, X) = foreground pixels in rows < Y, columns < X -- MODULO 2^16:
// the count of a rectangle comes out of the four corners exactly whenever the rectangle has
// fewer than 2^16 pixels, which is every rectangle that matters (a sub-tile's is a few hundred
// pixels); a larger one is called "mixed" without looking (conservative: the next level of
// tests, on smaller boxes, decides).  Half the bytes to derive per step and to keep in the caches.
typedef uint16_t sat_t;
// The test in two steps, so that a caller can have the table reads of several rectangles in flight
// together: rect_prepare does the arithmetic up to the pixel rectangle -- an answer that needs no
// look at the table (code >= 0: outside / mixed) or the four entries to read (code < 0) --,
// rect_finish turns the entries into the answer.
struct RectQ {
    int code;  // >= 0: the answer (kClsOut, or kClsMixed [| kFastDiv]); -1: read the table
    int fast;  // kFastDiv or 0, to be OR-ed onto a "mixed" answer
    int base;  // entry (Y0, X0)
    int dx, dy;  // X1 - X0, Y1 - Y0
};
__device__ inline RectQ rect_prepare(const float (&M)[12], const BoxW b, int W, int H, int satW) {
    RectQ q;
    q.fast = 0;
    q.base = q.dx = q.dy = 0;
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
    q.code = kClsMixed;
    if (!(cabs > 8.f * eps2 + 1e-30f)) return q;  // the denominator may vanish
    // every voxel of the box then has |a2| >= cabs - eps2 > 0; kFastDiv: all row values
    // lie in the range where divide2_shared_rcp equals the IEEE quotient (or the
    // difference cannot matter): 2^-59 <= |a2|, and |a_r| <= 2^59
    const int fast = (cabs >= 1.8e-18f && E[0] <= 5.7e17f && E[1] <= 5.7e17f && E[2] <= 5.7e17f)
                         ? kFastDiv
                         : 0;
    q.fast = fast;
    q.code = kClsMixed | fast;
    const float Ua = fmaxf(fabsf(umin), fabsf(umax));
    const float Va = fmaxf(fabsf(vmin), fabsf(vmax));
    if (!(Ua < 1.0e6f && Va < 1.0e6f)) return q;  // also NaN
    const float rden = 1.0001f / (cabs - eps2);
    const float mu = (eps0 + Ua * eps2) * rden + Ua * k20 + k12;
    const float mv = (eps1 + Va * eps2) * rden + Va * k20 + k12;
    const int pxlo = (int)ceilf(umin - mu - 0.5f);
    const int pxhi = (int)floorf(umax + mu + 0.5f);
    const int pylo = (int)ceilf(vmin - mv - 0.5f);
    const int pyhi = (int)floorf(vmax + mv + 0.5f);
    if (pxhi < 0 || pxlo >= W || pyhi < 0 || pylo >= H) {
        q.code = kClsOut;
        return q;
    }
    if (pxlo < 0 || pxhi >= W || pylo < 0 || pyhi >= H) return q;
    q.dx = pxhi + 1 - pxlo;
    q.dy = pyhi + 1 - pylo;
    if (q.dx * q.dy >= 65536) return q;  // (the table's entries are counts modulo 2^16)
    q.base = pylo * satW + pxlo;
    q.code = -1;
    return q;
}
// s00 .. s11: the table's entries (Y0, X0), (Y0, X1), (Y1, X0), (Y1, X1)
__device__ __forceinline__ int rect_finish(const RectQ &q, int s00, int s01, int s10, int s11) {
    const int cnt = (s11 - s01 - s10 + s00) & 0xffff;
    if (cnt == 0) return kClsCarved;  // no foreground in the rectangle
    return (cnt == q.dx * q.dy) ? kClsFg : (kClsMixed | q.fast);
}
__device__ inline int classify_box_m(const float (&M)[12], const BoxW b, int W, int H,
                                     const sat_t *__restrict__ sat, int satW) {
    const RectQ q = rect_prepare(M, b, W, H, satW);
    if (q.code >= 0) return q.code;
    const sat_t *e = sat + q.base;
    return rect_finish(q, e[0], e[q.dx], e[q.dy * satW], e[q.dy * satW + q.dx]);
}
// ... with the view's matrix in memory
__device__ inline int classify_box(const float *__restrict__ M, const BoxW b, int W, int H,
                                   const sat_t *__restrict__ sat, int satW) {
    float Mr[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Mr[i] = M[i];
    return classify_box_m(Mr, b, W, H, sat, satW);
}

// Pre-pass over coarse tiles of 64 x 32 x 32 voxels (64 sub-tiles each; striped
// slabs use 64 x 64 x 8 so that a coarse tile stays inside one stripe): one wave
// per coarse tile, lane i = view i.  A view that sees the whole coarse box as
// background decides all 64 sub-tiles at once; views that are "outside" or "all
// foreground" for the coarse box are that for every sub-tile too, so the main
// kernel re-classifies only the views left in the coarse "mixed" mask.
__device__ __forceinline__ void coarse_reset_counters(const CarveParams &p) {
    // the work-list and pool counters of the kernels that follow start at zero (this
    // saves a memset launch in front of every carve)
    if (blockIdx.x == 0 && p.workCount)
        for (int i = threadIdx.x; i < (kWorkLists + kPoolCounters) * kCounterStride; i += blockDim.x)
            p.workCount[i] = 0;  // poolNext follows workCount in the same allocation
    // the length of the undecided list alternates between two counters: this launch appends
    // to one (zeroed by the launch before) and zeroes the other for the next carve
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.undecidedCountNext) *p.undecidedCountNext = 0;
}

// One wave, lane i = view i: coarse tile ct against every view.  Writes the tile's mixed / fg
// view masks and returns its code (every lane): 1: some view carves the whole tile.  2 / 3: no
// view needs a closer look and none carves -- every voxel keeps its occupancy and is seen (2)
// or not even seen (3).  0: undecided.
__device__ __forceinline__ int coarse_classify(const CarveParams &p, const int ct, const int lane) {
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
            cls = classify_box(p.M + 12 * myv, box, p.W, p.H, p.sat + (size_t)myv * p.satStride,
                               p.satW) &
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
    return any_carved ? 1 : (any_mixed ? 0 : (any_fg ? 2 : 3));
}

// Up to kCoarseWaves coarse tiles per workgroup (p.coarsePerWg).  A lane is a (tile, view) PAIR:
// the workgroup's tiles x views are dealt to the lanes densely (with one wave per tile and lane =
// view, 36 views kept 36 of 64 lanes busy), the answers are OR-ed into LDS, and one thread per
// tile writes its masks and its code.  Lazy state (flags bit7): nothing else happens to a decided
// tile -- no fill launch follows --, so the undecided ones go on the classify kernels' list from
// here: collected in LDS, ONE append per workgroup (an append per tile, ~450 at 512^3 on one
// counter, was 5 of this kernel's 11 us).
constexpr int kCoarseWaves = 16;
__global__ __launch_bounds__(64 * kCoarseWaves) void carve_coarse_kernel(const CarveParams p) {
    __shared__ float s_box[kCoarseWaves][6];
    __shared__ unsigned long long s_mixed[kCoarseWaves][kMaxChunks], s_fg[kCoarseWaves][kMaxChunks];
    __shared__ int s_carved[kCoarseWaves];
    __shared__ int s_ct[kCoarseWaves];
    __shared__ int s_n, s_base;
    coarse_reset_counters(p);
    const int cw = p.coarsePerWg;
    const int ncoarse = p.coarseX * p.coarseY * p.coarseZ;
    const int t0 = blockIdx.x * cw, nt = min(cw, ncoarse - t0);  // this workgroup's tiles
    const bool listing = (p.flags & 128u) && p.undecidedList;    // workgroup-uniform
    if ((int)threadIdx.x < nt) {
        const int ct = t0 + threadIdx.x;
        const int cx = ct % p.coarseX;
        const int cy = (ct / p.coarseX) % p.coarseY;
        const int cz = ct / (p.coarseX * p.coarseY);
        const int cyN = 8 << p.cyShift, czN = 8 << p.czShift;
        const int x0 = cx * kCoarseX, y0 = cy * cyN, z0 = cz * czN;
        // (striped slabs: the box spans the foreign planes in between as well -- conservative)
        const BoxW b = make_box(p.s, x0, min(x0 + kCoarseX - 1, p.X - 1), y0,
                                min(y0 + cyN - 1, p.Y - 1), global_z(p, z0),
                                global_z(p, min(z0 + czN - 1, p.Z - 1)));
        float *o = s_box[threadIdx.x];
        o[0] = b.wy0, o[1] = b.wy1, o[2] = b.wx0, o[3] = b.wx1, o[4] = b.wz0, o[5] = b.wz1;
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) s_mixed[threadIdx.x][c] = s_fg[threadIdx.x][c] = 0ull;
        s_carved[threadIdx.x] = 0;
    }
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const int V = p.v1 - p.v0;
    const float rV = 1.0f / (float)V;
    for (int q = threadIdx.x; q < nt * V; q += blockDim.x) {
        // q = tile * V + view  (q < 16 * 256: the float quotient is off by at most one)
        int tile = (int)((float)q * rV);
        int view = q - tile * V;
        if (view < 0) {
            --tile;
            view += V;
        } else if (view >= V) {
            ++tile;
            view -= V;
        }
        const float *sb = s_box[tile];
        BoxW box;
        box.wy0 = sb[0], box.wy1 = sb[1], box.wx0 = sb[2], box.wx1 = sb[3], box.wz0 = sb[4],
        box.wz1 = sb[5];
        const int myv = p.v0 + view;
        const int cls = classify_box(p.M + 12 * myv, box, p.W, p.H, p.sat + (size_t)myv * p.satStride,
                                     p.satW) & 3;
        const unsigned long long bit = 1ull << (view & 63);
        if (cls == kClsMixed) atomicOr(&s_mixed[tile][view >> 6], bit);
        if (cls == kClsFg) atomicOr(&s_fg[tile][view >> 6], bit);
        if (cls == kClsCarved) s_carved[tile] = 1;  // (any writer writes the same)
    }
    __syncthreads();
    if ((int)threadIdx.x < nt) {
        const int ct = t0 + threadIdx.x;
        bool any_mixed = false, any_fg = false;
        for (int c = 0; c < p.nchunks; ++c) {
            const unsigned long long m = s_mixed[threadIdx.x][c], f = s_fg[threadIdx.x][c];
            p.coarseMixed[(size_t)ct * p.nchunks + c] = m;
            p.coarseFg[(size_t)ct * p.nchunks + c] = f;
            any_mixed = any_mixed || m;
            any_fg = any_fg || f;
        }
        // 1: some view carves the whole tile.  2 / 3: no view needs a closer look and none carves
        // -- every voxel keeps its occupancy and is seen (2) or not even seen (3).  0: undecided.
        const int code = s_carved[threadIdx.x] ? 1 : (any_mixed ? 0 : (any_fg ? 2 : 3));
        p.coarseCarved[ct] = (uint8_t)code;
        if (p.cstate && (p.flags & 4u)) p.cstate[ct] = code == 1 ? 3 : (code == 2 ? 2 : 0);
        if (listing && !(code == 1 || (code >= 2 && (p.flags & 4u)))) s_ct[atomicAdd(&s_n, 1)] = ct;
    }
    if (listing) {
        __syncthreads();
        if (threadIdx.x == 0 && s_n) s_base = atomicAdd(p.undecidedCount, s_n);
        __syncthreads();
        if ((int)threadIdx.x < s_n) p.undecidedList[s_base + threadIdx.x] = s_ct[threadIdx.x];
    }
}

// More than 64 * kMaxChunks views (the fused kernel's pre-pass; its masks do not fit the LDS
// arrays above): one wave per coarse tile, lane = view, chunk after chunk.
__global__ __launch_bounds__(256) void carve_coarse_wave_kernel(const CarveParams p) {
    coarse_reset_counters(p);
    const int ct = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ct >= p.coarseX * p.coarseY * p.coarseZ) return;
    const int code = coarse_classify(p, ct, threadIdx.x & 63);
    if ((threadIdx.x & 63) == 0) p.coarseCarved[ct] = (uint8_t)code;
}

// Coarse tiles that the pre-pass decided are constant: carved + seen (code 1), or, for a
// fresh model, untouched occupancy with (2) / without (3) the seen bit.  One workgroup writes
// the coarse tile's records -- 16 KB in one piece (8 KB on striped slabs).
__device__ __forceinline__ void coarse_fill(const CarveParams &p, const int ct, const int code) {
    if (!(code == 1 || (code >= 2 && (p.flags & 4u)))) {
        // what cannot be settled with a constant goes on the list of carve_classify_kernel:
        // undecided tiles, and -- on a model that is not fresh -- the ones whose voxels keep
        // their occupancy.  (Appended here, not in the pre-pass: there the atomic was one more
        // dependent round trip at the end of every wave.)
        if (threadIdx.x == 0 && p.undecidedList)
            p.undecidedList[atomicAdd(p.undecidedCount, 1)] = ct;
        return;
    }
    if (p.flags & 128u) return;  // lazy state: the tile's code says it all (arvx_device.h)
    const int cx = ct % p.coarseX, cy = (ct / p.coarseX) % p.coarseY, cz = ct / (p.coarseX * p.coarseY);
    const int tshift = p.cyShift + p.czShift;
    uint4 *dst = reinterpret_cast<uint4 *>(p.rec + (((size_t)ct << (tshift + 2)) * kRecU16));
    const int n16 = (kRecU16 * 2 / 16) << (tshift + 2);  // 16-byte pieces of the coarse tile
    const bool inside = (cx + 1) * kCoarseX <= p.X && ((cy + 1) << (3 + p.cyShift)) <= p.Y &&
                        ((cz + 1) << (3 + p.czShift)) <= p.Z;
    for (int i = threadIdx.x; i < n16; i += 256) {
        // piece i: record i / 16, half (i / 8) & 1 (occ, seen), entries 8 * (i & 7) ..
        const bool seen_half = (i >> 3) & 1;
        uint32_t w[4];
        if (code == 1 || (inside && !(code == 3 && seen_half))) {
            const uint32_t v = (code == 1) ? (seen_half ? 0xffffffffu : 0u) : 0xffffffffu;
            w[0] = w[1] = w[2] = w[3] = v;
        } else if (inside) {  // code 3, seen half of an interior tile: nothing seen
            w[0] = w[1] = w[2] = w[3] = 0u;
        } else {
            const int rec = i >> 4, tl = rec >> 2, wave = rec & 3;
            const int ty = (cy << p.cyShift) + (tl & ((1 << p.cyShift) - 1));
            const int tz = (cz << p.czShift) + (tl >> p.cyShift);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t pair = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const uint32_t in = row_inmask(p, cx, ty, tz, wave, 8 * (i & 7) + 2 * k + h);
                    // voxels outside the grid: occ 0, seen 1
                    const uint32_t v = seen_half ? (code == 2 ? 0xffffu : (~in & 0xffffu)) : in;
                    pair |= v << (16 * h);
                }
                w[k] = pair;
            }
        }
        // (non-temporal: 16 KB of constants per coarse tile that the carve itself never reads
        // again must not push the views' tables and bit planes out of the L2)
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v;
        v.x = w[0];
        v.y = w[1];
        v.z = w[2];
        v.w = w[3];
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(dst + i));
    }
}

// (flags bit4: every coarse tile as "untouched, not seen" -- a fresh model as records)
__global__ __launch_bounds__(256) void carve_fill_kernel(const CarveParams p) {
    int code = (p.flags & 16u) ? 3 : p.coarseCarved[blockIdx.x];
    if (code & kCodeWritten) code = 0;  // (written out already: rec_or_bitgrid_lazy_kernel)
    coarse_fill(p, blockIdx.x, code);
}

// Pre-pass and fill in one launch, one workgroup per coarse tile: its first wave classifies
// the tile (lane = view), then all four write its records if that settled it.  (As two
// kernels the pre-pass was a launch of its own whose waves each ended on one dependent chain
// matrix -> table entries -> code, and the fill read the codes back: 6.7 + 9.7 us at 512^3.)
__global__ __launch_bounds__(256) void carve_coarse_fill_kernel(const CarveParams p) {
    __shared__ int s_code;
    coarse_reset_counters(p);
    const int ct = blockIdx.x;
    // what earlier carves settled for the whole tile (workgroup-uniform; zero without a summary)
    const int known = (p.cstate && !(p.flags & 4u)) ? p.cstate[ct] : 0;
    if (known & 1) {  // carved and seen as a whole before: nothing a view could change
        if (threadIdx.x == 0) p.coarseCarved[ct] = 1;
        return;
    }
    if (threadIdx.x < 64) {
        const int code = coarse_classify(p, ct, threadIdx.x);
        if (threadIdx.x == 0) {
            p.coarseCarved[ct] = (uint8_t)code;
            s_code = code;
            if (p.cstate) {
                if (p.flags & 4u) p.cstate[ct] = code == 1 ? 3 : (code == 2 ? 2 : 0);
                else if (code == 1) p.cstate[ct] = 3;
                else if (code == 2) p.cstate[ct] = (uint8_t)(known | 2);  // (the classify kernel
                // of this launch stores the seen bits)
            }
        }
    }
    __syncthreads();
    // On a model that is not fresh a tile no view looks into (3), or one that some view sees as
    // a whole (2) when every voxel was seen before, has nothing left to do.
    if (!(p.flags & 4u) && (s_code == 3 || (s_code == 2 && (known & 2)))) return;
    coarse_fill(p, ct, s_code);
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

__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

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

// A sub-tile's record <-> the row-wise register form of the fused / row-mapped kernels:
// st[k] = the four x-voxels (4 q .. 4 q + 3, q = lane & 3) of row (y, zb + k) as four
// bytes, bit0 occupied, bit1 seen.
__device__ __forceinline__ void subtile_load(const CarveParams &p, const SubTile &t,
                                             const uint16_t *__restrict__ rec, int lane,
                                             uint32_t st[4]) {
    const int q = lane & 3, ry = (lane >> 2) & 7, rz0 = 4 * (lane >> 5);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (p.flags & 4u) {  // fresh model: all occupied, none seen; no load
            const int z = t.zb + k;
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                w |= (uint32_t)((t.lane_ok && z < p.Z && t.x + j < p.X) ? 1u : 2u) << (8 * j);
            st[k] = w;
        } else {
            const int r = (rz0 + k) * 8 + ry;
            const uint32_t o = rec[r], sn = rec[64 + r];
            st[k] = nibble_to_bytes((o >> (4 * q)) & 15u) | (nibble_to_bytes((sn >> (4 * q)) & 15u) << 1);
        }
    }
}

__device__ __forceinline__ void subtile_store(uint16_t *__restrict__ rec, int lane,
                                              const uint32_t st[4]) {
    const int q = lane & 3, ry = (lane >> 2) & 7, rz0 = 4 * (lane >> 5);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        uint32_t o = bytes_to_nibble(st[k] & 0x01010101u) << (4 * q);
        uint32_t sn = bytes_to_nibble((st[k] >> 1) & 0x01010101u) << (4 * q);
        uint32_t both = o | (sn << 16);  // the four lanes of a row sit next to each other
        both |= __shfl_xor(both, 1);
        both |= __shfl_xor(both, 2);
        if (q == 0) {
            const int r = (rz0 + k) * 8 + ry;
            rec[r] = (uint16_t)both;
            rec[64 + r] = (uint16_t)(both >> 16);
        }
    }
}

// carved + seen, the whole sub-tile
__device__ __forceinline__ void subtile_store_done(uint16_t *__restrict__ rec, int lane) {
    reinterpret_cast<uint32_t *>(rec)[lane] = lane < 32 ? 0u : 0xffffffffu;
}

// (carve_fused_kernel, the brute-force form, follows exact_view_blocks below: it is built on it)

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

// The whole record of a sub-tile from constants: occ = the voxels inside the grid (or none),
// seen = all (voxels outside the grid always count as seen) or only those outside.
__device__ __forceinline__ void subtile_store_const(const CarveParams &p, uint16_t *__restrict__ rec,
                                                    int lane, int tx, int ty, int tz, int wave,
                                                    bool occupied, bool seen) {
    uint32_t pair = 0;  // entries 2 * (lane & 31), + 1 of the occ (lane < 32) / seen half
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t in = row_inmask(p, tx, ty, tz, wave, 2 * (lane & 31) + h);
        const uint32_t v = lane < 32 ? (occupied ? in : 0u) : (seen ? 0xffffu : (~in & 0xffffu));
        pair |= v << (16 * h);
    }
    reinterpret_cast<uint32_t *>(rec)[lane] = pair;
}

// Sub-tile classification of the coarse tiles carve_fill_kernel could not settle, taken
// from the list the pre-pass wrote: a fixed grid of workgroups, one tile (four sub-tiles,
// one per wave) per turn.  64 VGPRs (the cap of 8 waves per SIMD; two loop-invariant pointers
// are spilled in the prologue and reloaded once per sub-tile): the work is a chain of dependent
// reads per sub-tile (coarse masks + matrix -> summed-area entries -> store).
// (Round 1 launched one workgroup per tile of the whole grid and filled the decided tiles
// from here, 16 bytes per thread into the byte plane: two million workgroups at 1024^3,
// 297 us of which the fill stores were the larger part.)
__global__ __launch_bounds__(256, 8) void carve_classify_kernel(const CarveParams p) {
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const int tshift = p.cyShift + p.czShift;
    const int ntiles = __builtin_amdgcn_readfirstlane(*p.undecidedCount) << tshift;
    for (int i = blockIdx.x; i < ntiles; i += gridDim.x) {
        const int ct = p.undecidedList[i >> tshift];
        const int tl = i & ((1 << tshift) - 1);
        const int tx = ct % p.coarseX;
        const int ty = (((ct / p.coarseX) % p.coarseY) << p.cyShift) + (tl & ((1 << p.cyShift) - 1));
        const int tz = ((ct / (p.coarseX * p.coarseY)) << p.czShift) + (tl >> p.cyShift);
        // (tiles and sub-tiles of an edge coarse tile that lie outside the grid keep the
        // "finished" records they were allocated with)
        if (ty >= p.tilesY || tz >= p.tilesZ || tx * kTileX + wave * kSubX >= p.X) continue;
        const SubTile t = subtile_of(p, tx, ty, tz, wave, lane);
        uint16_t *const rec = p.rec + rec_index(p, tx, ty, tz, wave) * kRecU16;
        const BoxW box = make_box(p.s, t.sx0, t.sx1, t.sy0, t.sy1, global_z(p, t.sz0),
                                  global_z(p, t.sz1));
        bool any_carved = false, any_fg = false, any_mixed = false;
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
                unsigned sel =
                    (unsigned)((cf >> lane) & 1ull) | ((unsigned)((cm >> lane) & 1ull) << 1);
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
                    cls = classify_box(Mr, box, p.W, p.H, p.sat + (size_t)myv * p.satStride,
                                       p.satW);
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
        const bool fresh = p.flags & 4u;
        if (any_carved) {  // carved implies seen (src/VoxelCarving.cpp:50-54): no load needed
            subtile_store_done(rec, lane);
        } else if (!any_mixed) {
            // no view needs a closer look and none carves: occupancy stays, and every voxel is
            // seen if some view sees the whole box (src/VoxelCarving.cpp:54)
            if (fresh)
                subtile_store_const(p, rec, lane, tx, ty, tz, wave, true, any_fg);
            else if (any_fg && lane >= 32)
                reinterpret_cast<uint32_t *>(rec)[lane] = 0xffffffffu;  // the seen half
        } else if ((p.flags & 12u) == 12u) {
            // the exact kernel may hand this sub-tile to several waves that merge their
            // results with atomics: the record must hold the sub-tile's initial state (a fresh
            // model exists only as a flag until now)
            subtile_store_const(p, rec, lane, tx, ty, tz, wave, true, false);
        }
        if (!any_carved && any_mixed && lane == 0) {
            // hand the sub-tile to the exact kernel: where it is, whether some view sees
            // all of it, and per chunk of 64 views which ones to evaluate (and how to divide).
            // Eight weight classes of eight lists, the longest items (most views to evaluate)
            // first: the waves start on those, and what the kernel ends on are the short ones;
            // consecutive sub-tiles go to consecutive lists of their class.
            int nmixed = 0;
#pragma unroll
            for (int c = 0; c < kMaxChunks; ++c)
                if (c < p.nchunks) nmixed += __popcll(mixed_c[c]);
            const int wclass = 7 - min(7, nmixed * 8 / (p.v1 - p.v0 + 1));
            // (numbered over the tiles of the grid, not over the list: the host sizes a list
            // for every 8th sub-tile of the grid)
            const int cls =
                wclass * 8 + (((((tz * p.tilesY + ty) * p.tilesX + tx) << 2) + wave) & 7);
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
}

// The same job with the lanes packed densely.  In carve_classify_kernel a wave owns ONE sub-tile
// and its lanes are the views: at most V of 64 lanes exist (36 in the data sets), and of those
// only the views that are still "mixed" for the coarse tile -- typically a quarter -- run the
// rectangle test, while the wave pays for all of it.  Here a workgroup owns a listed coarse tile
// (64 sub-tiles; 32 on striped slabs) -- or, when there are fewer listed tiles than workgroups,
// a QUARTER of one -- and a lane is a (sub-tile, view slot) pair: a wave takes the coarse tile's
// mixed views one (or four) at a time, every fourth group, so the rectangle test runs with all
// lanes busy and a lane's box is formed once per unit.  The lanes OR their answers into LDS;
// then every wave settles or queues a quarter of the unit's sub-tiles exactly as the kernel
// above does (one record store per sub-tile).
// (6 workgroups per CU: 75 VGPRs, no scratch.  4: step +2 % at 512^3 and 768^3 -- fewer units in
// flight for the same chain of dependent reads; 8: 64 VGPRs with 20 B of scratch, no better.)
#ifndef ARVX_DENSE_WGS_PER_CU
#define ARVX_DENSE_WGS_PER_CU 6
#endif
__global__ __launch_bounds__(256, ARVX_DENSE_WGS_PER_CU) void carve_classify_dense_kernel(const CarveParams p) {
    __shared__ unsigned long long s_mixed[kMaxChunks][64], s_fast[kMaxChunks][64];
    __shared__ unsigned s_flag[64];  // bit0: some view carves the sub-tile, bit1: some view sees all of it
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tshift = p.cyShift + p.czShift;
    const int nlisted = __builtin_amdgcn_readfirstlane(*p.undecidedCount);
    // whole coarse tiles while they fill the grid (1024^3: -10 % on the carve against quarters,
    // which in turn are what keeps small grids from waiting for a few long workgroups)
    const int split = nlisted >= (int)gridDim.x ? 0 : 2;  // log2 units per coarse tile
    const int ushift = tshift + 2 - split;                 // log2 sub-tiles of a unit
    const int U = 1 << ushift, VS = 64 >> ushift;          // ... and view slots of a wave
    const int sub = lane & (U - 1), vslot = lane >> ushift;
    const int nunits = nlisted << split;
    for (int ui = blockIdx.x; ui < nunits; ui += gridDim.x) {
        const int ct = __builtin_amdgcn_readfirstlane(p.undecidedList[ui >> split]);
        const int cx = ct % p.coarseX;
        const int cty = ((ct / p.coarseX) % p.coarseY) << p.cyShift;
        const int ctz = (ct / (p.coarseX * p.coarseY)) << p.czShift;
        const int s0 = (ui & ((1 << split) - 1)) * U;  // the unit's first sub-tile (of the coarse tile's)
        if (threadIdx.x < 64) {
#pragma unroll
            for (int c = 0; c < kMaxChunks; ++c) s_mixed[c][threadIdx.x] = s_fast[c][threadIdx.x] = 0;
            s_flag[threadIdx.x] = 0;
        }
        __syncthreads();
        // ---- phase 1: lane = (sub-tile, view slot)
        {
            const int sidx = s0 + sub, tl = sidx >> 2, sw = sidx & 3;
            const int ty = cty + (tl & ((1 << p.cyShift) - 1)), tz = ctz + (tl >> p.cyShift);
            const int sx0 = cx * kTileX + sw * kSubX, sy0 = ty * kTileY, sz0 = tz * kTileZ;
            const bool in_grid = ty < p.tilesY && tz < p.tilesZ && sx0 < p.X;
            BoxW box = make_box(p.s, 0, 0, 0, 0, 0, 0);
            if (in_grid)
                box = make_box(p.s, sx0, min(sx0 + kSubX - 1, p.X - 1), sy0,
                               min(sy0 + kTileY - 1, p.Y - 1), global_z(p, sz0),
                               global_z(p, min(sz0 + kTileZ - 1, p.Z - 1)));
            unsigned flag = 0;
            int k = 0;  // running number of the coarse tile's mixed views
#pragma unroll
            for (int chunk = 0; chunk < kMaxChunks; ++chunk) {
                const int vc = p.v0 + 64 * chunk;
                if (vc >= p.v1) continue;
                unsigned long long cm = uniform64(p.coarseMixed[(size_t)ct * p.nchunks + chunk]);
                const unsigned long long cf = uniform64(p.coarseFg[(size_t)ct * p.nchunks + chunk]);
                if (cf) flag |= 2u;  // inherited: the coarse rectangle contains every sub-tile's
                unsigned long long mixed = 0, fast = 0;
                int myb = -1;  // this lane's view of the group being collected
                while (cm) {
                    const int b = __ffsll((long long)cm) - 1;
                    cm &= cm - 1;
                    const int g = k >> (6 - ushift), slot = k & (VS - 1);
                    ++k;
                    const bool ours = (g & 3) == wave;  // wave-uniform
                    if (ours && slot == vslot) myb = b;
                    if (!ours || !(slot == VS - 1 || cm == 0)) continue;
                    // a group is complete (or the chunk ends): one rectangle test for 64 lanes
                    int cls = kClsOut;
                    if (myb >= 0 && in_grid) {
                        const int view = vc + myb;
                        cls = classify_box(p.M + 12 * view, box, p.W, p.H,
                                           p.sat + (size_t)view * p.satStride, p.satW);
                    }
                    if (myb >= 0) {
                        if (cls & kFastDiv) fast |= 1ull << myb;
                        cls &= 3;
                        if (cls == kClsMixed) mixed |= 1ull << myb;
                        if (cls == kClsCarved) flag |= 1u;
                        if (cls == kClsFg) flag |= 2u;
                    }
                    myb = -1;
                }
                if (mixed) atomicOr(&s_mixed[chunk][sub], mixed);
                if (fast) atomicOr(&s_fast[chunk][sub], fast);
            }
            if (flag) atomicOr(&s_flag[sub], flag);
        }
        __syncthreads();
        // ---- phase 2: the records the unit settles (and the initial state of shared items),
        // 16 bytes per thread: chunk c of a record = the eight occupancy rows of plane c (c < 8)
        // or the eight seen rows of plane c - 8.  (A wave per sub-tile with four bytes per lane
        // did the sub-tile's index arithmetic 64 lanes wide for 256 bytes: at 1024^3 that was
        // about half of this kernel's vector instructions.)
        {
            const bool fresh = p.flags & 4u;
            for (int q = threadIdx.x; q < (U << 4); q += 256) {
                const int sl = q >> 4, c = q & 15;
                const int sidx = s0 + sl, tl = sidx >> 2, sw = sidx & 3;
                const int tx = cx;
                const int ty = cty + (tl & ((1 << p.cyShift) - 1)), tz = ctz + (tl >> p.cyShift);
                const int x0 = tx * kTileX + sw * kSubX;
                // (tiles and sub-tiles of an edge coarse tile that lie outside the grid keep the
                // "finished" records they were allocated with)
                if (ty >= p.tilesY || tz >= p.tilesZ || x0 >= p.X) continue;
                bool any_mixed = false;
#pragma unroll
                for (int k = 0; k < kMaxChunks; ++k)
                    if (k < p.nchunks) any_mixed = any_mixed || s_mixed[k][sl] != 0;
                const unsigned flag = s_flag[sl];
                const bool any_carved = flag & 1u, any_fg = flag & 2u;
                const bool seen_half = c >= 8;
                uint4 v;
                if (any_carved) {  // carved implies seen (src/VoxelCarving.cpp:50-54)
                    v.x = v.y = v.z = v.w = seen_half ? 0xffffffffu : 0u;
                } else if (!any_mixed && !fresh) {
                    // no view needs a closer look and none carves: occupancy stays, and every
                    // voxel is seen if some view sees the whole box (src/VoxelCarving.cpp:54)
                    if (!(any_fg && seen_half)) continue;
                    v.x = v.y = v.z = v.w = 0xffffffffu;
                } else if (!any_mixed || (p.flags & 12u) == 12u) {
                    // a fresh model's record from constants: occupied inside the grid, seen where
                    // a view sees the whole box (outside the grid: always) -- and, for a sub-tile
                    // the exact kernel may hand to several waves that merge their results with
                    // atomics, the initial state (a fresh model exists only as a flag until now)
                    const bool seen = !any_mixed && any_fg;
                    const int z = tz * kTileZ + (c & 7);
                    const int ny = z < p.Z ? min(kTileY, p.Y - ty * kTileY) : 0;  // rows inside
                    const uint32_t xm = 0xffffu >> (kSubX - min(kSubX, p.X - x0));
                    uint32_t w[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const uint32_t in = (2 * k < ny ? xm : 0u) | (2 * k + 1 < ny ? xm << 16 : 0u);
                        w[k] = !seen_half ? in : (seen ? 0xffffffffu : ~in);
                    }
                    v.x = w[0];
                    v.y = w[1];
                    v.z = w[2];
                    v.w = w[3];
                } else {
                    continue;
                }
                uint16_t *const rec = p.rec + rec_index(p, tx, ty, tz, sw) * kRecU16;
                reinterpret_cast<uint4 *>(rec)[c] = v;
            }
        }
        // ... and queue: lane = sub-tile, so that the unit's appends to the work lists (an atomic
        // whose answer the item's place depends on) are in flight together instead of one
        // round trip per sub-tile
        if (wave == 0 && lane < U) {
            const int sl = lane, sidx = s0 + sl, tl = sidx >> 2, sw = sidx & 3;
            const int tx = cx;
            const int ty = cty + (tl & ((1 << p.cyShift) - 1)), tz = ctz + (tl >> p.cyShift);
            const unsigned flag = s_flag[sl];
            unsigned long long mixed_c[kMaxChunks], fast_c[kMaxChunks];
            int nmixed = 0;
#pragma unroll
            for (int c = 0; c < kMaxChunks; ++c) {
                mixed_c[c] = fast_c[c] = 0;
                if (c >= p.nchunks) continue;
                mixed_c[c] = s_mixed[c][sl];
                fast_c[c] = s_fast[c][sl];
                nmixed += __popcll(mixed_c[c]);
            }
            const bool in_grid = ty < p.tilesY && tz < p.tilesZ && tx * kTileX + sw * kSubX < p.X;
            if (in_grid && !(flag & 1u) && nmixed) {  // hand it to the exact kernel (see above)
                const int wclass = 7 - min(7, nmixed * 8 / (p.v1 - p.v0 + 1));
                const int cls =
                    wclass * 8 + (((((tz * p.tilesY + ty) * p.tilesX + tx) << 2) + sw) & 7);
                const int pos = atomicAdd(&p.workCount[cls * kCounterStride], 1);
                const size_t it = (size_t)cls * p.workCap + pos;
                p.itemInfo[it] = (unsigned long long)tx | ((unsigned long long)ty << 16) |
                                 ((unsigned long long)tz << 32) | ((unsigned long long)sw << 48) |
                                 ((unsigned long long)((flag & 2u) ? 1 : 0) << 50);
#pragma unroll
                for (int c = 0; c < kMaxChunks; ++c)
                    if (c < p.nchunks) {
                        p.itemMasks[(it * p.nchunks + c) * 2] = mixed_c[c];
                        p.itemMasks[(it * p.nchunks + c) * 2 + 1] = fast_c[c];
                    }
            }
        }
        __syncthreads();  // (the next unit reuses the LDS arrays)
    }
}

// One view applied exactly to the 16 voxels of every lane.  Returns true when all
// 1024 voxels of the sub-tile are carved and seen.
template <bool LEFT>
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
            const float a0 = row_sum<LEFT>(p0[0], p1[0][j], p20, p3[0]);
            const float a1 = row_sum<LEFT>(p0[1], p1[1][j], p21, p3[1]);
            const float a2 = row_sum<LEFT>(p0[2], p1[2][j], p22, p3[2]);
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
__device__ __forceinline__ void for_each_work_item_of(const CarveParams &p, const int lane,
                                                      const int srank, const int nstatic,
                                                      int incl, Body body);
template <bool kSplit, class Body>
__device__ __forceinline__ void for_each_work_item(const CarveParams &p, const int lane,
                                                   const int srank, const int nstatic,
                                                   Body body) {
    // the list fill counts, list l in lane l
    for_each_work_item_of<kSplit>(p, lane, srank, nstatic,
                                  (lane < kWorkLists) ? p.workCount[lane * kCounterStride] : 0, body);
}
// ... with the counts handed in (lane l: list l)
template <bool kSplit, class Body>
__device__ __forceinline__ void for_each_work_item_of(const CarveParams &p, const int lane,
                                                      const int srank, const int nstatic,
                                                      int incl, Body body) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    // inclusive prefix of the counts
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
        if (p.flags & 64u)  // experiment builds: two parts whatever the list length
            shift = 1;
        else if (16 * items <= p.nwaves)
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

template <bool LEFT>
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
            uint16_t *const rec = p.rec + rec_index(p, tx, ty, tz, wave) * kRecU16;
            uint32_t st[4];
            subtile_load(p, t, rec, lane, st);
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
                    done = exact_view<LEFT>(p, p.v0 + 64 * c + b, (fastdiv >> b) & 1ull, dwy, dwx, dwz,
                                      st, lane);
#ifdef ARVX_TIMELINE
                    wave_timeline.view_done();
#endif
                }
            }
#ifdef ARVX_TIMELINE
            wave_timeline.item_done();
#endif
            subtile_store(rec, lane, st);
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

// bits 0, 4, 8, 12 <-> bit 0 of the four bytes
__device__ __forceinline__ uint32_t spread4(uint32_t t) {
    return (t & 1u) | ((t & 0x10u) << 4) | ((t & 0x100u) << 8) | ((t & 0x1000u) << 12);
}
__device__ __forceinline__ uint32_t gather4(uint32_t b) {
    return (b & 1u) | ((b >> 4) & 0x10u) | ((b >> 8) & 0x100u) | ((b >> 12) & 0x1000u);
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
// need: bit 4 m + j set = block j of group m = 2 byi + bzi may be cut by this view's silhouette
// (block_tests below); the other blocks are not projected.
template <bool LEFT>
__device__ __forceinline__ bool exact_view_blocks(const CarveParams &p, const int view,
                                                  const bool fast, const float (&wy)[2],
                                                  const float (&wx)[4], const float (&wz)[2],
                                                  uint32_t (&st)[4], const unsigned need) {
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
    // `did`: the blocks that were projected (scalar): only those read the table and update
    // their state below -- after the block-level rectangle tests that is a few of the sixteen,
    // and the fixed frame of sixteen loads and updates per view was most of what was left
    uint32_t pix[4][4];  // pixel_tagged
    unsigned did = 0;
    auto project = [&](const int m, const int j, const double s0, const double s1,
                       const double s2) {
        did |= 1u << (4 * m + j);
        const float a0 = (float)s0, a1 = (float)s1, a2 = (float)s2;
        float u, v;
        if (fast) {
            divide2_shared_rcp(a0, a1, a2, u, v);
        } else {
            u = a0 / a2;
            v = a1 / a2;
        }
        pix[m][j] = pixel_tagged(u, v, p.W, wlim, hlim, zero_pix);
    };
    if constexpr (!LEFT) {
    // a_r = p0[y] + ((p1[x] + p2[z]) + p3) (row_sum): the inner sum q depends on x and z
    // only, so it is formed once per (x, z) of the lane -- 4 x 2 values per row -- and a voxel
    // costs ONE fp64 add per row.  p1 is an exact product, so fma(m1, wx, p2) IS
    // round(p1 + p2).  (Tried and dropped: forming q only for the columns j whose blocks are
    // projected -- the branches cost more than the sums they skip, carve +2..3 %.)
    double p0[2][3];
#pragma unroll
    for (int byi = 0; byi < 2; ++byi)
#pragma unroll
        for (int r = 0; r < 3; ++r) p0[byi][r] = (double)mf[r][0] * (double)wy[byi];
#pragma unroll
    for (int bzi = 0; bzi < 2; ++bzi) {
        if (!(need & (0x0f0fu << (4 * bzi)))) continue;  // groups m = bzi and 2 + bzi
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
            if (!((need >> (4 * m)) & 15u) || !__any(w != kDone4)) continue;  // nothing to do here
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!((need >> (4 * m + j)) & 1u)) continue;  // decided by its rectangle
                if (!__any(((w >> (8 * j)) & 0xffu) != 2u)) continue;  // block j is finished
                project(m, j, p0[byi][0] + q[0][j], p0[byi][1] + q[1][j], p0[byi][2] + q[2][j]);
            }
        }
    }
    } else {
    // a_r = ((p0[y] + p1[x]) + p2[z]) + p3: the inner sum depends on y and x
#pragma unroll
    for (int byi = 0; byi < 2; ++byi) {
        if (!((need >> (8 * byi)) & 0xffu)) continue;
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
            if (!((need >> (4 * m)) & 15u) || !__any(w != kDone4)) continue;  // nothing to do here
            const double dwz = (double)wz[bzi];
            const double p20 = (double)mf[0][2] * dwz, p21 = (double)mf[1][2] * dwz,
                         p22 = (double)mf[2][2] * dwz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!((need >> (4 * m + j)) & 1u)) continue;  // decided by its rectangle
                if (!__any(((w >> (8 * j)) & 0xffu) != 2u)) continue;  // block j is finished
                project(m, j, (p01[0][j] + p20) + p3[0], (p01[1][j] + p21) + p3[1],
                        (p01[2][j] + p22) + p3[2]);
            }
        }
    }
    }
    if (!did) return false;
    uint32_t word[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((did >> (4 * m + j)) & 1u) word[m][j] = bgv[(pix[m][j] & 0x7fffffffu) >> 5];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (!((did >> (4 * m)) & 15u)) continue;
        uint32_t w = st[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!((did >> (4 * m + j)) & 1u)) continue;
            const uint32_t isbg = __builtin_amdgcn_ubfe(word[m][j], pix[m][j], 1u);  // bit pix & 31
            const uint32_t seen = (pix[m][j] >> 31) << (8 * j + 1);
            w = (w | seen) & ~(isbg << (8 * j));
        }
        st[m] = w;
    }
    return __all(st[0] == kDone4 && st[1] == kDone4 && st[2] == kDone4 && st[3] == kDone4);
}

// The brute-force form: every sub-tile, every view (ARVX_CARVE_NO_CULL: the ablation and the
// yardstick of the tests; with the culling on, the one-kernel form ARVX_CARVE_FUSED and more than
// 256 views).  One wave per sub-tile, the views one after the other through exact_view_blocks --
// the exact kernel's own block-mapped evaluation: the matrix through the scalar cache, the (y, x)
// part of the rows hoisted, all sixteen blocks projected before the table is read, ONE wait per view
// (round 5; the row-mapped form before it waited four times per view and read the matrix with
// vector loads: 2.16 -> EXPERIMENTS.md round 5).
template <bool LEFT>
__global__ __launch_bounds__(256, 4) void carve_fused_kernel(const CarveParams p) {
#ifdef ARVX_TIMELINE
    TimelineScope timeline_scope(p.timeline);
#endif
    // Rows of tiles along x are dealt to the 8 XCDs cyclically (blocks b, b+8, b+16.. share
    // an XCD), which spreads the expensive surface tiles evenly: contiguous z ranges per XCD
    // left the XCDs that own the empty top and bottom of the grid idle (+40 % on the sphere
    // scene).
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
    uint16_t *const rec = p.rec + rec_index(p, tx, ty, tz, wave) * kRecU16;
    // Pure fill of a tile that the pre-pass decided: carved+seen (code 1), or, for a fresh
    // model, untouched occupancy with (2) / without (3) the seen bit.
    if (coarse_carved || (code >= 2 && (p.flags & 4u))) {
        uint32_t pair = 0;  // entries 2 * (lane & 31), + 1 of the occ (lane < 32) / seen half
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const uint32_t in = row_inmask(p, tx, ty, tz, wave, 2 * (lane & 31) + h);
            const uint32_t v = code == 1 ? (lane < 32 ? 0u : 0xffffu)
                                         : (lane < 32 ? in : (code == 2 ? 0xffffu : (~in & 0xffffu)));
            pair |= v << (16 * h);
        }
        reinterpret_cast<uint32_t *>(rec)[lane] = pair;
        if ((p.flags & 2u) && lane == 0) {
            atomicAdd(&p.stats[0], 1ull);
            if (coarse_carved) atomicAdd(&p.stats[1], 1ull);
        }
        return;
    }
    const SubTile t = subtile_of(p, tx, ty, tz, wave, lane);
    if (t.sx0 >= p.X) return;  // wave-uniform
    // block map: one voxel of each of the sixteen 4 x 4 x 4 blocks per lane (exact_view_blocks)
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    float wx[4], wy[2], wz[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) wx[j] = (float)(t.sx0 + 4 * j + lx) * p.s;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        wy[b] = (float)(t.sy0 + 4 * b + ly) * p.s;
        wz[b] = (float)(-global_z(p, t.sz0 + 4 * b + lz)) * p.s;
    }
    uint32_t st[4] = {kDone4, kDone4, kDone4, kDone4};
    bool loaded = false, all_carved = false, all_done = false;
    const BoxW box = make_box(p.s, t.sx0, t.sx1, t.sy0, t.sy1, global_z(p, t.sz0), global_z(p, t.sz1));

    int chunk = 0;
    for (int vc = p.v0; vc < p.v1 && !all_done && !all_carved; vc += 64, ++chunk) {
        const int myv = vc + lane;
        int cls = kClsOut;
        if (myv < p.v1) {
            if (!cull) {
                // every voxel is projected; the rectangle arithmetic only says whether the shared-
                // reciprocal division is the IEEE one on this box (no table is looked at)
                float Mr[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) Mr[i] = p.M[12 * myv + i];
                cls = kClsMixed | rect_prepare(Mr, box, p.W, p.H, p.satW).fast;
            } else {
                const unsigned long long cm = p.coarseMixed[(size_t)ct * p.nchunks + chunk];
                const unsigned long long cf = p.coarseFg[(size_t)ct * p.nchunks + chunk];
                if ((cf >> lane) & 1ull)
                    cls = kClsFg;  // inherited: the coarse rectangle contains this one
                else if ((cm >> lane) & 1ull)
                    cls = classify_box(p.M + 12 * myv, box, p.W, p.H,
                                       p.sat + (size_t)myv * p.satStride, p.satW);
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
        if (!loaded) {  // record -> blocks (as carve_exact_blocks_kernel)
            loaded = true;
#pragma unroll
            for (int byi = 0; byi < 2; ++byi)
#pragma unroll
                for (int bzi = 0; bzi < 2; ++bzi) {
                    const int r = (4 * bzi + lz) * 8 + 4 * byi + ly;
                    uint32_t o, sn;
                    if (p.flags & 4u) {  // fresh model, never written: all occupied, none seen
                        o = row_inmask(p, tx, ty, tz, wave, r);
                        sn = ~o & 0xffffu;
                    } else {
                        o = rec[r];
                        sn = rec[64 + r];
                    }
                    st[2 * byi + bzi] = spread4((o >> lx) & 0x1111u) | (spread4((sn >> lx) & 0x1111u) << 1);
                }
        }
        if (infg) {
#pragma unroll
            for (int m = 0; m < 4; ++m) st[m] |= kDone4;  // seen, src/VoxelCarving.cpp:54
        }
        while (mixed) {
            const int b = __ffsll((long long)mixed) - 1;
            mixed &= mixed - 1;
            if (p.flags & 2u) {  // how many of the 1 024 evaluations were still open
                int open = 0;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const uint32_t x4 = st[m] ^ kDone4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) open += ((x4 >> (8 * j)) & 0xffu) ? 1 : 0;
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) open += __shfl_xor(open, d);
                if (lane == 0) {
                    atomicAdd(&p.stats[5], 4ull);  // (in 256-voxel slices, as the row-mapped kernels count)
                    atomicAdd(&p.stats[6], (unsigned long long)open);
                }
            }
            all_done = exact_view_blocks<LEFT>(p, __builtin_amdgcn_readfirstlane(vc + b), (fastdiv >> b) & 1ull, wy,
                                               wx, wz, st, 0xffffu);
            if (all_done) break;
        }
    }

    if (all_carved) {
        subtile_store_done(rec, lane);
    } else if (loaded) {  // (else: empty view range, nothing changed) blocks -> record
#pragma unroll
        for (int byi = 0; byi < 2; ++byi)
#pragma unroll
            for (int bzi = 0; bzi < 2; ++bzi) {
                const uint32_t w = st[2 * byi + bzi];
                uint32_t both = (gather4(w & 0x01010101u) << lx) | (gather4((w >> 1) & 0x01010101u) << (16 + lx));
                both |= __shfl_xor(both, 1);
                both |= __shfl_xor(both, 2);
                if (lx != 0) continue;
                const int r = (4 * bzi + lz) * 8 + 4 * byi + ly;
                rec[r] = (uint16_t)both;
                rec[64 + r] = (uint16_t)(both >> 16);
            }
    }
}

// ---- the fp32 filter (round 5; experiment builds only: it loses, EXPERIMENTS.md) -------------------
//
// A voxel's pixel is (round(u), round(v)) of the reference's u = fl32(a0 / a2), v = fl32(a1 / a2),
// a_r = fl32(the fp64 row sum) -- 6 fp64 adds, 3 conversions and a correctly rounded two-wide
// division per voxel and view.  Almost always the same pixel comes out of plain fp32: three FMAs per
// row from the fp32 matrix, one v_rcp, two multiplications.  "Almost": the two differ only where u
// or v lies within the filter's error of a rounding tie k + 1/2 (the image's borders -1/2 and
// W - 1/2 are such ties too).  With
//   E_r  = sum_k |M[r][k]| max|w_k| (over the sub-tile) + |M[r][3]|
//   |ahat_r - a_r| <= e_r = 4.5 * 2^-24 * E_r     (three fp32 roundings of partial sums <= E_r here,
//                                                  one of the reference's own, fp64 dust)
//   uhat = fl32(ahat_0 * rcp(ahat_2)),  |rcp| error 1 ulp, the product's 1/2 ulp, the reference's
//                                        division 1/2 ulp:  <= 1.25 * 2^-22 |u| together
//   |uhat - u| <= ((e_0 + |u| e_2) / |ahat_2|) / (1 - e_2 / |ahat_2|) + 1.25 * 2^-22 |u|
// a lane whose uhat, vhat keep that distance from every tie has the reference's pixel for certain;
// the others -- a few in a thousand -- are flagged, and ONE exact evaluation per view (rarely two)
// settles all flagged (lane, block) pairs of the view together, each lane on its own block.  The
// guard e_2 / |ahat_2| < 1/64 (else: flagged) makes the second factor <= 1.016; 1.02 is used.
// NaN / infinity anywhere fails the comparisons and flags the lane.
// Pays where a view projects many of the sub-tile's blocks: its per-view set-up (the bounds, the
// hoisted partial sums) and the flagged pass are paid per view, not per block.  A launch runs all
// its views through the filter or none (both paths in one kernel: 108 B/lane of scratch).

// the table reads and the state update of a view: pix[m][j] for the blocks in `did`
__device__ __forceinline__ bool apply_view_pixels(const uint32_t *__restrict__ bgv, const uint32_t (&pix)[4][4],
                                                  const unsigned did, uint32_t (&st)[4]) {
    uint32_t word[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if ((did >> (4 * m + j)) & 1u) word[m][j] = bgv[(pix[m][j] & 0x7fffffffu) >> 5];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if (!((did >> (4 * m)) & 15u)) continue;
        uint32_t w = st[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!((did >> (4 * m + j)) & 1u)) continue;
            const uint32_t isbg = __builtin_amdgcn_ubfe(word[m][j], pix[m][j], 1u);  // bit pix & 31
            const uint32_t seen = (pix[m][j] >> 31) << (8 * j + 1);
            w = (w | seen) & ~(isbg << (8 * j));
        }
        st[m] = w;
    }
    return __all(st[0] == kDone4 && st[1] == kDone4 && st[2] == kDone4 && st[3] == kDone4);
}

#ifdef ARVX_EXPERIMENTS
// exact_view_blocks with the filter in front.  wmax: max |w| of the sub-tile's voxels (y, x, z terms).
template <bool LEFT>
__device__ __forceinline__ bool filtered_view_blocks(const CarveParams &p, const int view,
                                                     const bool fast, const float (&wy)[2],
                                                     const float (&wx)[4], const float (&wz)[2],
                                                     const float (&wmax)[3], uint32_t (&st)[4],
                                                     const unsigned need) {
    const uint32_t *__restrict__ bgv = p.bg + (size_t)view * p.bgWords;
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
    const float mf[3][4] = {{row0.x, row0.y, row0.z, row0.w}, {row1.x, row1.y, row1.z, row1.w},
                            {row2.x, row2.y, row2.z, row2.w}};
    const float wlim = (float)p.W - 0.5f, hlim = (float)p.H - 0.5f;
    const int zero_pix = 32 * (p.bgWords - 1);
    // the filter's error scales for this view and sub-tile (the same in every lane)
    const float kE = 4.5f * 5.9604644775390625e-08f * 1.02f;  // 4.5 * 2^-24 * 1.02
    float e[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
        e[r] = (fabsf(mf[r][0]) * wmax[0] + fabsf(mf[r][1]) * wmax[1] + fabsf(mf[r][2]) * wmax[2] +
                fabsf(mf[r][3])) * kE;
    const float kQ = 1.25f * 2.384185791015625e-07f * 1.02f;  // 1.25 * 2^-22 * 1.02
    uint32_t pix[4][4];
    unsigned did = 0;
    unsigned flagged = 0;  // per LANE: bit 4 m + j = this lane's voxel of block (m, j) is near a tie
#pragma unroll
    for (int byi = 0; byi < 2; ++byi) {
        if (!((need >> (8 * byi)) & 0xffu)) continue;
        if (!__any(st[2 * byi] != kDone4 || st[2 * byi + 1] != kDone4)) continue;
        float c[3][4];  // the (y, x) part of the rows: two roundings, the z term below a third
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float ty = fmaf(mf[r][0], wy[byi], mf[r][3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[r][j] = fmaf(mf[r][1], wx[j], ty);
        }
#pragma unroll
        for (int bzi = 0; bzi < 2; ++bzi) {
            const int m = 2 * byi + bzi;
            const uint32_t w = st[m];
            if (!((need >> (4 * m)) & 15u) || !__any(w != kDone4)) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!((need >> (4 * m + j)) & 1u)) continue;
                if (!__any(((w >> (8 * j)) & 0xffu) != 2u)) continue;
                const float a0 = fmaf(mf[0][2], wz[bzi], c[0][j]);
                const float a1 = fmaf(mf[1][2], wz[bzi], c[1][j]);
                const float a2 = fmaf(mf[2][2], wz[bzi], c[2][j]);
                const float rc = __builtin_amdgcn_rcpf(a2);
                const float u = a0 * rc, v = a1 * rc;
                const float arc = fabsf(rc);
                const float g2 = arc * e[2];
                const float q = g2 + kQ;
                const float du = fmaf(fabsf(u), q, arc * e[0]);
                const float dv = fmaf(fabsf(v), q, arc * e[1]);
                const float fu = __builtin_amdgcn_fractf(u) - 0.5f, fv = __builtin_amdgcn_fractf(v) - 0.5f;
                // (negated >=: a NaN anywhere flags the lane)
                const bool near = (int)!(fabsf(fu) >= du) | (int)!(fabsf(fv) >= dv) | (int)!(g2 < 0.015625f);
                did |= 1u << (4 * m + j);
                pix[m][j] = pixel_tagged(u, v, p.W, wlim, hlim, zero_pix);
                flagged |= near ? (1u << (4 * m + j)) : 0u;
            }
        }
    }
    if (!did) return false;
    // the flagged (lane, block) pairs in the reference's own arithmetic: every lane takes its lowest
    // flagged block, all lanes together
    while (__any(flagged != 0u)) {
        const bool act = flagged != 0u;
        const int b = act ? (__ffs((int)flagged) - 1) : 0;
        flagged &= flagged - 1u;
        const int j = b & 3, byi = (b >> 3) & 1, bzi = (b >> 2) & 1;
        const float lwx = j == 0 ? wx[0] : (j == 1 ? wx[1] : (j == 2 ? wx[2] : wx[3]));
        const float lwy = byi ? wy[1] : wy[0], lwz = bzi ? wz[1] : wz[0];
        float a[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double p0 = (double)mf[r][0] * (double)lwy, p1 = (double)mf[r][1] * (double)lwx,
                         p2 = (double)mf[r][2] * (double)lwz;
            a[r] = row_sum<LEFT>(p0, p1, p2, (double)mf[r][3]);
        }
        float u, v;
        if (fast) {
            divide2_shared_rcp(a[0], a[1], a[2], u, v);
        } else {
            u = a[0] / a[2];
            v = a[1] / a[2];
        }
        const uint32_t np = pixel_tagged(u, v, p.W, wlim, hlim, zero_pix);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                if ((did >> (4 * m + jj)) & 1u) pix[m][jj] = (act && b == 4 * m + jj) ? np : pix[m][jj];
    }
    return apply_view_pixels(bgv, pix, did, st);
}

#endif  // ARVX_EXPERIMENTS

// Rectangle tests per 4 x 4 x 4 BLOCK, for the views of one chunk that are "mixed" for the
// sub-tile: the silhouette's edge crosses the sub-tile's pixel rectangle, but most of its
// sixteen blocks lie on one side of it.  Four views at a time, lane = (view q = lane >> 4,
// block lane & 15), the same conservative classify_box as for sub-tiles and coarse tiles:
//   block all background in some view   -> carved and seen, finished for every view
//   block all foreground in some view   -> seen
//   only what is left ("mixed")         -> projected voxel by voxel in that view
// Returns, lane s = the s-th view of `views` (ascending bit order), the 16-bit mask of the
// blocks to project in it (bit 4 m + j, m = 2 byi + bzi); carved / seen get the blocks some
// view settled.  `part`/`pshift`: only every 2^pshift-th view belongs to this wave.
// (Tried and dropped: eight views per pass as two independent groups, so that two groups share
// the pass's two dependent round trips -- more scratch, carve +2 % at 512^3 and 1024^3.)
__device__ __forceinline__ unsigned block_tests(const CarveParams &p, const SubTile &t, int vbase,
                                                unsigned long long views, int part, int pshift,
                                                int lane, unsigned &carved, unsigned &seen) {
    const int blk = lane & 15, q = lane >> 4;
    const int j = blk & 3, m = blk >> 2, byi = m >> 1, bzi = m & 1;
    const int x0 = t.sx0 + 4 * j, y0 = t.sy0 + 4 * byi, z0 = t.sz0 + 4 * bzi;
    const bool inside = x0 < p.X && y0 < p.Y && z0 < p.Z;
    const BoxW box = make_box(p.s, x0, min(x0 + 3, p.X - 1), y0, min(y0 + 3, p.Y - 1),
                              global_z(p, z0), global_z(p, min(z0 + 3, p.Z - 1)));
    unsigned needLanes = 0;  // lane s: blocks to project in the s-th view
    int slot = 0;
    for (int nth = 0; views;) {
        int vb[4];
        int n = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // the next four views of this wave (scalar)
            vb[k] = 0;
            while (views && n == k) {
                const int b = __ffsll((long long)views) - 1;
                views &= views - 1;
                if ((nth++ & ((1 << pshift) - 1)) != part) continue;
                vb[k] = b;
                n = k + 1;
            }
        }
        if (!n) break;
        const int myb = q == 0 ? vb[0] : (q == 1 ? vb[1] : (q == 2 ? vb[2] : vb[3]));
        int cls = kClsOut;
        if (q < n && inside) {
            const int view = vbase + myb;
            cls = classify_box(p.M + 12 * view, box, p.W, p.H, p.sat + (size_t)view * p.satStride,
                               p.satW) & 3;
        }
        const unsigned long long c = __ballot(cls == kClsCarved), f = __ballot(cls == kClsFg),
                                 x = __ballot(cls == kClsMixed);
        carved |= (unsigned)((c | (c >> 16) | (c >> 32) | (c >> 48)) & 0xffffu);
        seen |= (unsigned)((f | (f >> 16) | (f >> 32) | (f >> 48)) & 0xffffu);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) {
                const unsigned need16 = (unsigned)((x >> (16 * k)) & 0xffffu);
                if (lane == slot) needLanes = need16;
                ++slot;
            }
    }
    return needLanes;
}

// The same tests with the views' matrices in LDS (sM: 12 floats per view, view 0 = p.v0) and NB
// passes per round: the arithmetic of the round's passes first, then their table reads together,
// then the answers -- one round trip to memory per NB * 4 views instead of two per 4 (the matrix,
// then the table).  vrel: the chunk's first view, relative to p.v0.  Every view of `views` belongs
// to this wave.
template <int NB>
__device__ __forceinline__ unsigned block_tests_lds(const CarveParams &p, const float *sM,
                                                    const SubTile &t, int vrel,
                                                    unsigned long long views, int lane,
                                                    unsigned &carved, unsigned &seen) {
    const int blk = lane & 15, q = lane >> 4;
    const int j = blk & 3, m = blk >> 2, byi = m >> 1, bzi = m & 1;
    const int x0 = t.sx0 + 4 * j, y0 = t.sy0 + 4 * byi, z0 = t.sz0 + 4 * bzi;
    const bool inside = x0 < p.X && y0 < p.Y && z0 < p.Z;
    const BoxW box = make_box(p.s, x0, min(x0 + 3, p.X - 1), y0, min(y0 + 3, p.Y - 1),
                              global_z(p, z0), global_z(p, min(z0 + 3, p.Z - 1)));
    unsigned needLanes = 0;  // lane s: blocks to project in the s-th view
    int slot = 0;
    while (views) {
        RectQ rq[NB];
        int nv[NB], myv[NB];
        int s00[NB], s01[NB], s10[NB], s11[NB];
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            int vb[4];
            nv[g] = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {  // the next four views (scalar)
                vb[k] = 0;
                if (views) {
                    vb[k] = __ffsll((long long)views) - 1;
                    views &= views - 1;
                    nv[g] = k + 1;
                }
            }
            myv[g] = vrel + (q == 0 ? vb[0] : (q == 1 ? vb[1] : (q == 2 ? vb[2] : vb[3])));
            rq[g].code = kClsOut;
            if (q < nv[g] && inside) {
                float Mr[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) Mr[i] = sM[12 * myv[g] + i];
                rq[g] = rect_prepare(Mr, box, p.W, p.H, p.satW);
            }
        }
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            s00[g] = s01[g] = s10[g] = s11[g] = 0;
            if (rq[g].code < 0) {
                const sat_t *e = p.sat + (size_t)(p.v0 + myv[g]) * p.satStride + rq[g].base;
                s00[g] = e[0];
                s01[g] = e[rq[g].dx];
                s10[g] = e[rq[g].dy * p.satW];
                s11[g] = e[rq[g].dy * p.satW + rq[g].dx];
            }
        }
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            if (!nv[g]) continue;  // (scalar)
            const int cls =
                (rq[g].code >= 0 ? rq[g].code : rect_finish(rq[g], s00[g], s01[g], s10[g], s11[g])) & 3;
            const unsigned long long c = __ballot(cls == kClsCarved), f = __ballot(cls == kClsFg),
                                     x = __ballot(cls == kClsMixed);
            carved |= (unsigned)((c | (c >> 16) | (c >> 32) | (c >> 48)) & 0xffffu);
            seen |= (unsigned)((f | (f >> 16) | (f >> 32) | (f >> 48)) & 0xffffu);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nv[g]) {
                    const unsigned need16 = (unsigned)((x >> (16 * k)) & 0xffffu);
                    if (lane == slot) needLanes = need16;
                    ++slot;
                }
        }
    }
    return needLanes;
}

#ifndef ARVX_EXACT_WAVES_PER_SIMD
#define ARVX_EXACT_WAVES_PER_SIMD 4  // (A/B builds: 5 -> 102 registers, see EXPERIMENTS.md)
#endif
#ifndef ARVX_EXACT_SPLIT_WAVES_PER_SIMD  // the small-grid instantiation (items shared between waves)
#define ARVX_EXACT_SPLIT_WAVES_PER_SIMD 3  // 149 registers, no scratch (4: 128 registers + 60 B/lane of
                                           // scratch, C1 / C2 carves 10 / 7 % slower: EXPERIMENTS.md round 5)
#endif
// SPLIT: items may be handed to several waves (flags bit3: small grids and slabs).  The large
// grids never do: their instantiation carries none of that code (no atomic merge of the parts,
// fewer registers alive across an item).
// FRESH: the model is fresh (flags bit2) -- known when the kernel is compiled: no record is read.
// FILTER (experiment builds): every view goes through the fp32 filter (filtered_view_blocks).
template <bool LEFT, bool SPLIT = true, bool FRESH = false, bool FILTER = false>
__global__ __launch_bounds__(256, (SPLIT && !FRESH ? ARVX_EXACT_SPLIT_WAVES_PER_SIMD : ARVX_EXACT_WAVES_PER_SIMD))
void carve_exact_blocks_kernel(const CarveParams p) {
#ifdef ARVX_TIMELINE
    WaveTimeline wave_timeline(p.timeline);
#endif
    const int lane = threadIdx.x & 63;
    // block map: one voxel per block
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;
    // (Tried and dropped: leaving the pure fill of the decided tiles to a quarter of these
    // workgroups so that it overlaps the exact work -- the fill saturates HBM and the
    // exact waves, which live on memory latency, slow down by more than the fill costs.)
    for_each_work_item<SPLIT>(p, lane, blockIdx.x * 4 + (threadIdx.x >> 6), p.nwaves,
                          [&](const size_t it, const int part, const int list, const int pshift_) {
            const int pshift = SPLIT ? pshift_ : 0;
            // the kernel ends on its longest items (an item's views run one after the other):
            // the items of the heavy weight classes get the SIMD's issue slots first
            // (512^3: -1.2 %, 1024^3: no change; the waves of a SIMD mostly hold items of
            // similar weight, and a view is bound by the SIMD's issue rate either way)
#ifndef ARVX_NO_SETPRIO  // (A/B builds)
            switch (list >> 4) {
                case 0: __builtin_amdgcn_s_setprio(3); break;
                case 1: __builtin_amdgcn_s_setprio(2); break;
                case 2: __builtin_amdgcn_s_setprio(1); break;
                default: __builtin_amdgcn_s_setprio(0); break;
            }
#endif
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
            uint16_t *const rec = p.rec + rec_index(p, tx, ty, tz, wave) * kRecU16;
            // record -> blocks.  st[byi * 2 + bzi] byte j = voxel (4 j + lx, 4 byi + ly,
            // 4 bzi + lz) of the sub-tile, bit0 occupied, bit1 seen: the lane reads the two
            // 16-bit entries of its four rows (y, z) and keeps bits lx, 4 + lx, 8 + lx, 12 + lx.
            const bool fg_seen = (info >> 50) & 1ull;  // seen by an all-foreground view
            uint32_t st[4];
#pragma unroll
            for (int byi = 0; byi < 2; ++byi)
#pragma unroll
                for (int bzi = 0; bzi < 2; ++bzi) {
                    const int r = (4 * bzi + lz) * 8 + 4 * byi + ly;
                    uint32_t o, sn;
                    if (FRESH || (p.flags & 4u)) {  // fresh model, never written: all occupied, none seen
                        o = row_inmask(p, tx, ty, tz, wave, r);
                        sn = ~o & 0xffffu;
                    } else if (pshift) {
                        // other parts of this item update the record with device-scope atomics
                        // while this one reads it, possibly from another XCD: a plain load
                        // would leave a copy of the line in this XCD's L2 for the atomics that
                        // follow to hit (each XCD has its own L2) -- read it the same way
                        const uint32_t *w32 = reinterpret_cast<const uint32_t *>(rec);
                        const int sh = 16 * (r & 1);
                        o = (__hip_atomic_load(w32 + (r >> 1), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT) >> sh) & 0xffffu;
                        sn = (__hip_atomic_load(w32 + 32 + (r >> 1), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT) >> sh) & 0xffffu;
                    } else {
                        o = rec[r];
                        sn = rec[64 + r];
                    }
                    if (fg_seen) sn = 0xffffu;
                    st[2 * byi + bzi] = spread4((o >> lx) & 0x1111u) | (spread4((sn >> lx) & 0x1111u) << 1);
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
            float wmax[3] = {0.f, 0.f, 0.f};  // max |w| over the sub-tile (the filter's error scale)
            if (FILTER) {
                wmax[0] = (float)(t.sy0 + 7) * p.s;
                wmax[1] = (float)(t.sx0 + 15) * p.s;
                wmax[2] = fmaxf(fabsf((float)(-global_z(p, t.sz0)) * p.s),
                                fabsf((float)(-global_z(p, t.sz0 + 7)) * p.s));
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
                // block-level rectangle tests first: what they settle is applied at once
                unsigned bcarved = 0, bseen = 0;
#ifdef ARVX_EXPERIMENTS  // (flags bit5: the A/B switch of the block tests exists in experiment builds only)
                const bool no_bt = p.flags & 32u;
#else
                constexpr bool no_bt = false;
#endif
                const unsigned needLanes = no_bt
                                               ? 0xffffu
                                               : block_tests(p, t, p.v0 + 64 * c, mixed, part, pshift,
                                                             lane, bcarved, bseen);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    uint32_t w = st[m];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if ((bseen >> (4 * m + j)) & 1u) w |= 2u << (8 * j);
                        if ((bcarved >> (4 * m + j)) & 1u)
                            w = (w & ~(0xffu << (8 * j))) | (2u << (8 * j));
                    }
                    st[m] = w;
                }
                done = __all(st[0] == kDone4 && st[1] == kDone4 && st[2] == kDone4 &&
                             st[3] == kDone4);
                int slot = 0;
                for (int nth = 0; mixed && !done; ++nth) {
                    const int b = __ffsll((long long)mixed) - 1;
                    mixed &= mixed - 1;
                    if ((nth & ((1 << pshift) - 1)) != part) continue;  // another part's view
                    const unsigned need = no_bt
                                              ? 0xffffu
                                              : (unsigned)__builtin_amdgcn_readlane(needLanes, slot);
                    ++slot;
                    if (!need) continue;  // every block settled by its rectangle
#ifdef ARVX_EXPERIMENTS
                    if (FILTER)
                        done = filtered_view_blocks<LEFT>(p, __builtin_amdgcn_readfirstlane(p.v0 + 64 * c + b),
                                                          (fastdiv >> b) & 1ull, wy, wx, wz, wmax, st, need);
                    else
#endif
                        done = exact_view_blocks<LEFT>(p, __builtin_amdgcn_readfirstlane(p.v0 + 64 * c + b),
                                                       (fastdiv >> b) & 1ull, wy, wx, wz, st, need);
#ifdef ARVX_TIMELINE
                    wave_timeline.view_done();
#endif
                }
            }
#ifdef ARVX_TIMELINE
            wave_timeline.item_done();
            wave_timeline.tick(2);
#endif
            // blocks -> record: a row's 16 bits sit in the four neighbouring lanes lx = 0..3,
            // four bits each; lane lx = 0 writes the row's two entries
#pragma unroll
            for (int byi = 0; byi < 2; ++byi)
#pragma unroll
                for (int bzi = 0; bzi < 2; ++bzi) {
                    const uint32_t w = st[2 * byi + bzi];
                    uint32_t both = (gather4(w & 0x01010101u) << lx) |
                                    (gather4((w >> 1) & 0x01010101u) << (16 + lx));
                    both |= __shfl_xor(both, 1);
                    both |= __shfl_xor(both, 2);
                    if (lx != 0) continue;
                    const int r = (4 * bzi + lz) * 8 + 4 * byi + ly;
                    if (pshift) {
                        // the parts of an item carve (clear occ) and see (set seen)
                        // independently: both are monotone, so the order in which they reach
                        // the record is irrelevant
                        uint32_t *w32 = reinterpret_cast<uint32_t *>(rec);
                        const int sh = 16 * (r & 1);
                        (void)__hip_atomic_fetch_and(w32 + (r >> 1),
                                                     ((both & 0xffffu) << sh) | ~(0xffffu << sh),
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        (void)__hip_atomic_fetch_or(w32 + 32 + (r >> 1), (both >> 16) << sh,
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        rec[r] = (uint16_t)both;
                        rec[64 + r] = (uint16_t)(both >> 16);
                    }
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
