// arvx_device.h -- shared device-side definitions for the gfx950 kernels.
//
// Arithmetic contract (must match what the reference computes per voxel,
// reference src/VoxelCarving.cpp:18-21,41-54 and src/Model.h:134-140):
//   w    = (float(y)*s, float(x)*s, float(-z)*s, 1)              fp32
//   p_k  = double(M[r][k]) * double(w[k])                        exact in fp64
//   a_r  = float(((p0 + p1) + p2) + p3)                          cv::gemm generic path: the
//          A * Bt branch of GEMMSingleMul<float, double> (a matrix-vector product is
//          turned into it), four accumulators, one product each, summed
//          `(s0 + s1 + s2 + s3) * alpha` -- ARVX_ASSOC_LEFT, the default; or
//          float(p0 + ((p1 + p2) + p3)) -- ARVX_ASSOC_RIGHT, what SURVEY 8c 3 recalls
//          (`s0 += s1 + s2 + s3`).  Both are recollections of OpenCV's source (none in this
//          image); the grouping is a RUNTIME property of a context (arvx_ctx_set_projection_
//          assoc), every projecting kernel exists for both, and include/arvx/opencv_dropin.hpp
//          settles it with one cv::gemm call where OpenCV exists.  tests/test_assoc_gpu.py
//          tells the two apart on a known-answer voxel.
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
constexpr int kWorkLists = 64;      // work lists carve_classify_kernel appends to: 8 weight
                                    // classes (most views to evaluate first) of 8 lists
constexpr int kPoolCounters = 8;    // ticket counters of the shared part of the work
constexpr int kCounterStride = 64;  // ints between counters: one 256-byte block each
constexpr int kMaxChunks = 4;       // split launch handles up to 256 views (else: fused kernel)

enum : int { kClsOut = 0, kClsFg = 1, kClsCarved = 2, kClsMixed = 3, kFastDiv = 4 };

// State of the grid on the device: one 256-byte RECORD per sub-tile (16 x 8 x 8 voxels, the
// unit one wave works on): uint16 occ[64] then uint16 seen[64], entry r = (z & 7) * 8 + (y & 7),
// bit x & 15.  Records are ordered coarse tile > tile > sub-tile, so that everything the carve
// writes as a unit is contiguous: a sub-tile is 256 B, a tile (64 x 8 x 8) 1 KB, a coarse
// tile (64 x 32 x 32; striped slabs 64 x 64 x 8) 16 KB / 8 KB.  Voxels of a record that lie
// outside the grid are kept at (occ 0, seen 1) -- "finished" for every test in the kernels.
// 2 bits per voxel: N / 4 bytes.  The one-byte-per-voxel plane of the C-ABI
// (arvx_state_upload / _download / _device_ptr) is converted from / to this form on demand.
constexpr int kRecU16 = 128;  // uint16 per record

struct CarveParams {
    uint16_t *rec;          // slab state records (see above)
    const uint8_t *ccode;   // readers: per coarse tile 0 = its records hold the state, 1..3 = the
                            // tile is a constant and its records are NOT written (lazy state,
                            // below); null: every record holds the state
    const float *M;         // V x 12
    const uint32_t *bg;     // V x bgWords, bit = 1 where the mask pixel is background
    const uint16_t *sat;    // V x satStride, summed-area table of foreground pixels (mod 2^16)
    unsigned long long *stats;
    unsigned long long *timeline;  // diagnostic builds only (ARVX_TIMELINE), else null
    int X, Y, Z;            // slab extent in voxels (Z = planes held)
    int zoff;               // global z of slab plane 0 (contiguous slabs)
    int zstride, zphase;    // striped slabs: local 8-plane group g is global group
                            // g*zstride + zphase (contiguous: 1, 0)
    float s;                // voxel edge
    int W, H;
    int bgWords, satStride;
    int satW;               // entries per row of a view's table (>= W + 1, padded to whole lines)
    int v0, v1;             // view range [v0, v1)
    unsigned flags;         // bit0 no cull, bit1 stats, bit2 state is fresh (skip the load),
                            // bit3 the exact kernel may split items between waves, bit4 fill
                            // as a fresh model, bit5 no block tests, bit6 always two parts
                            // (5, 6: experiment builds), bit7 the decided coarse tiles are
                            // not written: their code (coarseCarved -> the context's ccode) is
                            // their state
    int tilesX, tilesY, tilesZ;
    // coarse pre-pass results
    int coarseX, coarseY, coarseZ, nchunks;
    int cyShift, czShift;   // coarse tile = 64 x (8 << cyShift) x (8 << czShift)
    unsigned long long *coarseMixed;  // [ncoarse][nchunks] views to re-classify per sub-tile
    unsigned long long *coarseFg;     // [ncoarse][nchunks] views that see only foreground
    uint8_t *coarseCarved;            // [ncoarse] 0 undecided, 1 carved, 2 all seen, 3 none seen
    int coarsePerWg;                  // coarse tiles per workgroup of carve_coarse_kernel
    // what earlier carves of THIS model settled for whole coarse tiles, or null: bit0 = every
    // voxel carved and seen, bit1 = every voxel seen.  Both are monotone under carving (a later
    // view cannot un-carve or un-see), so a carve of a model that is not fresh skips the tiles
    // with bit0 and does not revisit an all-foreground tile with bit1; the host drops the
    // summary when something else writes the state (uploads, the closure).
    uint8_t *cstate;
    int *undecidedList;               // [ncoarse] coarse tiles carve_classify_kernel walks
    int *undecidedCount;              // its length (this launch) ...
    int *undecidedCountNext;          // ... and the counter the next launch will use
    // work lists of the split launch (carve_classify_kernel -> carve_exact_kernel)
    int *workCount;                 // [kWorkLists * kCounterStride] items per list
    int *poolNext;                  // [kPoolCounters * kCounterStride] tickets of the shared pool
    int nwaves;                     // waves of the persistent exact kernel
    int workCap;                    // capacity of one list
    unsigned long long *itemInfo;   // [kWorkLists * workCap] tx | ty<<16 | tz<<32 | wave<<48 | fg<<50
    unsigned long long *itemMasks;  // [..][nchunks][2] mixed views, shared-rcp division ok
    // the streaming carve (carve_stream_kernels.h): one persistent launch, work handed from
    // workgroup to workgroup as tagged granules
    int *sctl;                  // control block: counter k at sctl[k * kCounterStride]
    unsigned long long *listG;  // [8 parts][listCap][listStride] undecided coarse tiles, as they are found
    unsigned long long *itemG;  // [kWorkLists][workCap][itemStride] sub-tiles for exact work
    unsigned epoch;             // tag of this launch's granules (never 0)
    int listCap;                // entries per list part
    int listStride, itemStride; // granules per list entry / per item (items: workCap per list)
    int nA, cwA;                // coarse units: nA of cwA coarse tiles each
    int splitLog2;              // log2 sub-tile units per listed coarse tile (0: whole, 2: quarters)
    unsigned *fault;            // Ctx::d_fault
};

// global z of local plane lz
__device__ __forceinline__ int global_z(const CarveParams &p, int lz) {
    return p.zoff + (((lz >> 3) * p.zstride + p.zphase) << 3) + (lz & 7);
}

// record of sub-tile `wave` (x / 16 inside the tile) of tile (tx, ty, tz)
__host__ __device__ __forceinline__ size_t rec_index(const CarveParams &p, int tx, int ty, int tz,
                                                     int wave) {
    const int ct = tx + p.coarseX * ((ty >> p.cyShift) + p.coarseY * (tz >> p.czShift));
    const int tl = (ty & ((1 << p.cyShift) - 1)) | ((tz & ((1 << p.czShift) - 1)) << p.cyShift);
    return ((((size_t)ct << (p.cyShift + p.czShift)) + tl) << 2) + wave;
}
__host__ __device__ __forceinline__ size_t rec_count(const CarveParams &p) {
    return ((size_t)p.coarseX * p.coarseY * p.coarseZ) << (p.cyShift + p.czShift + 2);
}
// Lazy state.  A carve of a fresh model settles most coarse tiles (64 x 32 x 32 voxels, 16 KB of
// records) as a whole: everything carved and seen (code 1), untouched and seen (2), untouched and
// not even seen (3).  Writing those constants is N / 4 bytes of HBM traffic per carve (the
// largest single share of the carve at 1024^3) that no later stage needs: the per-coarse-tile
// code IS the state of such a tile.  The readers that follow a carve directly -- downloads,
// occupancy packing, the bit planes of the colour pass and the closure -- take the code array
// and synthesise the constants; every other stage first materialises them (arvx_capi.hip,
// need_rec).
__device__ __forceinline__ int coarse_of(const CarveParams &p, int tx, int ty, int tz) {
    return tx + p.coarseX * ((ty >> p.cyShift) + p.coarseY * (tz >> p.czShift));
}
// (bit 7 of a code: the tile's records have been written out since -- by the closure, which
// fills voxels into a few coded tiles, rec_or_bitgrid_lazy_kernel -- and hold the state again)
constexpr int kCodeWritten = 0x80;
__device__ __forceinline__ int lazy_code(const CarveParams &p, int tx, int ty, int tz) {
    if (!p.ccode) return 0;
    const int c = (int)p.ccode[coarse_of(p, tx, ty, tz)];
    return (c & kCodeWritten) ? 0 : c;
}
// in-grid voxels of entry r of sub-tile `wave` of tile (tx, ty, tz)
__device__ __forceinline__ uint32_t row_inmask(const CarveParams &p, int tx, int ty, int tz,
                                               int wave, int r) {
    const int x0 = tx * kTileX + wave * kSubX;
    const int y = ty * kTileY + (r & 7), z = tz * kTileZ + (r >> 3);
    if (y >= p.Y || z >= p.Z || x0 >= p.X) return 0u;
    const int nx = min(kSubX, p.X - x0);
    return 0xffffu >> (kSubX - nx);
}
// entry r of a sub-tile of a coarse tile with code 1..3 (voxels outside the grid: occ 0, seen 1)
__device__ __forceinline__ uint32_t lazy_occ(const CarveParams &p, int code, int tx, int ty, int tz,
                                             int wave, int r) {
    return code == 1 ? 0u : row_inmask(p, tx, ty, tz, wave, r);
}
__device__ __forceinline__ uint32_t lazy_seen(const CarveParams &p, int code, int tx, int ty,
                                              int tz, int wave, int r) {
    return code == 3 ? (~row_inmask(p, tx, ty, tz, wave, r) & 0xffffu) : 0xffffu;
}

// Granules: what crosses workgroups INSIDE a launch (/opt/skills/guides/MI355X_MICROARCH.md, inter-
// workgroup visibility): one naturally aligned 8-byte word {32-bit payload, 32-bit tag} written by
// ONE sc1 store and read by sc1 loads (relaxed, agent scope: past the CU's L1).  A granule is valid
// when its tag is the current launch's: no flag, no fence, no ordering between granules, and
// nothing to reset between launches.
__device__ __forceinline__ void granule_store(unsigned long long *p, uint32_t payload, uint32_t tag) {
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | payload, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long granule_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 4 bits -> bit 0 of 4 bytes, and back
__device__ __forceinline__ uint32_t nibble_to_bytes(uint32_t nib) {
    return (nib * 0x00204081u) & 0x01010101u;
}
__device__ __forceinline__ uint32_t bytes_to_nibble(uint32_t b) {  // b & 0x01010101 only
    return (b * 0x01020408u) >> 24;
}

// (int)std::round(u) for u > -0.5 in ONE instruction: v_cvt_rpi_i32_f32 is floor(u + 1/2)
// with the sum NOT rounded to fp32 first (a rounded sum would turn 0.49999997 into 1).
// For u > -0.5 that is round-half-away-from-zero: ties k + 1/2 go up, and (-0.5, 0) gives 0
// like roundf's -0.  Verified on this hardware against floor(u) + (fract(u) >= 0.5) on every
// float in (-0.5, 2^24] (arvx_selftest_round, tests/test_carve_gpu.py); below -0.5 it differs
// from std::round (ties go up, not away), and there the callers have already said "outside".
__device__ __forceinline__ int round_pixel(float u) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(u));
    return r;
}

// Pixel of a projected voxel from the two quotients u = a0/a2, v = a1/a2.
// Reference: px = (int)std::round(u), inside iff 0 <= px < W (same for v,H),
// src/VoxelCarving.cpp:44-45.  Restated without computing round() first:
//   round(u) >= 0    <=>  u > -0.5        (round(-0.5) = -1, round(-0.4999) = -0)
//   round(u) <= W-1  <=>  u < W - 0.5     (round(W-0.5) = W)
// both bounds are exact floats for W <= 16384; NaN/Inf fail them, which is the
// "outside" x86's cvttss2si gives the reference there.
__device__ __forceinline__ bool pixel_from_quotients(float u, float v, int W, float wlim,
                                                     float hlim, int &pix, int outside_pix = 0) {
    // (bitwise &: with && the compiler builds a chain of exec-masked branches per voxel)
    const bool in = (u > -0.5f) & (u < wlim) & (v > -0.5f) & (v < hlim);
    const int px = round_pixel(u), py = round_pixel(v);
    const int at = __mul24(py, W) + px;  // (24-bit multiply: full rate; in range where it counts)
    pix = in ? at : outside_pix;  // keeps the unconditional table read in bounds
    return in;
}

// The same as ONE value: bit 31 = inside the image, bits 0..30 = the pixel's bit index in the
// view's background plane (outside: `outside_pix`, an always-zero bit).  Sixteen of these stay
// live across the table read of a view; as sixteen lane masks + sixteen ints they spilled.
__device__ __forceinline__ uint32_t pixel_tagged(float u, float v, int W, float wlim, float hlim,
                                                 int outside_pix) {
    const bool in = (u > -0.5f) & (u < wlim) & (v > -0.5f) & (v < hlim);
    const int at = __mul24(round_pixel(v), W) + round_pixel(u);
    return in ? ((uint32_t)at | 0x80000000u) : (uint32_t)outside_pix;
}

// a0/a2 and a1/a2, correctly rounded, with ONE reciprocal.
//
// Why this equals the IEEE quotient.  The compiler expands an fp32 `/` on gfx9 into
//     bs = v_div_scale(b, b, a)        as = v_div_scale(a, b, a)
//     r0 = v_rcp(bs)
//     e  = fma(-bs, r0, 1)             r  = fma(e, r0, r0)
//     q0 = as * r
//     t0 = fma(-bs, q0, as)            q1 = fma(t0, r, q0)
//     t1 = fma(-bs, q1, as)
//     q  = v_div_fmas(t1, r, q1)       = fma(t1, r, q1), times 2^+-32 when something was scaled
//     result = v_div_fixup(q, b, a)    replaces q only for zero / infinite / NaN operands and
//                                      for quotients that overflow or are subnormal
// and that sequence IS the correctly rounded quotient (it is how the hardware divides).  The
// function below is the same instruction sequence with bs = b, as = a: whenever v_div_scale
// returns its operands unchanged and v_div_fixup passes q through, both compute the same bits,
// and two quotients with one denominator can share r.  v_div_scale rescales exactly when
//   (i)   b is subnormal or |b| >= 2^126 (its reciprocal would be subnormal),
//   (ii)  the biased exponent of a is <= 23, i.e. 0 < |a| < 2^-103 (the residuals t would lose
//         bits to the subnormal range),
//   (iii) exponent(a) - exponent(b) >= 96 (the quotient may overflow), or
//   (iv)  the quotient is subnormal, exponent(a) - exponent(b) <= -126.
// Callers use the function only where the rectangle test has shown 2^-60 <= |a2| <= 2^60 and
// |a0|, |a1| <= 2^60 for every voxel of the box (kFastDiv, classify_box): (i) cannot happen;
// under (ii) and (iv) both the IEEE quotient and this one are smaller than 2^-43 in magnitude,
// which is pixel 0 and "inside" either way (pixel_from_quotients: u > -0.5, round = 0); under
// (iii) both exceed 2^95, outside any image either way.  a = 0 gives a zero of some sign from
// both (the sign does not matter to the pixel).  tests/test_carve_gpu.py::
// test_shared_reciprocal_division_is_ieee asserts bit equality with `/` on every operand triple
// outside (ii)-(iv) -- random, whole exponent range, engineered rounding ties -- and the harmless
// outcome inside them.
__device__ __forceinline__ void divide2_shared_rcp(float a0, float a1, float b, float &u,
                                                   float &v) {
    // both quotients go through the same five steps: two-wide (v_pk_mul_f32 / v_pk_fma_f32 do
    // two fp32 operations per lane and issue slot); element-wise identical to the scalar form
    typedef float f2 __attribute__((ext_vector_type(2)));
    float r = __builtin_amdgcn_rcpf(b);
    const float nb = -b;
    const float e = fmaf(nb, r, 1.0f);
    r = fmaf(e, r, r);
#ifndef ARVX_DIV_SCALAR
    const f2 a = {a0, a1}, rr = {r, r}, nbb = {nb, nb};
    f2 q = a * rr;
    f2 t = __builtin_elementwise_fma(nbb, q, a);
    q = __builtin_elementwise_fma(t, rr, q);
    t = __builtin_elementwise_fma(nbb, q, a);
    q = __builtin_elementwise_fma(t, rr, q);
    u = q.x;
    v = q.y;
#else  // A/B: one quotient after the other
    float q = a0 * r;
    float t = fmaf(nb, q, a0);
    q = fmaf(t, r, q);
    t = fmaf(nb, q, a0);
    u = fmaf(t, r, q);
    q = a1 * r;
    t = fmaf(nb, q, a1);
    q = fmaf(t, r, q);
    t = fmaf(nb, q, a1);
    v = fmaf(t, r, q);
#endif
}

// Rounded pixel of one projected voxel (IEEE divides); a0,a1,a2 = fp32 row results.
__device__ __forceinline__ bool pixel_of(float a0, float a1, float a2, int W, int H,
                                         int &pix) {
    return pixel_from_quotients(a0 / a2, a1 / a2, W, (float)W - 0.5f, (float)H - 0.5f, pix);
}

// One row of M * world from its four exact fp64 products (p0: the y term, p1: the x term,
// p2: the z term of Model::toWord's swapped coordinates, p3 = M[r][3]).
template <bool LEFT>
__device__ __forceinline__ float row_sum(double p0, double p1, double p2, double p3) {
    if (LEFT) return (float)(((p0 + p1) + p2) + p3);
    return (float)(p0 + ((p1 + p2) + p3));
}

}  // namespace arvx
