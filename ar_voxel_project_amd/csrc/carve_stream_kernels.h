// carve_stream_kernels.h -- the dense carve of a FRESH model in ONE persistent launch (round 4).
//
// Replaces, for the case the benchmark and a drop-in carve() of a new Model are (reference
// src/VoxelCarving.cpp:60-72 on a model from src/Model.cpp:9-14), the chain
//     carve_coarse_kernel -> carve_classify_dense_kernel -> carve_exact_blocks_kernel
// of carve_kernels.h by one grid of resident workgroups (4 per compute unit) that go through the
// same three kinds of work without a launch boundary, a grid barrier or a sorted hand-over:
//   A  coarse units   (cwA coarse tiles x all views, lane = (tile, view) pair): codes of the
//                     decided tiles (lazy state, arvx_device.h) and a LIST ENTRY per undecided one;
//   B  sub-tile units (a listed coarse tile or a quarter of one, lane = (sub-tile, view slot)):
//                     settled records written, the others published as ITEMS in eight weight classes;
//   C  items          (one wave each, block map, exact projection): the per-voxel work.
// Work moves from workgroup to workgroup as tagged 8-byte GRANULES (arvx_device.h: one sc1 store,
// sc1 loads, valid when the tag is this launch's) -- a list entry or an item is complete when every
// one of its granules carries the tag, so there is no flag, no fence and no ordering to get wrong;
// units and items are handed out by tickets (sharded counters), never by position in the grid.
// A workgroup does A units while its shard has any, waits for the A phase to be complete (ONE
// granule, written by whoever finishes the last A unit: the list length), does B units until every
// B ticket is drawn, and then its waves take items independently, heaviest class first: first
// what the workgroup itself kept from its B units (up to one item per wave, in LDS: no wait at
// all), then, once the B phase is complete (one line of eight granules: the final item counts),
// from the queues.  So the latency chains of A and B run beside the first items instead of in front
// of all of them, and the matrices of all views sit in LDS (12 floats each) for every rectangle
// test of the launch.
// What a waiting wave polls is ONE line that nobody adds to (the first version of this kernel let
// idle waves look at all ticket counters and at the slots of items yet to come, with short sleeps:
// 4096 pollers x 128 lines -- the B phase took 45-110 us instead of 10, an item 31 us to get).
//
// Forward progress never depends on which workgroups are resident (several of these launches may
// share the chip: jobs in flight on other streams): a wait is only ever for work that some RUNNING
// workgroup has taken -- B waits for A units (all taken by ticket; a wait that lasts checks for
// coarse units nobody took and does them itself), C waits for B units (a workgroup enters C only
// when all B tickets are drawn) -- and A and B units themselves wait for nothing.  Every wait gives
// up after about a second and marks Ctx::h_fault instead of hanging the device.
// The control block is all zero between launches: the last workgroup to leave resets it.
#pragma once

#include "carve_kernels.h"

namespace arvx {

constexpr int kStreamShards = 64;   // ticket shards of the A units
constexpr int kStreamShardsB = 16;  // ... of the B units: shard s holds the odd (s >= 8) or even units of list s % 8
constexpr int kStreamLists = 8;     // the list of undecided coarse tiles, in eight parts (A unit u appends to u % 8)
// Counters (one 256-byte block each: counters that share a line serialise, and ONE word takes about
// 88 atomics per microsecond -- nothing here is added to by more than a few dozen waves at a time;
// "everything done" is counted per shard first and per launch by the shards' last arrivals).
enum : int {
    kSA_Next = 0,                            // [64] coarse-unit tickets, unit u in shard u % 64
    kSA_Done = kSA_Next + kStreamShards,     // [64] ... finished
    kSB_Next = kSA_Done + kStreamShards,     // [16] sub-tile-unit tickets
    kSB_Done = kSB_Next + kStreamShardsB,    // [16] ... finished
    kSC_Count = kSB_Done + kStreamShardsB,   // [64] items appended per list (class * 8 + sub-tile % 8)
    kSC_Pool = kSC_Count + kWorkLists,       // [8] tickets of the shared part of the items
    kSX_Done = kSC_Pool + kPoolCounters,     // [64] workgroups that have left, workgroup w in shard w % 64
    kSL_Res = kSX_Done + kStreamShards,      // [8] list entries reserved per list part
    kS_ATop = kSL_Res + kStreamLists,        // shards whose A units / B units / workgroups are all done
    kS_BTop,
    kS_XTop,
    kStreamCounters
};
// behind the counters, tagged (never reset): the eight granules of block kStreamFlagA = the lengths of
// the list parts, written when the last A unit is done; the 64 granules of blocks kStreamFlagB.. = the
// items per list, written when the last B unit is done
constexpr int kStreamFlagA = kStreamCounters, kStreamFlagB = kStreamCounters + 1;
constexpr int kStreamLines = kStreamCounters + 3;
constexpr unsigned kStreamSpinLimit = 1u << 19;

constexpr int kStreamTilesA = 4;  // coarse tiles per A unit, at most

struct StreamLds {
    float M[kMaxChunks * 64 * 12];  // every view's matrix
    // A: a wave's unit (the waves of a workgroup work independently)
    float box[4][kStreamTilesA][6];
    unsigned long long amixed[4][kStreamTilesA][kMaxChunks], afg[4][kStreamTilesA][kMaxChunks];
    int acarved[4][kStreamTilesA];
    int waves_left;  // waves of the workgroup still at work
};

__device__ __forceinline__ int *sctr(const CarveParams &p, int k) {
    return p.sctl + (size_t)k * kCounterStride;
}
__device__ __forceinline__ int peek(const int *c) {
    return __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one ticket for the wave (every lane gets it)
__device__ __forceinline__ int wave_ticket(int *c, int lane) {
    int t = 0;
    if (lane == 0) t = atomicAdd(c, 1);
    return __builtin_amdgcn_readfirstlane(t);
}
__device__ __forceinline__ void stream_fault(const CarveParams &p, unsigned mark, int lane) {
    if (lane == 0 && p.fault)
        __hip_atomic_store(p.fault, mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// units of shard k when there are n in all (unit u sits in shard u % 64)
__device__ __forceinline__ int shard_count64(int n, int k) { return n > k ? (n - k + 63) >> 6 : 0; }
__device__ __forceinline__ unsigned long long *stream_flag(const CarveParams &p, int block) {
    return reinterpret_cast<unsigned long long *>(p.sctl + (size_t)block * kCounterStride);
}
// sleeps that grow with the wait: 0.1 us at first, 0.85 us from the fifth on
__device__ __forceinline__ void stream_backoff(unsigned spin) {
    if (spin < 2) __builtin_amdgcn_s_sleep(4);
    else if (spin < 4) __builtin_amdgcn_s_sleep(12);
    else __builtin_amdgcn_s_sleep(32);
}
// One arrival (the wave's lane 0) at shard counter `shard`, which expects `count`; the shard's last
// arrival arrives at `top`, which expects `nshards`.  True (in every lane) for the arrival that
// completes the launch's.  Every other arrival -- and whatever its wave had got back from memory
// before it -- is then performed: a returning atomic is issued after the ones before it returned.
__device__ __forceinline__ bool stream_arrive(const CarveParams &p, int shard, int count, int top,
                                              int nshards, int lane) {
    int last = 0;
    if (lane == 0 && atomicAdd(sctr(p, shard), 1) + 1 == count)
        last = atomicAdd(sctr(p, top), 1) + 1 == nshards;
    return __builtin_amdgcn_readfirstlane(last) != 0;
}
// the B phase is complete: the final length of every item list (lane l: list l), for everybody.
// (Read with an atomic: the value at the point where the appends were performed.)
__device__ __forceinline__ void stream_publish_counts(const CarveParams &p, const int lane) {
    granule_store(stream_flag(p, kStreamFlagB) + lane, (unsigned)atomicAdd(sctr(p, kSC_Count + lane), 0),
                  p.epoch);
}

// ---- A: one coarse unit, one wave (the job of carve_coarse_kernel for cwA coarse tiles) ----------
// Lane = (tile, view) pair, the unit's pairs dealt to the lanes densely, four passes at a time:
// the arithmetic of all four first, then their table reads together, then the answers.
__device__ __forceinline__ void stream_coarse_unit(const CarveParams &p, StreamLds &L, const int u,
                                                   const int wave, const int lane) {
    const int cw = p.cwA;
    const int ncoarse = p.coarseX * p.coarseY * p.coarseZ;
    const int t0 = u * cw, nt = min(cw, ncoarse - t0);
    if (lane < nt) {
        const int ct = t0 + lane;
        const int cx = ct % p.coarseX;
        const int cy = (ct / p.coarseX) % p.coarseY;
        const int cz = ct / (p.coarseX * p.coarseY);
        const int cyN = 8 << p.cyShift, czN = 8 << p.czShift;
        const int x0 = cx * kCoarseX, y0 = cy * cyN, z0 = cz * czN;
        // (striped slabs: the box spans the foreign planes in between as well -- conservative)
        const BoxW b = make_box(p.s, x0, min(x0 + kCoarseX - 1, p.X - 1), y0,
                                min(y0 + cyN - 1, p.Y - 1), global_z(p, z0),
                                global_z(p, min(z0 + czN - 1, p.Z - 1)));
        float *o = L.box[wave][lane];
        o[0] = b.wy0, o[1] = b.wy1, o[2] = b.wx0, o[3] = b.wx1, o[4] = b.wz0, o[5] = b.wz1;
#pragma unroll
        for (int c = 0; c < kMaxChunks; ++c) L.amixed[wave][lane][c] = L.afg[wave][lane][c] = 0ull;
        L.acarved[wave][lane] = 0;
    }
    wave_lds_sync();
    const int V = p.v1 - p.v0;
    const float rV = 1.0f / (float)V;
    const int npairs = nt * V;
    for (int q0 = 0; q0 < npairs; q0 += 256) {
        RectQ rq[4];
        int tile[4], view[4];
        int s00[4], s01[4], s10[4], s11[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = q0 + 64 * k + lane;
            rq[k].code = kClsOut;
            tile[k] = view[k] = 0;
            if (q < npairs) {
                // q = tile * V + view  (q < 4 * 256: the float quotient is off by at most one)
                int t = (int)((float)q * rV);
                int vw = q - t * V;
                if (vw < 0) {
                    --t;
                    vw += V;
                } else if (vw >= V) {
                    ++t;
                    vw -= V;
                }
                tile[k] = t;
                view[k] = vw;
                const float *sb = L.box[wave][t];
                BoxW box;
                box.wy0 = sb[0], box.wy1 = sb[1], box.wx0 = sb[2], box.wx1 = sb[3], box.wz0 = sb[4],
                box.wz1 = sb[5];
                float Mr[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) Mr[i] = L.M[12 * vw + i];
                rq[k] = rect_prepare(Mr, box, p.W, p.H, p.satW);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s00[k] = s01[k] = s10[k] = s11[k] = 0;
            if (rq[k].code < 0) {
                const sat_t *e = p.sat + (size_t)(p.v0 + view[k]) * p.satStride + rq[k].base;
                s00[k] = e[0];
                s01[k] = e[rq[k].dx];
                s10[k] = e[rq[k].dy * p.satW];
                s11[k] = e[rq[k].dy * p.satW + rq[k].dx];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (q0 + 64 * k + lane >= npairs) continue;
            const int cls = (rq[k].code >= 0 ? rq[k].code : rect_finish(rq[k], s00[k], s01[k], s10[k], s11[k])) & 3;
            const unsigned long long bit = 1ull << (view[k] & 63);
            if (cls == kClsMixed) atomicOr(&L.amixed[wave][tile[k]][view[k] >> 6], bit);
            if (cls == kClsFg) atomicOr(&L.afg[wave][tile[k]][view[k] >> 6], bit);
            if (cls == kClsCarved) L.acarved[wave][tile[k]] = 1;  // (any writer writes the same)
        }
    }
    wave_lds_sync();
    // codes, and a list entry per undecided tile
    bool und = false, any_fg = false;
    if (lane < nt) {
        bool any_mixed = false;
        for (int c = 0; c < p.nchunks; ++c) {
            any_mixed = any_mixed || L.amixed[wave][lane][c];
            any_fg = any_fg || L.afg[wave][lane][c];
        }
        // 1: some view carves the whole tile.  2 / 3: no view needs a closer look and none
        // carves -- every voxel keeps its occupancy and is seen (2) or not even seen (3).
        const int code = L.acarved[wave][lane] ? 1 : (any_mixed ? 0 : (any_fg ? 2 : 3));
        p.coarseCarved[t0 + lane] = (uint8_t)code;  // (lazy state: the code IS a decided tile's state)
        if (p.cstate) p.cstate[t0 + lane] = code == 1 ? 3 : (code == 2 ? 2 : 0);
        und = code == 0;
    }
    const unsigned long long um = __ballot(und);
    const int part = u & (kStreamLists - 1);  // the list part this unit appends to
    if (um) {
        int base = 0;
        if (lane == 0) base = atomicAdd(sctr(p, kSL_Res + part), __popcll(um));
        base = __builtin_amdgcn_readfirstlane(base);
        if (und) {
            const int pos = base + __popcll(um & ((1ull << lane) - 1ull));
            unsigned long long *e = p.listG + ((size_t)part * p.listCap + pos) * p.listStride;
            granule_store(e, (unsigned)(t0 + lane) | (any_fg ? 0x80000000u : 0u), p.epoch);
            for (int c = 0; c < p.nchunks; ++c) {
                const unsigned long long m = L.amixed[wave][lane][c];
                granule_store(e + 1 + 2 * c, (unsigned)m, p.epoch);
                granule_store(e + 2 + 2 * c, (unsigned)(m >> 32), p.epoch);
            }
        }
    }
    wave_lds_sync();  // (the wave's next unit reuses its LDS arrays)
    // (after the reservation has returned.)  Whoever finishes the LAST unit publishes the lengths
    // of the list parts.
    const int sh = u & (kStreamShards - 1);
    if (stream_arrive(p, kSA_Done + sh, shard_count64(p.nA, sh), kS_ATop, min(kStreamShards, p.nA), lane)) {
        int len = 0;
        if (lane < kStreamLists) {
            len = atomicAdd(sctr(p, kSL_Res + lane), 0);
            granule_store(stream_flag(p, kStreamFlagA) + lane, (unsigned)len, p.epoch);
        }
        if (!__ballot(len != 0)) stream_publish_counts(p, lane);  // no B unit will ever do it
    }
}

// ---- B: the lengths of the list parts, once the A phase is complete --------------------------------
// A unit = 16 sub-tiles of a listed coarse tile (a quarter of it; half on striped slabs).  Unit j of
// list part k sits in shard k + 8 (j % 2).  cnt: lane s < 16 gets the units of shard s, nsh the shards
// that have any.  Returns -1, or -2 - u: after a long wait, coarse unit u that nobody had taken (its
// workgroup is not resident) -- the caller does it and calls again.
__device__ __forceinline__ int stream_wait_lists(const CarveParams &p, const int lane, int &cnt, int &nsh) {
    for (unsigned spin = 0;; ++spin) {
        unsigned long long g = 0;
        bool ok = true;
        if (lane < kStreamLists) {
            g = granule_load(stream_flag(p, kStreamFlagA) + lane);
            ok = (uint32_t)(g >> 32) == p.epoch;
        }
        if (__all(ok)) {
            // lane l < 16: the units of list part l % 8 with index % 2 == l / 8
            const int n = __shfl((int)(uint32_t)g, lane & 7) << p.splitLog2;
            cnt = lane < kStreamShardsB ? (n > (lane >> 3) ? (n - (lane >> 3) + 1) >> 1 : 0) : 0;
            nsh = __popcll(__ballot(cnt > 0));
            return -1;
        }
        if ((spin & 31u) == 31u) {
            // waiting for a while: a coarse unit nobody has taken (its workgroup is not resident)?
            const int nxt = peek(sctr(p, kSA_Next + lane));
            const unsigned long long av = __ballot(nxt < shard_count64(p.nA, lane));
            if (av) {
                const int k = __ffsll((long long)av) - 1;
                const int t = wave_ticket(sctr(p, kSA_Next + k), lane);
                if (t < shard_count64(p.nA, k)) return -2 - (k + kStreamShards * t);
            }
        }
        if (spin > kStreamSpinLimit) {
            stream_fault(p, 2u, lane);
            cnt = 0;
            nsh = 0;
            return -1;
        }
        stream_backoff(spin);
    }
}

// ---- B: one sub-tile unit, one wave (the job of carve_classify_dense_kernel for 16 sub-tiles) ----
// Lane = (sub-tile lane % 16, view slot lane / 16): the coarse tile's mixed views four at a time,
// eight views per round -- the arithmetic of the round first, then its table reads together.
// NCH: chunks of 64 views (a template parameter: the masks live in registers).
template <int NCH>
__device__ __forceinline__ void stream_subtile_unit(const CarveParams &p, const StreamLds &L,
                                                    const int j, const unsigned pay, const int lane) {
    // j: the unit's number in its list part: coarse tile j >> split, then the piece
    const unsigned g0 = (unsigned)__builtin_amdgcn_readlane((int)pay, 0);
    const int ct = (int)(g0 & 0x7fffffffu);
    const bool anyfg = g0 >> 31;
    const int cx = ct % p.coarseX;
    const int cty = ((ct / p.coarseX) % p.coarseY) << p.cyShift;
    const int ctz = (ct / (p.coarseX * p.coarseY)) << p.czShift;
    const int s0 = (j & ((1 << p.splitLog2) - 1)) * 16;  // the unit's first sub-tile (of the coarse tile's)
    const int sub = lane & 15, vslot = lane >> 4;
    // this lane's sub-tile
    const int sidx = s0 + sub, tl = sidx >> 2, sw = sidx & 3;
    const int tx = cx;
    const int ty = cty + (tl & ((1 << p.cyShift) - 1)), tz = ctz + (tl >> p.cyShift);
    const int sx0 = cx * kTileX + sw * kSubX, sy0 = ty * kTileY, sz0 = tz * kTileZ;
    // (tiles and sub-tiles of an edge coarse tile that lie outside the grid keep the "finished"
    // records they were allocated with)
    const bool in_grid = ty < p.tilesY && tz < p.tilesZ && sx0 < p.X;
    BoxW box = make_box(p.s, 0, 0, 0, 0, 0, 0);
    if (in_grid)
        box = make_box(p.s, sx0, min(sx0 + kSubX - 1, p.X - 1), sy0, min(sy0 + kTileY - 1, p.Y - 1),
                       global_z(p, sz0), global_z(p, min(sz0 + kTileZ - 1, p.Z - 1)));
    unsigned flag = anyfg ? 2u : 0u;  // inherited: the coarse rectangle contains every sub-tile's
    unsigned long long mixed_c[NCH], fast_c[NCH];
#pragma unroll
    for (int chunk = 0; chunk < NCH; ++chunk) {
        mixed_c[chunk] = fast_c[chunk] = 0;
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)pay, 1 + 2 * chunk);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)pay, 2 + 2 * chunk);
        unsigned long long cm = ((unsigned long long)hi << 32) | lo;  // (scalar)
        while (cm) {
            RectQ rq[2];
            int myb[2];
            int s00[2], s01[2], s10[2], s11[2];
#pragma unroll
            for (int g = 0; g < 2; ++g) {  // two groups of four views; this lane's view of each
                int vb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    vb[k] = -1;
                    if (cm) {
                        vb[k] = __ffsll((long long)cm) - 1;
                        cm &= cm - 1;
                    }
                }
                myb[g] = vslot == 0 ? vb[0] : (vslot == 1 ? vb[1] : (vslot == 2 ? vb[2] : vb[3]));
                rq[g].code = kClsOut;
                if (myb[g] >= 0 && in_grid) {
                    const int view = 64 * chunk + myb[g];
                    float Mr[12];
#pragma unroll
                    for (int i = 0; i < 12; ++i) Mr[i] = L.M[12 * view + i];
                    rq[g] = rect_prepare(Mr, box, p.W, p.H, p.satW);
                }
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                s00[g] = s01[g] = s10[g] = s11[g] = 0;
                if (rq[g].code < 0) {
                    const sat_t *e = p.sat + (size_t)(p.v0 + 64 * chunk + myb[g]) * p.satStride + rq[g].base;
                    s00[g] = e[0];
                    s01[g] = e[rq[g].dx];
                    s10[g] = e[rq[g].dy * p.satW];
                    s11[g] = e[rq[g].dy * p.satW + rq[g].dx];
                }
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                if (myb[g] < 0) continue;
                int cls = rq[g].code >= 0 ? rq[g].code : rect_finish(rq[g], s00[g], s01[g], s10[g], s11[g]);
                if (cls & kFastDiv) fast_c[chunk] |= 1ull << myb[g];
                cls &= 3;
                if (cls == kClsMixed) mixed_c[chunk] |= 1ull << myb[g];
                if (cls == kClsCarved) flag |= 1u;
                if (cls == kClsFg) flag |= 2u;
            }
        }
    }
    // what the four view slots of a sub-tile found, together (in every one of its lanes)
    flag |= __shfl_xor(flag, 16);
    flag |= __shfl_xor(flag, 32);
    int nmixed = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        unsigned lo = (unsigned)mixed_c[c], hi = (unsigned)(mixed_c[c] >> 32);
        lo |= __shfl_xor(lo, 16), hi |= __shfl_xor(hi, 16);
        lo |= __shfl_xor(lo, 32), hi |= __shfl_xor(hi, 32);
        mixed_c[c] = ((unsigned long long)hi << 32) | lo;
        lo = (unsigned)fast_c[c], hi = (unsigned)(fast_c[c] >> 32);
        lo |= __shfl_xor(lo, 16), hi |= __shfl_xor(hi, 16);
        lo |= __shfl_xor(lo, 32), hi |= __shfl_xor(hi, 32);
        fast_c[c] = ((unsigned long long)hi << 32) | lo;
        nmixed += __popcll(mixed_c[c]);
    }
    // ---- the records the unit settles, 16 bytes per lane and turn (chunk c of a record = the eight
    // occupancy rows of plane c (c < 8) or the eight seen rows of plane c - 8)
    const unsigned settled = (in_grid ? 1u : 0u) | ((flag & 1u) ? 2u : 0u) | (nmixed ? 4u : 0u) |
                             ((flag & 2u) ? 8u : 0u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int sl = 4 * i + (lane >> 4), c = lane & 15;  // sub-tile sl of the unit, chunk c
        const unsigned st = (unsigned)__shfl((int)settled, sl);
        const int sidx2 = s0 + sl, tl2 = sidx2 >> 2, sw2 = sidx2 & 3;
        const int ty2 = cty + (tl2 & ((1 << p.cyShift) - 1)), tz2 = ctz + (tl2 >> p.cyShift);
        const int x0 = cx * kTileX + sw2 * kSubX;
        if (!(st & 1u)) continue;  // outside the grid
        const bool any_carved = st & 2u, any_mixed = st & 4u, any_fg = st & 8u;
        const bool seen_half = c >= 8;
        uint4 v;
        if (any_carved) {  // carved implies seen (src/VoxelCarving.cpp:50-54)
            v.x = v.y = v.z = v.w = seen_half ? 0xffffffffu : 0u;
        } else if (!any_mixed) {
            // a fresh model's record from constants: occupied inside the grid, seen where a view
            // sees the whole box (outside the grid: always)
            const int z = tz2 * kTileZ + (c & 7);
            const int ny = z < p.Z ? min(kTileY, p.Y - ty2 * kTileY) : 0;  // rows inside
            const uint32_t xm = 0xffffu >> (kSubX - min(kSubX, p.X - x0));
            uint32_t w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t in = (2 * k < ny ? xm : 0u) | (2 * k + 1 < ny ? xm << 16 : 0u);
                w[k] = !seen_half ? in : (any_fg ? 0xffffffffu : ~in);
            }
            v.x = w[0];
            v.y = w[1];
            v.z = w[2];
            v.w = w[3];
        } else {
            continue;  // an item: its wave writes the record
        }
        uint16_t *const rec = p.rec + rec_index(p, cx, ty2, tz2, sw2) * kRecU16;
        reinterpret_cast<uint4 *>(rec)[c] = v;
    }
    // ---- ... and the items: lanes 0..15 = the sub-tiles, so that the unit's appends to the item
    // lists (an atomic whose answer the item's place depends on) are in flight together.  Eight
    // weight classes of eight lists, most views to evaluate first; consecutive sub-tiles go to
    // consecutive lists of their class (a list never gets more than every 8th sub-tile of the grid).
    if (lane < 16 && in_grid && !(flag & 1u) && nmixed) {
        const int wclass = 7 - min(7, nmixed * 8 / (p.v1 - p.v0 + 1));
        const int list = wclass * 8 + (((((tz * p.tilesY + ty) * p.tilesX + tx) << 2) + sw) & 7);
        const int pos = atomicAdd(sctr(p, kSC_Count + list), 1);
        unsigned long long *e = p.itemG + ((size_t)list * p.workCap + (size_t)pos) * p.itemStride;
        granule_store(e, (unsigned)tx | ((unsigned)ty << 16), p.epoch);
        granule_store(e + 1, (unsigned)tz | ((unsigned)sw << 16) | ((flag & 2u) ? 1u << 18 : 0u), p.epoch);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            granule_store(e + 2 + 4 * c, (unsigned)mixed_c[c], p.epoch);
            granule_store(e + 3 + 4 * c, (unsigned)(mixed_c[c] >> 32), p.epoch);
            granule_store(e + 4 + 4 * c, (unsigned)fast_c[c], p.epoch);
            granule_store(e + 5 + 4 * c, (unsigned)(fast_c[c] >> 32), p.epoch);
        }
    }
}

// ---- C: the final lengths of the item lists (lane l: list l), once the B phase is complete ------
__device__ __forceinline__ bool stream_wait_counts(const CarveParams &p, const int lane, int &cnt) {
    for (unsigned spin = 0;; ++spin) {
        const unsigned long long g = granule_load(stream_flag(p, kStreamFlagB) + lane);
        if (__all((uint32_t)(g >> 32) == p.epoch)) {
            cnt = (int)(uint32_t)g;
            return true;
        }
        if (spin > kStreamSpinLimit) {
            stream_fault(p, 3u, lane);
            return false;
        }
        stream_backoff(spin);
    }
}
// ... and item `it` (list * workCap + place): lane k gets the payload of its granule k
__device__ __forceinline__ bool stream_fetch_item(const CarveParams &p, const int lane, const size_t it,
                                                  unsigned &pay) {
    const unsigned long long *e = p.itemG + it * p.itemStride;
    for (unsigned spin = 0;; ++spin) {  // (published before the lengths; in flight at most)
        unsigned long long g = 0;
        bool ok = true;
        if (lane < p.itemStride) {
            g = granule_load(e + lane);
            ok = (uint32_t)(g >> 32) == p.epoch;
        }
        if (__all(ok)) {
            pay = (unsigned)g;
            return true;
        }
        if (spin > kStreamSpinLimit) {
            stream_fault(p, 4u, lane);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// ---- C: one item (the job of carve_exact_blocks_kernel for one sub-tile) ------------------------
#ifndef ARVX_BT_BATCH
#define ARVX_BT_BATCH 2  // passes (of four views) per round of the block tests
#endif
template <bool LEFT, int NCH>
__device__ __forceinline__ void stream_exact_item(const CarveParams &p, const StreamLds &L, const int lane,
                                                  const unsigned pay, const int wclass) {
    const int lx = lane & 3, ly = (lane >> 2) & 3, lz = lane >> 4;  // block map: a voxel per block
    // the launch ends on its longest items: the heavy classes get the SIMD's issue slots first
    switch (wclass >> 1) {
        case 0: __builtin_amdgcn_s_setprio(3); break;
        case 1: __builtin_amdgcn_s_setprio(2); break;
        case 2: __builtin_amdgcn_s_setprio(1); break;
        default: __builtin_amdgcn_s_setprio(0); break;
    }
    const unsigned g0 = (unsigned)__builtin_amdgcn_readlane((int)pay, 0);
    const unsigned g1 = (unsigned)__builtin_amdgcn_readlane((int)pay, 1);
    const int tx = (int)(g0 & 0xffffu), ty = (int)(g0 >> 16);
    const int tz = (int)(g1 & 0xffffu), wave = (int)((g1 >> 16) & 3u);
    const bool fg_seen = (g1 >> 18) & 1u;  // seen by an all-foreground view
    const SubTile t = subtile_of(p, tx, ty, tz, wave, lane);
    uint16_t *const rec = p.rec + rec_index(p, tx, ty, tz, wave) * kRecU16;
    // st[byi * 2 + bzi] byte j = voxel (4 j + lx, 4 byi + ly, 4 bzi + lz) of the sub-tile, bit0
    // occupied, bit1 seen.  A fresh model: all occupied, none seen, nothing to load.
    uint32_t st[4];
#pragma unroll
    for (int byi = 0; byi < 2; ++byi)
#pragma unroll
        for (int bzi = 0; bzi < 2; ++bzi) {
            const int r = (4 * bzi + lz) * 8 + 4 * byi + ly;
            const uint32_t o = row_inmask(p, tx, ty, tz, wave, r);
            const uint32_t sn = fg_seen ? 0xffffu : (~o & 0xffffu);
            st[2 * byi + bzi] = spread4((o >> lx) & 0x1111u) | (spread4((sn >> lx) & 0x1111u) << 1);
        }
    // world coordinates as the reference's toWord gives them (fp32, src/Model.h:134-140)
    float wx[4], wy[2], wz[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) wx[j] = (float)(t.sx0 + 4 * j + lx) * p.s;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        wy[b] = (float)(t.sy0 + 4 * b + ly) * p.s;
        wz[b] = (float)(-global_z(p, t.sz0 + 4 * b + lz)) * p.s;
    }
    bool done = false;
    for (int c = 0; c < NCH && !done; ++c) {
        const unsigned m_lo = (unsigned)__builtin_amdgcn_readlane((int)pay, 2 + 4 * c);
        const unsigned m_hi = (unsigned)__builtin_amdgcn_readlane((int)pay, 3 + 4 * c);
        const unsigned f_lo = (unsigned)__builtin_amdgcn_readlane((int)pay, 4 + 4 * c);
        const unsigned f_hi = (unsigned)__builtin_amdgcn_readlane((int)pay, 5 + 4 * c);
        unsigned long long mixed = ((unsigned long long)m_hi << 32) | m_lo;
        const unsigned long long fastdiv = ((unsigned long long)f_hi << 32) | f_lo;
        // block-level rectangle tests first: what they settle is applied at once
        unsigned bcarved = 0, bseen = 0;
#if ARVX_BT_BATCH > 0
        const unsigned needLanes = block_tests_lds<ARVX_BT_BATCH>(p, L.M, t, 64 * c, mixed, lane, bcarved, bseen);
#else
        const unsigned needLanes = block_tests(p, t, p.v0 + 64 * c, mixed, 0, 0, lane, bcarved, bseen);
#endif
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            uint32_t w = st[m];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if ((bseen >> (4 * m + j)) & 1u) w |= 2u << (8 * j);
                if ((bcarved >> (4 * m + j)) & 1u) w = (w & ~(0xffu << (8 * j))) | (2u << (8 * j));
            }
            st[m] = w;
        }
        done = __all(st[0] == kDone4 && st[1] == kDone4 && st[2] == kDone4 && st[3] == kDone4);
        int slot = 0;
        while (mixed && !done) {
            const int b = __ffsll((long long)mixed) - 1;
            mixed &= mixed - 1;
            const unsigned need = (unsigned)__builtin_amdgcn_readlane((int)needLanes, slot);
            ++slot;
            if (!need) continue;  // every block settled by its rectangle
            done = exact_view_blocks<LEFT>(p, __builtin_amdgcn_readfirstlane(p.v0 + 64 * c + b),
                                           (fastdiv >> b) & 1ull, wy, wx, wz, st, need);
        }
    }
    // blocks -> record: a row's 16 bits sit in the four neighbouring lanes lx = 0..3, four bits
    // each; lane lx = 0 writes the row's two entries
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
            rec[r] = (uint16_t)both;
            rec[64 + r] = (uint16_t)(both >> 16);
        }
}

#ifdef ARVX_TIMELINE  // diagnostic build (tools/stream_timeline.py): 8 words per wave --
// start, end of A, end of B, first item in hand, end, items, ticks spent taking items, A/B units
#define ARVX_TL(k) do { if (tl) tl[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ARVX_TL(k) do { } while (0)
#endif

template <bool LEFT, int NCH>
__global__ __launch_bounds__(256, 4) void carve_stream_kernel(const CarveParams p) {
    __shared__ StreamLds L;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#ifdef ARVX_TIMELINE
    unsigned long long *tl = (lane == 0 && p.timeline) ? p.timeline + 16ull * (blockIdx.x * 4 + wave) : nullptr;
    unsigned long long tl_items = 0, tl_take = 0, tl_units = 0, tl_draws = 0, tl_first_draw = 0, tl_bunit = 0;
    unsigned long long tl_bt[4] = {0, 0, 0, 0};
    ARVX_TL(0);
#endif
    {   // every view's matrix: 12 floats
        const int n = (p.v1 - p.v0) * 12;
        for (int i = threadIdx.x; i < n; i += 256) L.M[i] = p.M[12 * p.v0 + i];
        if (threadIdx.x == 0) L.waves_left = 4;
    }
    __syncthreads();
    const int G = (int)gridDim.x;
    const int wid = (int)blockIdx.x * 4 + wave;  // this wave, of 4 G
    const int home = wid & (kStreamShards - 1);
    // ---- A: this wave's share of its shard's coarse units (the others of the shard take theirs;
    // a unit nobody takes is found by the waits in stream_draw_b)
    {
        const int sharers = (4 * G - home + kStreamShards - 1) / kStreamShards;  // waves with this home
        const int cnt = shard_count64(p.nA, home);
        const int quota = (cnt + sharers - 1) / sharers;
        for (int n = 0; n < quota; ++n) {
            const int t = wave_ticket(sctr(p, kSA_Next + home), lane);
            if (t >= cnt) break;
            stream_coarse_unit(p, L, home + kStreamShards * t, wave, lane);
#ifdef ARVX_TIMELINE
            ++tl_units;
#endif
        }
    }
    ARVX_TL(1);
    // ---- B: sub-tile units until every B ticket of the launch is drawn.  What a unit waits for is
    // requested early: its ticket while the wave still waits for the A phase (or works on the unit
    // before), its list entry together with the arrival of the unit before.
    {
        int s = home & (kStreamShardsB - 1), cnt = -1, nsh = 0;
        int tv = 0;  // (lane 0) the ticket drawn ahead
        if (lane == 0) tv = atomicAdd(sctr(p, kSB_Next + s), 1);
        for (;;) {
            const int r = stream_wait_lists(p, lane, cnt, nsh);
            if (r == -1) break;
            stream_coarse_unit(p, L, -2 - r, wave, lane);
        }
        int arr = 0, arr_sh = -1;  // (lane 0) the latest unit's arrival at its shard counter, to be looked at
        bool publish = false;
        for (;;) {
            const int t = __builtin_amdgcn_readfirstlane(tv);
            const bool have = t < __builtin_amdgcn_readlane(cnt, s);
            unsigned long long g = 0;
            const unsigned long long *e = nullptr;
            int j = 0;
            if (have) {  // the unit's list entry: on its way while the arrival below is looked at
                j = 2 * t + (s >> 3);
                e = p.listG + ((size_t)(s & 7) * p.listCap + (size_t)(j >> p.splitLog2)) * p.listStride;
                if (lane < p.listStride) g = granule_load(e + lane);
            }
            if (arr_sh >= 0) {  // the shard's last arrival arrives at the top; the top's last publishes
                int last = 0;
                if (lane == 0 && arr + 1 == __builtin_amdgcn_readlane(cnt, arr_sh))
                    last = atomicAdd(sctr(p, kS_BTop), 1) + 1 == nsh;
                publish = publish || __builtin_amdgcn_readfirstlane(last) != 0;
                arr_sh = -1;
            }
            if (!have) {
                // this shard has no unit left: one look at all shards
                const int nxt = lane < kStreamShardsB ? peek(sctr(p, kSB_Next + lane)) : 0;
                const unsigned av = (unsigned)__ballot(lane < kStreamShardsB && nxt < cnt);
                if (!av) break;
                const int h = home & (kStreamShardsB - 1);
                const unsigned rot = ((av >> h) | (av << (16 - h))) & 0xffffu;
                s = (h + __ffs((int)rot) - 1) & 15;
                if (lane == 0) tv = atomicAdd(sctr(p, kSB_Next + s), 1);
                continue;
            }
            if (lane == 0) tv = atomicAdd(sctr(p, kSB_Next + s), 1);  // the next unit's ticket
            for (unsigned spin = 0;; ++spin) {  // (published before the list lengths; in flight at most)
                const bool ok = lane >= p.listStride || (uint32_t)(g >> 32) == p.epoch;
                if (__all(ok)) break;
                if (spin > kStreamSpinLimit) {
                    stream_fault(p, 5u, lane);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                if (lane < p.listStride) g = granule_load(e + lane);
            }
            stream_subtile_unit<NCH>(p, L, j, (unsigned)g, lane);
            // (after the unit's appends have returned) its arrival: looked at in the next turn
            arr_sh = s;
            if (lane == 0) arr = atomicAdd(sctr(p, kSB_Done + s), 1);
#ifdef ARVX_TIMELINE
            tl_units += 1ull << 32;
#endif
        }
        // whoever finished the LAST unit publishes the final lengths of the item lists
        if (publish) stream_publish_counts(p, lane);
    }
    ARVX_TL(2);
    // ---- C: every wave for itself, as in carve_exact_blocks_kernel: one item by the wave's own
    // index (the lists in order are the items by weight), the rest by tickets
    {
        int cnt = 0;
        if (stream_wait_counts(p, lane, cnt)) {
            ARVX_TL(3);
            for_each_work_item_of<false>(p, lane, blockIdx.x * 4 + wave, p.nwaves, cnt,
                                         [&](const size_t it, const int, const int list, const int) {
                unsigned pay = 0;
                if (!stream_fetch_item(p, lane, it, pay)) return;
                stream_exact_item<LEFT, NCH>(p, L, lane, pay, list >> 3);
#ifdef ARVX_TIMELINE
                ++tl_items;
#endif
            });
        }
    }
    ARVX_TL(4);
#ifdef ARVX_TIMELINE
    if (tl) {
        tl[5] = tl_items;
        tl[6] = tl_take;
        tl[7] = tl_units;
        tl[8] = tl_first_draw;
        tl[9] = tl_bunit;
        tl[10] = tl_draws;
        tl[11] = tl_bt[0];
        tl[12] = tl_bt[1];
        tl[13] = tl_bt[2];
        tl[14] = tl_bt[3];
    }
#endif
    // ---- the last workgroup to leave resets the control block for the next launch
    if (lane == 0 && atomicSub(&L.waves_left, 1) == 1) {
        const int sh = (int)(blockIdx.x & (kStreamShards - 1));
        if (atomicAdd(sctr(p, kSX_Done + sh), 1) + 1 == shard_count64(G, sh) &&
            atomicAdd(sctr(p, kS_XTop), 1) + 1 == min(kStreamShards, G))
            for (int k = 0; k < kStreamCounters; ++k)
                __hip_atomic_store(sctr(p, k), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace arvx
