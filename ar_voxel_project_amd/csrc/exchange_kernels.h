// exchange_kernels.h -- the packed occupancy of a slab in a form that is cheap to ship.
//
// The end-of-carve collective moves the bit-packed occupancy of every slab to
// every rank; over xGMI that, not the carve, sets the pace of a multi-GPU step.  Most
// 64-bit words of the packed plane are all-zero (empty space) or all-one (inside the
// object): a slab is shipped as
//     [0]                  number of mixed words
//     [1, 1+nb)            bitmap of the all-one words            (nb = ceil(n / 64))
//     [1+nb, 1+2nb)        bitmap of the mixed words
//     [1+2nb, H)           per group of 64 words: how many mixed words precede it (u32)
//     [H, H+cap)           the mixed words, in order              (H = 1 + 2nb + ceil(nb/2))
// A receiver rebuilds the plain words with one kernel and no scan of its own.  If a slab
// has more than `cap` mixed words the packet says so in [0] and the receiver raises a
// flag: the caller then falls back to the plain all-gather for that exchange.
#pragma once

#include "arvx_device.h"
#include "mc_kernels.h"  // wg_exclusive_scan, mc_scan_blocks_kernel

namespace arvx {

__host__ __device__ inline long long occ_packet_header(long long n) {
    const long long nb = (n + 63) / 64;
    return 1 + 2 * nb + (nb + 1) / 2;
}

// Three launches: classify (bitmaps + the number of mixed words before each group INSIDE its
// workgroup's 32 groups + the workgroup's total), one workgroup scanning those totals, and
// the ordered write.  (First version: per-group counts, then the three-launch block scan of
// the marching-cubes hand-off -- five launches of 2..7 us each for 16 MB of input.)
constexpr int kOccGroupsPerWave = 8;
constexpr int kOccGroupsPerWg = 4 * kOccGroupsPerWave;

// a wave classifies eight groups of 64 words (eight 512-byte reads in flight, two ballots
// each) and lane l < 8 keeps the bitmaps of group l; the workgroup scans its 32 counts
__global__ __launch_bounds__(256) void occ_classify_kernel(const unsigned long long *__restrict__ words,
                                                           long long n,
                                                           unsigned long long *__restrict__ out,
                                                           int *__restrict__ wg_sum) {
    __shared__ long long wtot[4];
    const long long nb = (n + 63) / 64;
    const int lane = threadIdx.x & 63;
    const long long g0 =
        (long long)blockIdx.x * kOccGroupsPerWg + (threadIdx.x >> 6) * kOccGroupsPerWave;
    unsigned long long w[kOccGroupsPerWave];
#pragma unroll
    for (int it = 0; it < kOccGroupsPerWave; ++it) {
        const long long i = (g0 + it) * 64 + lane;
        w[it] = i < n ? words[i] : 0ull;
    }
    unsigned long long my_ones = 0ull, my_mixed = 0ull;
#pragma unroll
    for (int it = 0; it < kOccGroupsPerWave; ++it) {
        // (a word past n reads as 0: neither all-one nor mixed)
        const unsigned long long ones = __ballot(w[it] == ~0ull);
        const unsigned long long mixed = __ballot(w[it] != 0ull && w[it] != ~0ull);
        if (lane == it) {
            my_ones = ones;
            my_mixed = mixed;
        }
    }
    const long long g = g0 + lane;
    long long total;
    const long long pre = wg_exclusive_scan((long long)__popcll(my_mixed), wtot, &total);
    if (lane < kOccGroupsPerWave && g < nb) {
        out[1 + g] = my_ones;
        out[1 + nb + g] = my_mixed;
        reinterpret_cast<unsigned *>(out + 1 + 2 * nb)[g] = (unsigned)pre;  // + its workgroup's offset later
    }
    if (threadIdx.x == 0) wg_sum[blockIdx.x] = (int)total;
}

// one wave per group: the group's final offset and its mixed words, in order
__global__ __launch_bounds__(256) void occ_write_kernel(const unsigned long long *__restrict__ words,
                                                        long long n, long long cap,
                                                        const long long *__restrict__ wg_off,
                                                        int nwg,
                                                        unsigned long long *__restrict__ out) {
    const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nb = (n + 63) / 64;
    if (g >= nb) return;
    const int lane = threadIdx.x & 63;
    const long long H = occ_packet_header(n);
    unsigned *goff = reinterpret_cast<unsigned *>(out + 1 + 2 * nb);
    const unsigned long long mixed = out[1 + nb + g];
    const long long off = wg_off[g / kOccGroupsPerWg] + (long long)goff[g];
    if (lane == 0) {
        goff[g] = (unsigned)off;
        if (g == 0) out[0] = (unsigned long long)wg_off[nwg];
    }
    if ((mixed >> lane) & 1ull) {
        const long long at = off + __popcll(mixed & ((1ull << lane) - 1ull));
        if (at < cap) out[H + at] = words[g * 64 + lane];
    }
}

// ---- records -> packet, without the plain words in between (round 4) ---------------------------
//
// The hand-off of a rank was pack (records -> its planes' words), classify, scan, write: four
// launches and two passes over the words for what the records say directly.  Here the classify
// kernel reads the records itself (lazy-aware: a settled coarse tile costs no load), the write
// kernel sums the workgroup totals before it in its own prologue (no scan launch) and takes the
// mixed words from the records again (only the mixed lanes read).  When the caller hands in the
// whole grid's plane, the rank's own words are stored at their place there as well, so that the
// expansion on this rank can skip its own packet.  X % 32 == 0 and X * Y % 64 == 0.

// n / d for 32-bit unsigned n by a multiplication (the kernels below divide word indices by the
// row length and the plane height per lane and word: the compiler's 64-bit division sequence was
// most of the first version's 43 us).  d >= 1; exact for every n < 2^32.
struct FastDiv {
    unsigned d, m, s;  // q = (mulhi(m, n) + n) >> s, with the 33-bit sum formed as t + ((n - t) >> 1)
};
__host__ __device__ inline FastDiv fast_div(unsigned d) {
    FastDiv f;
    f.d = d;
    unsigned s = 0;
    while ((1ull << s) < d) ++s;
    f.s = s;
    f.m = (unsigned)((((1ull << s) - d) << 32) / d + 1);
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv f) {
    if (f.s == 0) return n;  // d == 1
    const unsigned t = __umulhi(f.m, n);
    return (t + ((n - t) >> 1)) >> (f.s - 1);
}
struct OccGeom {
    FastDiv wpr, Y;  // 32-bit words per row, rows per plane
    FastDiv P64;     // 64-bit words per plane
};

// 64-bit word i of the rank's planes in local order (planes zl0 .. zl0 + nz - 1); 2 i + 1 < 2^32.
// SEEN: the records' seen bits instead of their occupancy (the host hand-off ships both planes).
template <bool SEEN = false>
__device__ __forceinline__ unsigned long long occ_word_from_rec(const CarveParams &p, const OccGeom &og,
                                                                int zl0, unsigned i) {
    unsigned long long w = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const unsigned j = 2u * i + (unsigned)h;
        const unsigned row = fdiv(j, og.wpr);     // z * Y + y
        const unsigned zi = fdiv(row, og.Y);
        const int k = (int)(j - row * og.wpr.d), y = (int)(row - zi * og.Y.d);
        const int z = zl0 + (int)zi;
        const int r = (z & 7) * 8 + (y & 7);
        const int code = lazy_code(p, k >> 1, y >> 3, z >> 3);
        uint32_t e;
        // (X % 32 == 0: every voxel of a 32-bit word lies inside the grid, where the records'
        // "outside = seen" convention does not reach)
        if (code && SEEN) {
            e = lazy_seen(p, code, k >> 1, y >> 3, z >> 3, (k & 1) * 2, r) |
                (lazy_seen(p, code, k >> 1, y >> 3, z >> 3, (k & 1) * 2 + 1, r) << 16);
        } else if (code) {
            e = lazy_occ(p, code, k >> 1, y >> 3, z >> 3, (k & 1) * 2, r) |
                (lazy_occ(p, code, k >> 1, y >> 3, z >> 3, (k & 1) * 2 + 1, r) << 16);
        } else {
            const uint16_t *rec = p.rec + rec_index(p, k >> 1, y >> 3, z >> 3, (k & 1) * 2) * kRecU16;
            e = (uint32_t)rec[(SEEN ? 64 : 0) + r] | ((uint32_t)rec[kRecU16 + (SEEN ? 64 : 0) + r] << 16);
        }
        w |= (unsigned long long)e << (32 * h);
    }
    return w;
}
// where local word i lies in the whole grid's plane
__device__ __forceinline__ long long occ_global_index(const CarveParams &p, const OccGeom &og, int zl0,
                                                      unsigned i) {
    const unsigned zi = fdiv(i, og.P64);
    return (long long)global_z(p, zl0 + (int)zi) * og.P64.d + (i - zi * og.P64.d);
}

// occ_classify_kernel on records (same packet fields), + the rank's own words into `full`
template <bool SEEN = false>
__global__ __launch_bounds__(256) void occ_pack_classify_kernel(const CarveParams p, const OccGeom og,
                                                                int zl0, long long n,
                                                                unsigned long long *__restrict__ out,
                                                                int *__restrict__ wg_sum,
                                                                unsigned long long *__restrict__ full) {
    __shared__ long long wtot[4];
    const long long nb = (n + 63) / 64;
    const int lane = threadIdx.x & 63;
    const long long g0 =
        (long long)blockIdx.x * kOccGroupsPerWg + (threadIdx.x >> 6) * kOccGroupsPerWave;
    unsigned long long my_ones = 0ull, my_mixed = 0ull;
#pragma unroll
    for (int it = 0; it < kOccGroupsPerWave; ++it) {
        const long long i = (g0 + it) * 64 + lane;
        unsigned long long w = 0ull;  // (a word past n reads as 0: neither all-one nor mixed)
        if (i < n) {
            w = occ_word_from_rec<SEEN>(p, og, zl0, (unsigned)i);
            if (full) __builtin_nontemporal_store(w, full + occ_global_index(p, og, zl0, (unsigned)i));
        }
        const unsigned long long ones = __ballot(w == ~0ull);
        const unsigned long long mixed = __ballot(w != 0ull && w != ~0ull);
        if (lane == it) {
            my_ones = ones;
            my_mixed = mixed;
        }
    }
    const long long g = g0 + lane;
    long long total;
    const long long pre = wg_exclusive_scan((long long)__popcll(my_mixed), wtot, &total);
    if (lane < kOccGroupsPerWave && g < nb) {
        out[1 + g] = my_ones;
        out[1 + nb + g] = my_mixed;
        reinterpret_cast<unsigned *>(out + 1 + 2 * nb)[g] = (unsigned)pre;  // + its workgroup's offset later
    }
    if (threadIdx.x == 0) wg_sum[blockIdx.x] = (int)total;
}

// occ_write_kernel with the workgroup offsets summed here (four groups per workgroup: one
// classify workgroup's) and the mixed words taken from the records
template <bool SEEN = false>
__global__ __launch_bounds__(256) void occ_pack_write_kernel(const CarveParams p, const OccGeom og,
                                                             int zl0, long long n,
                                                             long long cap,
                                                             const int *__restrict__ wg_sum, int nwg,
                                                             unsigned long long *__restrict__ out) {
    __shared__ long long s_part[4];
    __shared__ long long s_off, s_total;
    const long long nb = (n + 63) / 64;
    const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    {   // mixed words before this workgroup's groups: the totals of the classify workgroups before
        const int mine = (int)(((long long)blockIdx.x * 4) / kOccGroupsPerWg);
        const int upto = blockIdx.x == 0 ? nwg : mine;  // (workgroup 0 also leaves the packet's total)
        long long a = 0, b = 0;
        for (int k = threadIdx.x; k < upto; k += 256) {
            const int v = wg_sum[k];
            b += v;
            if (k < mine) a += v;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            a += __shfl_xor(a, d);
            b += __shfl_xor(b, d);
        }
        if (lane == 0) s_part[threadIdx.x >> 6] = a;
        __syncthreads();
        if (threadIdx.x == 0) s_off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        __syncthreads();
        if (lane == 0) s_part[threadIdx.x >> 6] = b;
        __syncthreads();
        if (threadIdx.x == 0) s_total = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = (unsigned long long)s_total;
    if (g >= nb) return;
    const long long H = occ_packet_header(n);
    unsigned *goff = reinterpret_cast<unsigned *>(out + 1 + 2 * nb);
    const unsigned long long mixed = out[1 + nb + g];
    const long long off = s_off + (long long)goff[g];
    // (every lane has read goff[g] before lane 0 overwrites it: the wave runs in lockstep and the
    // store below follows the load in program order)
    const long long i = g * 64 + lane;
    unsigned long long w = 0;
    const bool take = (mixed >> lane) & 1ull;
    if (take) w = occ_word_from_rec<SEEN>(p, og, zl0, (unsigned)i);
    if (lane == 0) goff[g] = (unsigned)off;
    if (take) {
        const long long at = off + __popcll(mixed & ((1ull << lane) - 1ull));
        if (at < cap) out[H + at] = w;
    }
}

// all packets of one all-gather (world x S words) -> the plain words of every OTHER rank's
// slab, at slab q's place q * n in `full`.  A wave rebuilds kExpandChunks x 128 words: lane l
// owns words 2l, 2l+1 of a chunk (one 16-byte store), i.e. bits 2(l%32), 2(l%32)+1 of group
// 2c + l/32.
constexpr int kExpandChunks = 4;

// wpg = 0: contiguous slabs, rank q's words go to q * n, `self` is skipped (its words are
// there already).  wpg > 0: striped slabs -- the planes are cut into groups of 8 (wpg words
// each, even) and rank q owns groups q, q + world, ...: its word i goes to
// ((i / wpg) * world + q) * wpg + i % wpg; self = -1 expands every rank, the caller's included,
// self = the caller's rank skips its packet (its words are in `full` already:
// occ_pack_classify_kernel).  *overflow is raised when ANY packet, the skipped one included,
// announces more mixed words than cap.
__global__ __launch_bounds__(256) void occ_expand_kernel(const unsigned long long *__restrict__ in,
                                                         long long S, int world, int self,
                                                         long long n, long long cap,
                                                         unsigned long long *__restrict__ full,
                                                         int *__restrict__ overflow,
                                                         long long wpg) {
    const long long nb = (n + 63) / 64;
    const long long per = (nb + 2 * kExpandChunks - 1) / (2 * kExpandChunks);  // waves per slab
    const long long wv = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= per * world) return;
    const int q = (int)(wv / per);
    const long long c0 = (wv % per) * kExpandChunks;  // first 128-word chunk
    const int lane = threadIdx.x & 63;
    const unsigned long long *pk = in + (long long)q * S;
    // the caller's own packet is looked at too, although its words are never expanded: every
    // rank must arrive at the same verdict about an exchange, or one of them repairs it (a
    // collective) while the others go on
    if ((long long)pk[0] > cap) {
        if (c0 == 0 && lane == 0) *overflow = 1;
        return;
    }
    if (q == self) return;
    const long long H = occ_packet_header(n);
    const unsigned *goff = reinterpret_cast<const unsigned *>(pk + 1 + 2 * nb);
    unsigned long long *dst = full + (wpg ? 0 : (long long)q * n);
    const bool pair_ok = wpg ? true : (((long long)q * n) & 1) == 0;  // 16-byte aligned pairs
    const int bit = 2 * (lane & 31);
    unsigned long long ones[kExpandChunks], mixed[kExpandChunks];
    long long off[kExpandChunks];
#pragma unroll
    for (int k = 0; k < kExpandChunks; ++k) {
        const long long g = 2 * (c0 + k) + (lane >> 5);
        const bool ok = g < nb;
        ones[k] = ok ? pk[1 + g] : 0ull;
        mixed[k] = ok ? pk[1 + nb + g] : 0ull;
        off[k] = ok ? (long long)goff[g] : 0;
    }
#pragma unroll
    for (int k = 0; k < kExpandChunks; ++k) {
        const long long i = (c0 + k) * 128 + 2 * lane;
        if (i >= n) continue;
        const unsigned m2 = (unsigned)(mixed[k] >> bit) & 3u, o2 = (unsigned)(ones[k] >> bit) & 3u;
        const long long at = H + off[k] + __popcll(mixed[k] & ((1ull << bit) - 1ull));
        unsigned long long w0 = (o2 & 1u) ? ~0ull : 0ull, w1 = (o2 & 2u) ? ~0ull : 0ull;
        if (m2 & 1u) w0 = pk[at];
        if (m2 & 2u) w1 = pk[at + (m2 & 1u)];
        // (striped: i is even and wpg is even, so i and i + 1 lie in the same group)
        const long long at_dst = wpg ? ((i / wpg) * world + q) * wpg + i % wpg : i;
        if (pair_ok && i + 1 < n) {
            // (the merged plane is written once per job and read by nobody on this device while
            // the next carve runs: non-temporal, so that it does not push the views' tables and
            // bit planes out of the L2)
            typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
            u64x2 v;
            v.x = w0;
            v.y = w1;
            __builtin_nontemporal_store(v, reinterpret_cast<u64x2 *>(dst + at_dst));
        } else {
            dst[at_dst] = w0;
            if (i + 1 < n) dst[at_dst + 1] = w1;
        }
    }
}

}  // namespace arvx
