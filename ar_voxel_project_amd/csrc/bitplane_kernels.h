// bitplane_kernels.h -- one bit per voxel: the neighbourhood tests of the colour
// pass (Model::isInner, reference src/Model.h:126-132) and of the closure
// (3x3x3 .. 9x9x9 box, src/Postprocessing3d.cpp:31-48) as word operations, and
// the ordered compaction of the voxels they select.
//
// Layout: a voxel row (fixed y,z) is XW = ceil(X/64) 64-bit words, voxel x at bit
// x%64 of word x/64; bits past X are zero.  Rows follow each other in (z, y)
// order, so ascending word/bit order is ascending flat index x + X*(y + Y*z).
// A 512^3 plane is 16 MiB and stays in L2; the planes are built from the state records
// (state_kernels.h, bitgrid_from_rec_kernel).
//
// A sparse per-voxel list (the colour pass's colours, the closure's filled voxels) is the ordered
// compaction of such a plane: entry k belongs to the k-th set bit.  Next to the plane the
// context keeps one SparseWord per word -- the word's bits and the number of set bits before
// it, 16 bytes, ONE load --, so that the list position of a voxel is rank + popcount(bits below
// it) instead of a binary search (sparse_find).
#pragma once

#include "arvx_device.h"

namespace arvx {

struct BitGrid {
    int X, Y, Z;  // Z = planes held in this bit plane
    int XW;
};

__device__ __forceinline__ unsigned long long bit_word(const unsigned long long *bits,
                                                       const BitGrid &g, int xw, int y, int z) {
    if (xw < 0 || xw >= g.XW || y < 0 || y >= g.Y || z < 0 || z >= g.Z) return 0ull;
    return bits[((size_t)z * g.Y + y) * g.XW + xw];
}

// box dilation (r = radius, 1..4).  x and y in one launch: the rows y-r..y+r, each dilated along x on the way (round 4: one launch and
// one pass over the plane less than bit_dilate_x_kernel + the y pass).  in2 (may be null): a second
// plane OR-ed onto `in` as it is read -- the closure's "occupied" is the colour pass's occupancy
// plane | its never-seen plane once handleUnseen has run --, and merged (with in2): in | in2 written
// out for the kernels that follow.
__global__ __launch_bounds__(256) void bit_dilate_xy_kernel(const unsigned long long *__restrict__ in,
                                                            const unsigned long long *__restrict__ in2,
                                                            const BitGrid g, int r,
                                                            unsigned long long *__restrict__ out,
                                                            unsigned long long *__restrict__ merged) {
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (w >= (size_t)g.XW * g.Y * g.Z) return;
    const int xw = (int)(w % g.XW);
    const size_t row = w / g.XW;
    const int y = (int)(row % g.Y), z = (int)(row / g.Y);
    unsigned long long acc = 0ull;
    for (int dy = -r; dy <= r; ++dy) {
        unsigned long long c = bit_word(in, g, xw, y + dy, z);
        unsigned long long L = bit_word(in, g, xw - 1, y + dy, z), R = bit_word(in, g, xw + 1, y + dy, z);
        if (in2) {
            c |= bit_word(in2, g, xw, y + dy, z);
            L |= bit_word(in2, g, xw - 1, y + dy, z);
            R |= bit_word(in2, g, xw + 1, y + dy, z);
            if (dy == 0) merged[w] = c;
        }
        acc |= c;
        for (int k = 1; k <= r; ++k) acc |= (c << k) | (c >> k) | (L >> (64 - k)) | (R << (64 - k));
    }
    const int nbits = min(64, g.X - xw * 64);  // keep the padding clear
    if (nbits < 64) acc &= (1ull << nbits) - 1ull;
    out[w] = acc;
}

// ---- ordered compaction of the set bits ------------------------------------------------

constexpr int kBitBlock = 4096;  // words per workgroup of the three-launch form (arvx_colors_upload)

__global__ __launch_bounds__(256) void bit_count_kernel(const unsigned long long *__restrict__ bits,
                                                        size_t nwords, int *__restrict__ counts) {
    __shared__ int wsum[4];
    const size_t base = (size_t)blockIdx.x * kBitBlock;
    int mine = 0;
    for (int it = 0; it < kBitBlock / 256; ++it) {
        const size_t w = base + (size_t)it * 256 + threadIdx.x;
        if (w < nwords) mine += __popcll(bits[w]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

struct __attribute__((aligned(16))) SparseWord {
    unsigned long long bits;
    int rank;  // set bits before this word
    int pad;
};

// index[k] = flat index (x + X*row) of the k-th set bit, ascending; words[w] = {bits, set bits
// before word w}.  Either output may be null.
__global__ __launch_bounds__(256) void bit_write_kernel(const unsigned long long *__restrict__ bits,
                                                        size_t nwords, const BitGrid g,
                                                        const long long *__restrict__ offsets,
                                                        int *__restrict__ index,
                                                        SparseWord *__restrict__ words) {
    __shared__ int wtot[4];
    const size_t base = (size_t)blockIdx.x * kBitBlock;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long run = offsets[blockIdx.x];
    for (int it = 0; it < kBitBlock / 256; ++it) {
        const size_t w = base + (size_t)it * 256 + threadIdx.x;
        unsigned long long b = (w < nwords) ? bits[w] : 0ull;
        const int n = __popcll(b);
        int sc = n;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(sc, d);
            if (lane >= d) sc += t;
        }
        if (lane == 63) wtot[wave] = sc;
        __syncthreads();
        long long slot = run + sc - n;
        for (int v = 0; v < wave; ++v) slot += wtot[v];
        if (words && w < nwords) words[w] = SparseWord{b, (int)slot, 0};
        if (b && index) {
            const size_t row = w / g.XW;
            const int x0 = (int)(w % g.XW) * 64;
            const int first = (int)(row * g.X) + x0;
            while (b) {
                index[slot++] = first + (__ffsll((long long)b) - 1);
                b &= b - 1ull;
            }
        }
        run += wtot[0] + wtot[1] + wtot[2] + wtot[3];
        __syncthreads();
    }
}

// ---- scans in ONE launch (round 4) ------------------------------------------------------------
//
// scan_lookback_kernel: a workgroup takes a chunk of kScanChunk entries by TICKET, sums it, publishes
// the sum, finds the sum of the chunks before it by looking back at them (status words: tagged
// 8-byte granules {aggregate or inclusive prefix, tag}; a workgroup waits only for lower tickets,
// i.e. for workgroups that are running), and writes its offsets; the total goes to *total (device)
// and to the page-locked word *total_host, which the host reads at the call's ONE synchronisation.
// (Round 4 compacted the bit planes the same way; round 5 gave the compaction counts from the
// plane's producer instead -- below -- because the tickets paced it.)
// ticket_ctr: 32-bit (a 64-bit returning atomic on one word paced the launch), never reset, wraps;
// ticket_base: the tickets all earlier launches took (modulo 2^32).
constexpr uint32_t kCompactPrefix = 0x80000000u;  // status: the value is an inclusive prefix

// one value per thread across the 256-thread workgroup: returns the exclusive prefix, *total the sum
__device__ __forceinline__ long long wg_scan_exclusive(long long mine, long long *wtot, long long *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long sc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long t = __shfl_up(sc, d);
        if (lane >= d) sc += t;
    }
    if (lane == 63) wtot[wave] = sc;
    __syncthreads();
    long long pre = sc - mine;
    for (int w = 0; w < wave; ++w) pre += wtot[w];
    *total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    return pre;
}

// (scan_lookback_kernel, below.)  The set bits (or counts) before chunk c: wave 0 of the workgroup publishes the chunk's aggregate
// and looks back over the chunks before it, 64 at a time, until it meets one that has published
// its inclusive prefix; then it publishes its own.  Returns the exclusive prefix (in every lane of
// the wave).  A chunk only waits for lower tickets: workgroups that are running.
__device__ __forceinline__ long long lookback_exclusive(unsigned long long *__restrict__ status, int c,
                                                        long long agg, uint32_t tag,
                                                        unsigned *__restrict__ fault, int lane) {
    long long excl = 0;
    if (c > 0) {
        if (lane == 0) granule_store(status + c, (uint32_t)agg, tag);
        // K windows of 64 chunks are requested together and consumed nearest first, until one holds
        // an inclusive prefix.  (K = 8 -- one round trip for 512 chunks, the workgroups of a launch
        // start together -- against K = 1, the classic walk: 36.5 / 24.8 us against 34.8 / 22.0 for the
        // compaction and the 250-chunk scan of EXPERIMENTS.md round 4, no difference at 1024^3: the
        // polls all go to the same few lines, fewer of them is at least not worse.)
#ifndef ARVX_LOOKBACK_WINDOWS
#define ARVX_LOOKBACK_WINDOWS 1
#endif
        constexpr int K = ARVX_LOOKBACK_WINDOWS;
        int look = c - 1;  // lane l of window j looks at chunk look - 64 j - l
        bool done = false;
        for (unsigned spin = 0; !done;) {
            unsigned long long gr[K];
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int k = look - 64 * j - lane;
                gr[j] = ((unsigned long long)tag << 32) | kCompactPrefix;  // (before chunk 0: prefix 0)
                if (k >= 0) gr[j] = granule_load(status + k);
            }
            bool stalled = false;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (done || stalled) continue;
                const bool valid = (uint32_t)(gr[j] >> 32) == tag;
                const bool pref = valid && ((uint32_t)gr[j] & kCompactPrefix);
                const unsigned long long vmask = __ballot(valid), pmask = __ballot(pref);
                const int fp = pmask ? __ffsll((long long)pmask) - 1 : 64;  // nearest chunk with a prefix
                const unsigned long long need = fp < 63 ? ((2ull << fp) - 1ull) : ~0ull;
                if ((vmask & need) != need) {  // a chunk in between has not published yet
                    stalled = true;
                    continue;
                }
                long long v = (lane <= fp) ? (long long)((uint32_t)gr[j] & ~kCompactPrefix) : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
                excl += v;
                if (fp < 64) done = true;
                else look -= 64;
            }
            if (stalled) {
                if (++spin > (1u << 20)) {
                    if (lane == 0 && fault)
                        __hip_atomic_store(fault, 6u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    if (lane == 0) {
        // the granules carry prefixes in 31 bits: a total that does not fit is reported like a
        // give-up (mark 7: the host fails the call at its synchronisation) instead of wrapping
        if ((unsigned long long)(excl + agg) >= (unsigned long long)kCompactPrefix && fault)
            __hip_atomic_store(fault, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        granule_store(status + c, (uint32_t)(excl + agg) | kCompactPrefix, tag);
    }
    return excl;
}

// ---- the compaction without tickets (round 5) ---------------------------------------------------
//
// Every plane that is compacted is PRODUCED by a kernel of the same call (the surface of the
// colour pass, the closure's fill plane): that kernel -- one word per thread, as before -- also
// adds each wave's set bits to the count of its chunk of kBitChunk words.  The compaction then needs
// no look-back, no status words and no ticket: a chunk's workgroup adds up the counts of the chunks
// before it (a few thousand ints from the L2) and goes straight to its words.  (The ticketed form
// was paced by its tickets: a returning atomic on one word is served every ~25 ns, 13 of the kernel's
// 29 us at 512 chunks -- EXPERIMENTS.md rounds 4 and 5.)  The counts live in two buffers used in
// turn: the compaction that reads one zeroes the other for the next producer.
constexpr int kBitChunk = 1024;  // words per workgroup of the compaction (kIt = 4 per thread)

// the sum of one value per thread over the 256-thread workgroup, in every thread
__device__ __forceinline__ long long wg_sum(long long mine, long long *wtot) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d);
    if ((threadIdx.x & 63) == 0) wtot[threadIdx.x >> 6] = mine;
    __syncthreads();
    const long long t = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
    return t;
}
// a workgroup's 256 consecutive words (one per thread, all inside one chunk) counted into their
// chunk: ONE atomic per workgroup (one per wave cost the producers 4 us at 512^3)
__device__ __forceinline__ void chunk_count_add(int *__restrict__ counts, size_t w, int bits_set) {
    __shared__ long long s_cnt[4];
    static_assert(kBitChunk % 256 == 0, "a workgroup's words lie in one chunk");
    const long long t = wg_sum(bits_set, s_cnt);
    if (threadIdx.x == 0 && t) atomicAdd(counts + w / kBitChunk, (int)t);
}

// surf = occupied and not inner (all six neighbours occupied; outside the planes held = empty) for
// the planes [zlo, zhi) of the plane set (local planes of `occ`; the other planes get zeros), and
// counts[chunk] += its set bits.  One word per thread; the grid covers whole waves.
__global__ __launch_bounds__(256) void bit_surface_count_kernel(const unsigned long long *__restrict__ occ,
                                                                const BitGrid g, int zlo, int zhi,
                                                                unsigned long long *__restrict__ out,
                                                                int *__restrict__ counts) {
    const size_t nwords = (size_t)g.XW * g.Y * g.Z;
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long s = 0ull;
    if (w < nwords) {
        const int xw = (int)(w % g.XW);
        const size_t row = w / g.XW;
        const int y = (int)(row % g.Y), z = (int)(row / g.Y);
        if (z >= zlo && z < zhi) {
            const unsigned long long c = occ[w];
            if (c) {
                unsigned long long inner = c;
                inner &= (c << 1) | (bit_word(occ, g, xw - 1, y, z) >> 63);
                inner &= (c >> 1) | (bit_word(occ, g, xw + 1, y, z) << 63);
                inner &= bit_word(occ, g, xw, y - 1, z) & bit_word(occ, g, xw, y + 1, z);
                inner &= bit_word(occ, g, xw, y, z - 1) & bit_word(occ, g, xw, y, z + 1);
                s = c & ~inner;
            }
        }
        out[w] = s;
    }
    chunk_count_add(counts, w, __popcll(s));
}

// out = (OR of the planes z-r..z+r of `in`) & ~minus for the planes [zlo, zhi), zeros elsewhere;
// counts[chunk] += its set bits (the closure's fill plane: bit_dilate_xy_kernel's output dilated
// along z, without the voxels that are occupied already)
// mark (may be null): the per-coarse-tile codes of the lazy state (arvx_device.h); a coded tile that
// receives a bit here gets bit 7 of its code set (every writer stores the same byte), for
// rec_or_bitgrid_lazy_kernel.  A coarse tile is 64 voxels in x -- one word -- by 8 << cyShift rows by
// 8 << czShift planes, coarseX x coarseY tiles per plane of tiles.
struct CoarseMark {
    uint8_t *code;
    int coarseX, coarseY, cyShift, czShift;
};
__global__ __launch_bounds__(256) void bit_dilate_z_count_kernel(const unsigned long long *__restrict__ in,
                                                                 const BitGrid g, int r,
                                                                 const unsigned long long *__restrict__ minus,
                                                                 int zlo, int zhi,
                                                                 unsigned long long *__restrict__ out,
                                                                 int *__restrict__ counts,
                                                                 const CoarseMark mark) {
    const size_t nwords = (size_t)g.XW * g.Y * g.Z;
    const size_t plane = (size_t)g.XW * g.Y;
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;
    unsigned long long acc = 0ull;
    if (w < nwords) {
        const int z = (int)(w / plane);
        if (z >= zlo && z < zhi) {
            for (int k = -r; k <= r; ++k)
                if (z + k >= 0 && z + k < g.Z) acc |= in[(size_t)((long long)w + (long long)k * (long long)plane)];
            acc &= ~minus[w];
        }
        out[w] = acc;
        if (acc && mark.code) {
            const int xw = (int)(w % g.XW), y = (int)((w / g.XW) % g.Y);
            uint8_t *c = mark.code + xw + mark.coarseX * ((y >> (3 + mark.cyShift)) +
                                                         mark.coarseY * (z >> (3 + mark.czShift)));
            const uint8_t v = *c;
            if (v && !(v & 0x80)) *c = (uint8_t)(v | 0x80);
        }
    }
    chunk_count_add(counts, w, __popcll(acc));
}

// The ordered compaction of a plane whose chunk counts are known (one workgroup per chunk, in any
// order): index[k] = flat index of the k-th set bit for k < cap, words[w] = {bits, set bits before
// word w}; the list's length goes to *total and the page-locked *total_host.  zero_next: the other
// count buffer, cleared for the next producer.
__global__ __launch_bounds__(256) void bit_compact_counted_kernel(
    const unsigned long long *__restrict__ bits, size_t nwords, const BitGrid g,
    const int *__restrict__ counts, int *__restrict__ zero_next, long long cap, int *__restrict__ index,
    SparseWord *__restrict__ words, long long *__restrict__ total, long long *__restrict__ total_host) {
    constexpr int kIt = kBitChunk / 256;  // words per thread: word it * 256 + thread (whole lines per wave)
    __shared__ long long wtot[4];
    __shared__ int s_itw[kIt * 4];  // set bits of iteration it in wave w, at it * 4 + w
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x;
    const size_t base = (size_t)c * kBitChunk;
    unsigned long long b[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const size_t w = base + (size_t)it * 256 + threadIdx.x;
        b[it] = (w < nwords) ? bits[w] : 0ull;
    }
    // the set bits before the chunk: the counts of the chunks before it
    long long mine = 0;
    for (int i = threadIdx.x; i < c; i += 256) mine += counts[i];
    if (threadIdx.x == 0) zero_next[c] = 0;
    // a word's rank inside the chunk: the bits of the earlier iterations + those of the lower waves
    // in its iteration + the wave's scan (the words of one iteration are 256 consecutive ones)
    int before[kIt];
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
        const int n = __popcll(b[it]);
        int sc = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(sc, d);
            if (lane >= d) sc += t;
        }
        before[it] = sc - n;
        if (lane == 63) s_itw[it * 4 + wave] = sc;
    }
    const long long excl = wg_sum(mine, wtot);  // (its barriers also publish s_itw)
    if (c == (int)gridDim.x - 1 && threadIdx.x == 0) {
        const long long all = excl + counts[c];
        *total = all;
        if (total_host) __hip_atomic_store(total_host, all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // The list entries: words with more than kLightBits entries (faces of the model that run along
    // x) are written by the whole wave -- lane j takes bit j, the entries leave as one run --, the
    // others bit by bit by their own lane.
    constexpr int kLightBits = 4;
    int pre = 0;  // set bits of the chunk before (iteration it, this wave)
#pragma unroll
    for (int it = 0; it < kIt; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < wave) pre += s_itw[it * 4 + q];
        const size_t w = base + (size_t)it * 256 + threadIdx.x;
        unsigned long long bw = b[it];
        int slot = (int)(excl + pre + before[it]);  // (< 2^31)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q >= wave) pre += s_itw[it * 4 + q];
        if (words && w < nwords) words[w] = SparseWord{bw, slot, 0};
        if (!index) continue;
        const size_t row = w / g.XW;
        const int first = (int)(row * g.X) + (int)(w % g.XW) * 64;
        const bool heavy = __popcll(bw) > kLightBits;
        unsigned long long hm = __ballot(heavy);
        while (hm) {
            const int L = __ffsll((long long)hm) - 1;
            hm &= hm - 1ull;
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)bw, L),
                           hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(bw >> 32), L);
            const int s0 = __builtin_amdgcn_readlane(slot, L), f0 = __builtin_amdgcn_readlane(first, L);
            const bool set = ((lane < 32 ? lo >> lane : hi >> (lane - 32)) & 1u) != 0u;
            const int at = s0 + (int)__builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
            if (set && at < cap) index[at] = f0 + lane;
        }
        if (!heavy) {
            while (bw) {
                if (slot < cap) index[slot] = first + (__ffsll((long long)bw) - 1);
                ++slot;
                bw &= bw - 1ull;
            }
        }
    }
}

// The scans in ONE launch, the same way: a chunk of kScanChunk entries per workgroup and ticket.
constexpr int kScanChunk = 4096;  // (8192: the triangle scan 19.8 -> 21.3 us, the column scan 10.1 -> 14.2: EXPERIMENTS.md round 5)

// Exclusive scan of one small count per entry (counts[i] >= 0): offsets[i] (32 bits); the sum goes to
// *total and *total_host.  kLookup: the count of entry i is table[keys[4 i + 3] & 255] (the triangles of
// marching-cubes cell i from its cube index: `keys` is the cell list, `table` 256 bytes, copied to
// LDS first -- read from constant memory with a different index in every lane it was a second round
// trip per entry), and entries at or behind *n_dev count nothing.
template <bool kLookup>
__global__ __launch_bounds__(256) void scan_lookback_kernel(
    const int *__restrict__ counts, const int8_t *__restrict__ table, long long n,
    const long long *__restrict__ n_dev, int *__restrict__ offsets,
    unsigned *__restrict__ ticket_ctr, unsigned ticket_base,
    unsigned long long *__restrict__ status, uint32_t tag, long long *__restrict__ total,
    long long *__restrict__ total_host, unsigned *__restrict__ fault) {
    __shared__ long long wtot[4];
    __shared__ int s_chunk;
    __shared__ long long s_excl;
    __shared__ int s_val[kScanChunk];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_chunk = (int)(atomicAdd(ticket_ctr, 1u) - ticket_base);
    __syncthreads();
    __shared__ int8_t s_table[256];
    if (kLookup) s_table[threadIdx.x] = table[threadIdx.x];
    const int c = s_chunk;
    const int nchunks = (int)((n + kScanChunk - 1) / kScanChunk);
    const long long n_eff = (kLookup && n_dev) ? (*n_dev < n ? *n_dev : n) : n;
    // the chunk's counts: read with neighbouring threads at neighbouring entries (whole lines),
    // handed to the threads that scan 16 consecutive ones each through LDS
    if (kLookup) {
        int key[kScanChunk / 256];
#pragma unroll
        for (int it = 0; it < kScanChunk / 256; ++it) {
            const long long i = (long long)c * kScanChunk + it * 256 + threadIdx.x;
            key[it] = (i < n_eff) ? (counts[4 * i + 3] & 255) : -1;
        }
        __syncthreads();  // (the table)
#pragma unroll
        for (int it = 0; it < kScanChunk / 256; ++it)
            s_val[it * 256 + threadIdx.x] = key[it] >= 0 ? (int)s_table[key[it]] : 0;
    } else {
#pragma unroll
        for (int it = 0; it < kScanChunk / 256; ++it) {
            const long long i = (long long)c * kScanChunk + it * 256 + threadIdx.x;
            s_val[it * 256 + threadIdx.x] = (i < n) ? counts[i] : 0;
        }
    }
    __syncthreads();
    int v[kScanChunk / 256];
    long long mine = 0;
#pragma unroll
    for (int k = 0; k < kScanChunk / 256; ++k) {
        v[k] = s_val[threadIdx.x * (kScanChunk / 256) + k];
        mine += v[k];
    }
    long long agg;
    long long pre = wg_scan_exclusive(mine, wtot, &agg);
    if (wave == 0) {
        const long long excl = lookback_exclusive(status, c, agg, tag, fault, lane);
        if (lane == 0) {
            s_excl = excl;
            if (c == nchunks - 1) {
                *total = excl + agg;
                if (total_host)
                    __hip_atomic_store(total_host, excl + agg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    pre += s_excl;
    // (the sums fit 31 bits: the status words carry them) back through LDS, so that the offsets
    // leave as whole lines too
#pragma unroll
    for (int k = 0; k < kScanChunk / 256; ++k) {
        s_val[threadIdx.x * (kScanChunk / 256) + k] = (int)pre;
        pre += v[k];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kScanChunk / 256; ++it) {
        const long long i = (long long)c * kScanChunk + it * 256 + threadIdx.x;
        if (i < n_eff) offsets[i] = s_val[it * 256 + threadIdx.x];
    }
}

// the plane of a list that came from the host (arvx_colors_upload): bit index[k] set
__global__ __launch_bounds__(256) void bits_from_index_kernel(const int *__restrict__ index,
                                                              long long n, const BitGrid g,
                                                              unsigned long long *__restrict__ bits) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int i = index[k];
    const int x = i % g.X;
    const size_t row = (size_t)(i / g.X);
    atomicOr(bits + row * g.XW + (x >> 6), 1ull << (x & 63));
}

// a sparse list's index (null: empty list)
struct SparseList {
    const SparseWord *w;
};
// position of bit sh of a word in the list, or -1
__device__ __forceinline__ int sparse_rank(const SparseWord &e, int sh) {
    if (!((e.bits >> sh) & 1ull)) return -1;
    return e.rank + __popcll(e.bits & ((1ull << sh) - 1ull));
}
__device__ __forceinline__ SparseWord sparse_word(const SparseList &l, size_t at) {
    const uint4 q = *reinterpret_cast<const uint4 *>(l.w + at);  // one 16-byte load
    SparseWord e;
    e.bits = (unsigned long long)q.x | ((unsigned long long)q.y << 32);
    e.rank = (int)q.z;
    e.pad = 0;
    return e;
}
// position of voxel (x, row) in the list, or -1
__device__ __forceinline__ int sparse_find(const SparseList &l, int XW, int x, size_t row) {
    if (!l.w) return -1;
    return sparse_rank(sparse_word(l, row * XW + (x >> 6)), x & 63);
}

}  // namespace arvx
