// arvx_capi.hip -- host side of libarvx.so: the C-ABI declared in
// include/arvx/arvx.h over the gfx950 kernels.  No CPU compute path exists
// here: every entry point that does work launches HIP kernels and fails with
// ARVX_ERR_HIP when no device is usable.
#include "arvx/arvx.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "arvx_ctx.h"
#include "carve_kernels.h"
#ifdef ARVX_EXPERIMENTS
#include "carve_stream_kernels.h"  // the one-launch carve: loses by 11-24 %, experiment builds only
#endif
#include "views_kernels.h"
#include "state_kernels.h"
#include "undistort_kernels.h"
#include "color_kernels.h"
#include "closure_kernels.h"
#include "bitplane_kernels.h"
#include "fast_carve_kernels.h"
#include "mc_kernels.h"
#include "mc_mesh_kernels.h"
#include "exchange_kernels.h"
#include <algorithm>

namespace {

thread_local std::string g_last_error;
// the grouping new contexts start with (arvx_set_projection_assoc)
std::atomic<int> g_default_assoc{ARVX_ASSOC_LEFT};

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

}  // namespace

namespace arvx {
int fail_hip(hipError_t e, const char *what, const char *file, int line) {
    return fail(ARVX_ERR_HIP, "%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
}
int fail_msg(int code, const char *msg) { return fail(code, "%s", msg); }
}  // namespace arvx

using arvx::Ctx;

#define ARVX_CHECK_CTX(ctx)                                          \
    do {                                                             \
        if (!(ctx)) return fail(ARVX_ERR_INVALID, "null context");   \
        ARVX_HIP(hipSetDevice((ctx)->device));                       \
    } while (0)

// The entry points of the occupancy hand-off (pack / compress / expand) launch on the
// context's exchange stream when the caller has set one: everything they launch inside goes
// there, and the caller orders the two streams with events.
struct ExchangeStreamScope {
    Ctx *ctx;
    hipStream_t saved;
    explicit ExchangeStreamScope(Ctx *c) : ctx(c), saved(c->stream) {
        if (c->xstream) c->stream = c->xstream;
    }
    ~ExchangeStreamScope() { ctx->stream = saved; }
};

// ---- streams ----------------------------------------------------------------------------
// hipStreamCreate / hipStreamDestroy cost 1.4 - 2.2 ms each on this stack (rocprofv3
// --hip-trace of tools/cpp/arvx_bench6: more than the whole carve of its largest model), and a
// drop-in caller makes a context per Model (src/main.cpp:306-440 builds eight in a row): the
// streams of destroyed contexts are kept per device and handed to the next context.
// A/B switches of the launch path.  The shipped library reads NO environment variable: its
// behaviour never depends on the caller's environment.  An experiment build
// (make EXTRA=-DARVX_EXPERIMENTS, e.g. into ab_libs/ for tools/carve_ab.py) turns the same
// names into environment switches again.
#ifdef ARVX_EXPERIMENTS
static bool experiment_flag(const char *name) { return getenv(name) != nullptr; }
static int experiment_int(const char *name) { return getenv(name) ? atoi(getenv(name)) : 0; }
#else
static constexpr bool experiment_flag(const char *) { return false; }
static constexpr int experiment_int(const char *) { return 0; }
#endif

namespace {
std::mutex g_stream_mutex;
std::vector<std::pair<int, hipStream_t>> g_idle_streams;  // (device, stream), idle and drained

hipError_t acquire_stream(int device, hipStream_t *out) {
    static const bool no_reuse = experiment_flag("ARVX_NO_STREAM_REUSE");
    if (!no_reuse) {
        std::lock_guard<std::mutex> lock(g_stream_mutex);
        for (size_t i = 0; i < g_idle_streams.size(); ++i)
            if (g_idle_streams[i].first == device) {
                *out = g_idle_streams[i].second;
                g_idle_streams.erase(g_idle_streams.begin() + i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}

void release_stream(int device, hipStream_t s) {  // s: synchronised by the caller
    static const bool no_reuse = experiment_flag("ARVX_NO_STREAM_REUSE");
    std::lock_guard<std::mutex> lock(g_stream_mutex);
    if (!no_reuse && g_idle_streams.size() < 64) {
        g_idle_streams.emplace_back(device, s);
        return;
    }
    (void)hipStreamDestroy(s);
}
}  // namespace

// ---- bit-plane helpers (bitplane_kernels.h) ----------------------------------------------

// After a synchronisation of ctx->stream: did a kernel that waits for other workgroups give up
// (arvx_ctx.h, h_fault)?  The results of that launch are then undefined.
static int check_fault(Ctx *ctx) {
    if (!ctx->h_fault || !*ctx->h_fault) return ARVX_OK;
    const unsigned what = *ctx->h_fault;
    *ctx->h_fault = 0u;
    ctx->vstrip_key = 0;   // the ticket counters no longer match the launches: start over
    ctx->carve_layout = 0;
    ctx->stream_layout = 0;
    // (the scans' tickets and status granules too: a chunk that gave up published a prefix under a
    // valid tag -- the control block is zeroed again at its next use)
    ctx->pool_compact.release();
    ctx->compact_tickets = 0;
    ctx->compact_epoch = 0;
    ctx->counts_clean[0] = ctx->counts_clean[1] = false;
    ctx->packets_valid = false;
    return fail(ARVX_ERR_HIP, "a kernel gave up waiting for another workgroup (mark %u): the "
                "results of the calls since the last synchronisation are undefined", what);
}

#define ARVX_SYNC(ctx)                                        \
    do {                                                      \
        ARVX_HIP(hipStreamSynchronize((ctx)->stream));        \
        if (int arvx_f_ = check_fault(ctx)) return arvx_f_;   \
    } while (0)

static int ensure_scratch(Ctx *ctx, size_t need) {
    if (ctx->scratch_bytes >= need) return ARVX_OK;
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr;
    ctx->scratch_bytes = 0;
    ARVX_HIP(hipMalloc(&ctx->d_scratch, need));
    ctx->scratch_bytes = need;
    return ARVX_OK;
}

static int need_rec(Ctx *ctx, bool lazy_ok = false);
static void carve_geometry(const Ctx *ctx, arvx::CarveParams &p);

// the paint plane where it takes part (null: nobody is painted)
static const unsigned long long *paint_plane(const Ctx *ctx) {
    return ctx->paint_valid ? (const unsigned long long *)ctx->pool_paint.p : nullptr;
}

// The context's planes ze0 .. ze0 + g.Z - 1 as bit planes (bitplane_kernels.h) from the
// records: `occ` = occupied, or what the closure calls occupied; `unseen` (may be null) = the
// voxels whose colour is UNSEEN_COLOR.
static int launch_bit_pack(Ctx *ctx, const arvx::BitGrid &g, int closure_occupied, int apply_unseen,
                           unsigned long long *occ, unsigned long long *unseen) {
    if (int rc = need_rec(ctx, true)) return rc;  // (bitgrid_from_rec_kernel reads lazy tiles)
    const size_t nwords = (size_t)g.XW * g.Y * g.Z;
    arvx::CarveParams p;
    carve_geometry(ctx, p);
    p.rec = ctx->d_rec;
    hipLaunchKernelGGL(arvx::bitgrid_from_rec_kernel, dim3((unsigned)((nwords + 255) / 256)),
                       dim3(256), 0, ctx->stream, p, 0, g.Z, closure_occupied, apply_unseen,
                       closure_occupied ? paint_plane(ctx) : nullptr, occ, unseen);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

// counts the set bits per block and scans the counts; *total = number of set bits
static int bit_compact_count(Ctx *ctx, const unsigned long long *bits, size_t nwords, int *d_cnt,
                             long long *d_off, long long *total) {
    const int nblk = (int)((nwords + arvx::kBitBlock - 1) / arvx::kBitBlock);
    hipLaunchKernelGGL(arvx::bit_count_kernel, dim3(nblk), dim3(256), 0, ctx->stream, bits, nwords,
                       d_cnt);
    ARVX_HIP(hipGetLastError());
    hipLaunchKernelGGL(arvx::surface_scan_kernel, dim3(1), dim3(256), 0, ctx->stream, d_cnt, nblk,
                       d_off);
    ARVX_HIP(hipGetLastError());
    ARVX_HIP(hipMemcpyAsync(total, d_off + nblk, sizeof *total, hipMemcpyDeviceToHost,
                            ctx->stream));
    ARVX_SYNC(ctx);
    return ARVX_OK;
}

// the control block of the one-launch compaction / scan kernels: ticket counter, device totals,
// `nchunks` status words; *status: where they start
static int compact_control(Ctx *ctx, size_t nchunks, uint8_t **base, unsigned long long **status) {
    const size_t off_status = 128;  // [0] ticket counter, [8 + 8 k] device totals, then the status words
    // (room for the longest array a context ever compacts or scans -- one entry per voxel, a chunk per
    // 4096 of them --, so that the block never moves between the launches of a call: the kernels of
    // a call read each other's totals from it)
    const size_t most = ctx->nvox_ext / 4096 + 4096;
    const size_t need = off_status + std::max(nchunks, most) * sizeof(unsigned long long) + 64;
    if (ctx->pool_compact.cap < need) {
        ARVX_HIP(ctx->pool_compact.reserve(need));
        ARVX_HIP(hipMemsetAsync(ctx->pool_compact.p, 0, ctx->pool_compact.cap, ctx->stream));
        ctx->compact_tickets = 0;
        ctx->compact_epoch = 0;
    }
    if (++ctx->compact_epoch == 0u) {  // (2^32 launches: start the tags over on clean memory)
        ARVX_HIP(hipMemsetAsync(ctx->pool_compact.p, 0, ctx->pool_compact.cap, ctx->stream));
        ctx->compact_tickets = 0;
        ctx->compact_epoch = 1u;
    }
    *base = (uint8_t *)ctx->pool_compact.p;
    *status = (unsigned long long *)(*base + off_status);
    return ARVX_OK;
}

// Exclusive scan of n counts in ONE launch (scan_lookback_kernel): offsets[i]; the sum goes to device
// word *d_total_out and to ctx->h_totals[slot].  cells: the counts are the triangles of the
// marching-cubes cells of that list (n: its capacity, *n_dev: its length), not an array.
static int scan_counts(Ctx *ctx, const int *counts, const int4 *cells, long long n, const long long *n_dev,
                       int *d_offsets, int slot, const long long **d_total_out) {
    const size_t nchunks = (size_t)((n + arvx::kScanChunk - 1) / arvx::kScanChunk);
    uint8_t *base = nullptr;
    unsigned long long *status = nullptr;
    if (int rc = compact_control(ctx, nchunks, &base, &status)) return rc;
    long long *d_total = (long long *)(base + 8 + 8 * slot);
    ctx->h_totals[slot] = -1;
    if (cells) {
        const int8_t *table = nullptr;  // (the triangle counts of Bourke's table, on the device)
        ARVX_HIP(hipGetSymbolAddress((void **)&table, HIP_SYMBOL(arvx::kMcTri)));
        table += offsetof(arvx::McTriTable, n);
        hipLaunchKernelGGL(arvx::scan_lookback_kernel<true>, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream,
                           (const int *)cells, table, n, n_dev, d_offsets, (unsigned *)base,
                           (unsigned)ctx->compact_tickets, status, ctx->compact_epoch, d_total,
                           ctx->d_totals_host + slot, ctx->d_fault);
    } else {
        hipLaunchKernelGGL(arvx::scan_lookback_kernel<false>, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream,
                           counts, (const int8_t *)nullptr, n, (const long long *)nullptr, d_offsets,
                           (unsigned *)base, (unsigned)ctx->compact_tickets, status, ctx->compact_epoch,
                           d_total, ctx->d_totals_host + slot, ctx->d_fault);
    }
    ARVX_HIP(hipGetLastError());
    ctx->compact_tickets += nchunks;
    if (d_total_out) *d_total_out = d_total;
    return ARVX_OK;
}

// The ordered compaction in ONE launch (bit_compact_counted_kernel) of a plane whose producer left the
// set bits of every chunk of kBitChunk words in `counts`: up to `cap` entries of the list go to d_index,
// every word's SparseWord to d_words; the list's true length is left in device word *d_total_out (for
// the kernels that follow) and in the page-locked ctx->h_totals[slot], which the caller reads after
// its synchronisation.
static int bit_compact(Ctx *ctx, const unsigned long long *bits, size_t nwords, const arvx::BitGrid &g,
                       long long cap, int *d_index, arvx::SparseWord *d_words, int slot,
                       const long long **d_total_out) {
    const size_t nchunks = (nwords + arvx::kBitChunk - 1) / arvx::kBitChunk;
    uint8_t *base = nullptr;
    unsigned long long *status = nullptr;
    if (int rc = compact_control(ctx, 0, &base, &status)) return rc;  // (the device totals live there)
    long long *d_total = (long long *)(base + 8 + 8 * slot);
    ctx->h_totals[slot] = -1;
    // the counts the producer has just left in buffer `counts_cur`; the other buffer is cleared for
    // the next producer on the way
    int *cur = (int *)ctx->pool_chunk_counts.p + (size_t)ctx->counts_cur * ctx->counts_stride;
    int *other = (int *)ctx->pool_chunk_counts.p + (size_t)(ctx->counts_cur ^ 1) * ctx->counts_stride;
    hipLaunchKernelGGL(arvx::bit_compact_counted_kernel, dim3((unsigned)nchunks), dim3(256), 0, ctx->stream,
                       bits, nwords, g, (const int *)cur, other, cap, d_index, d_words, d_total,
                       ctx->d_totals_host + slot);
    ARVX_HIP(hipGetLastError());
    ctx->counts_clean[ctx->counts_cur ^ 1] = true;
    if (d_total_out) *d_total_out = d_total;
    return ARVX_OK;
}
// The count buffer the NEXT plane's producer adds into (zeroed): the two buffers take turns -- the
// compaction that reads one clears the other --, so a call costs no fill launch of its own.
static int chunk_counts(Ctx *ctx, size_t nwords, int **counts) {
    const size_t n = (nwords + arvx::kBitChunk - 1) / arvx::kBitChunk;
    if (ctx->counts_stride < n || !ctx->pool_chunk_counts.p) {
        ARVX_HIP(ctx->pool_chunk_counts.reserve(2 * n * sizeof(int)));
        ctx->counts_stride = n;
        ctx->counts_clean[0] = ctx->counts_clean[1] = false;
    }
    const int k = ctx->counts_cur ^ 1;
    int *buf = (int *)ctx->pool_chunk_counts.p + (size_t)k * ctx->counts_stride;
    if (!ctx->counts_clean[k])  // (first use, or a call that failed between producer and compaction)
        ARVX_HIP(hipMemsetAsync(buf, 0, ctx->counts_stride * sizeof(int), ctx->stream));
    ctx->counts_clean[k] = false;  // the producer is about to add into it
    ctx->counts_cur = k;
    *counts = buf;
    return ARVX_OK;
}

// The total a one-launch compaction / scan left for the host (after the call's synchronisation).  The
// kernel stores it into the page-locked word; should the host not see that store, the device's own
// copy of the word is fetched instead (-1: neither holds a count).
static long long host_total(Ctx *ctx, int slot) {
    const long long t = ctx->h_totals[slot];
    if (t >= 0 || !ctx->pool_compact.p) return t;
    // never expected since the word is allocated coherent (EXPERIMENTS.md round 4): counted, so that
    // a recurrence shows (arvx_get_stats: host_total_fallbacks; the list tests assert 0)
    ++ctx->host_total_fallbacks;
    long long dev = -1;
    if (hipMemcpy(&dev, (const uint8_t *)ctx->pool_compact.p + 8 + 8 * slot, sizeof dev,
                  hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return dev;
}

static int bit_compact_write(Ctx *ctx, const unsigned long long *bits, size_t nwords,
                             const arvx::BitGrid &g, const long long *d_off, int *d_index,
                             arvx::SparseWord *d_words) {
    const int nblk = (int)((nwords + arvx::kBitBlock - 1) / arvx::kBitBlock);
    hipLaunchKernelGGL(arvx::bit_write_kernel, dim3(nblk), dim3(256), 0, ctx->stream, bits, nwords,
                       g, d_off, d_index, d_words);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

// xyz -> device, kernel, results back: the two projection self-tests share this frame
template <class Launch>
static int selftest_xyz(Ctx *ctx, int64_t n, const int32_t *xyz, size_t out_floats, float *out,
                        Launch launch) {
    if (n < 1 || !xyz || !out) return fail(ARVX_ERR_INVALID, "bad argument");
    const size_t in_bytes = (size_t)n * 3 * sizeof(int32_t);
    if (int rc = ensure_scratch(ctx, in_bytes + out_floats * sizeof(float) + 64)) return rc;
    int *d_xyz = (int *)ctx->d_scratch;
    float *d_out = (float *)((uint8_t *)ctx->d_scratch + ((in_bytes + 15) & ~(size_t)15));
    ARVX_HIP(hipMemcpyAsync(d_xyz, xyz, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    launch(d_xyz, d_out);
    ARVX_HIP(hipGetLastError());
    ARVX_HIP(hipMemcpyAsync(out, d_out, out_floats * sizeof(float), hipMemcpyDeviceToHost,
                            ctx->stream));
    ARVX_SYNC(ctx);
    return ARVX_OK;
}

extern "C" {

int arvx_version(void) { return ARVX_VERSION; }

const char *arvx_last_error(void) { return g_last_error.c_str(); }

int arvx_projection_assoc(void) { return g_default_assoc.load(); }

int arvx_set_projection_assoc(int assoc) {
    if (assoc != ARVX_ASSOC_RIGHT && assoc != ARVX_ASSOC_LEFT)
        return fail(ARVX_ERR_INVALID, "grouping %d (ARVX_ASSOC_RIGHT or ARVX_ASSOC_LEFT)", assoc);
    g_default_assoc.store(assoc);
    return ARVX_OK;
}

int arvx_ctx_set_projection_assoc(arvx_ctx *ctx, int assoc) {
    if (!ctx) return fail(ARVX_ERR_INVALID, "null context");
    if (assoc != ARVX_ASSOC_RIGHT && assoc != ARVX_ASSOC_LEFT)
        return fail(ARVX_ERR_INVALID, "grouping %d (ARVX_ASSOC_RIGHT or ARVX_ASSOC_LEFT)", assoc);
    if (ctx->assoc != assoc) ctx->color_ready = false;  // (colours were voted with the other one)
    ctx->assoc = assoc;
    return ARVX_OK;
}

int arvx_ctx_projection_assoc(const arvx_ctx *ctx, int *assoc) {
    if (!ctx || !assoc) return fail(ARVX_ERR_INVALID, "null argument");
    *assoc = ctx->assoc;
    return ARVX_OK;
}

int arvx_device_count(int *count) {
    if (!count) return fail(ARVX_ERR_INVALID, "null count");
    *count = 0;
    ARVX_HIP(hipGetDeviceCount(count));
    return ARVX_OK;
}

int arvx_compose_projection(const float K[9], const float Rt[12], float M[12]) {
    if (!K || !Rt || !M) return fail(ARVX_ERR_INVALID, "null matrix");
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) {
            // cv::gemm small-matrix branch: float, left to right, unfused
            float t = K[3 * r] * Rt[c];
            t = t + K[3 * r + 1] * Rt[4 + c];
            t = t + K[3 * r + 2] * Rt[8 + c];
            M[4 * r + c] = t;
        }
    return ARVX_OK;
}

int arvx_ctx_create_slab(arvx_ctx **out, int device, int X, int Y, int Z, float voxel_size,
                         int z_begin, int z_end) {
    return arvx_ctx_create_slab_halo(out, device, X, Y, Z, voxel_size, z_begin, z_end, 1);
}

int arvx_ctx_create_slab_halo(arvx_ctx **out, int device, int X, int Y, int Z, float voxel_size,
                              int z_begin, int z_end, int halo) {
    if (!out) return fail(ARVX_ERR_INVALID, "null out");
    if (halo < 1 || halo > 16) return fail(ARVX_ERR_INVALID, "halo %d (1..16 planes)", halo);
    *out = nullptr;
    if (X < 1 || Y < 1 || Z < 1) return fail(ARVX_ERR_INVALID, "grid dims must be >= 1");
    if ((int64_t)X * Y * Z > (int64_t)INT32_MAX)
        return fail(ARVX_ERR_INVALID, "X*Y*Z exceeds INT_MAX (Model::flatten returns int)");
    if (!(voxel_size > 0.f)) return fail(ARVX_ERR_INVALID, "voxel size must be > 0");
    if (z_begin < 0 || z_end > Z || z_begin >= z_end)
        return fail(ARVX_ERR_INVALID, "bad slab [%d,%d) of Z=%d", z_begin, z_end, Z);
    int ndev = 0;
    ARVX_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(ARVX_ERR_INVALID, "device %d of %d", device, ndev);
    ARVX_HIP(hipSetDevice(device));
    arvx_ctx *c = new (std::nothrow) arvx_ctx();
    if (!c) return fail(ARVX_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->X = X;
    c->Y = Y;
    c->Z = Z;
    c->z0 = z_begin;
    c->z1 = z_end;
    c->s = voxel_size;
    c->assoc = g_default_assoc.load();
    c->halo = halo;
    c->ze0 = z_begin - halo > 0 ? z_begin - halo : 0;
    c->ze1 = z_end + halo < Z ? z_end + halo : Z;
    c->nvox = (size_t)X * Y * (size_t)(z_end - z_begin);
    c->nvox_ext = (size_t)X * Y * (size_t)(c->ze1 - c->ze0);
    hipError_t e = acquire_stream(device, &c->own_stream);
    if (e != hipSuccess) {
        delete c;
        return arvx::fail_hip(e, "hipStreamCreate", __FILE__, __LINE__);
    }
    c->stream = c->own_stream;
    e = hipMalloc(&c->d_stats, 16 * sizeof(unsigned long long));  // 8 counters + flags
    if (e != hipSuccess) {
        arvx_ctx_destroy(c);
        return arvx::fail_hip(e, "hipMalloc(stats)", __FILE__, __LINE__);
    }
    // Coherent: without the flag a mapped allocation is non-coherent host memory -- the device's stores
    // to it need not be seen by a host that has the line in its cache (the totals' words are reset by
    // the host before every launch: a call then read its own -1 back, once in a few hundred calls)
    e = hipHostMalloc((void **)&c->h_fault, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) {
        memset(c->h_fault, 0, 64);
        e = hipHostGetDevicePointer((void **)&c->d_fault, c->h_fault, 0);
        c->h_totals = (long long *)(c->h_fault + 2);  // six 8-byte words behind the fault word
        c->d_totals_host = (long long *)(c->d_fault + 2);
    }
    if (e != hipSuccess) {
        arvx_ctx_destroy(c);
        return arvx::fail_hip(e, "hipHostMalloc(fault word)", __FILE__, __LINE__);
    }
    c->fresh_pending = true;  // a fresh Model exists only as this flag: see need_rec
    e = hipMemsetAsync(c->d_stats, 0, 64, c->stream);
    if (e != hipSuccess) {
        arvx_ctx_destroy(c);
        return arvx::fail_hip(e, "hipMemsetAsync", __FILE__, __LINE__);
    }
    *out = c;
    return ARVX_OK;
}

int arvx_ctx_create_striped(arvx_ctx **out, int device, int X, int Y, int Z, float voxel_size,
                            int world, int rank) {
    if (!out) return fail(ARVX_ERR_INVALID, "null out");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world)
        return fail(ARVX_ERR_INVALID, "rank %d of world %d", rank, world);
    if (Z < 1 || Z % 8) return fail(ARVX_ERR_INVALID, "striped slabs need Z %% 8 == 0 (Z=%d)", Z);
    const int groups = Z / 8;
    const int mine = (groups - rank + world - 1) / world;  // groups rank, rank+world, ...
    if (mine < 1) return fail(ARVX_ERR_INVALID, "rank %d owns no planes (Z=%d)", rank, Z);
    // build it as a contiguous context of mine*8 planes, then switch the z mapping
    int rc = arvx_ctx_create_slab(out, device, X, Y, Z, voxel_size, 0, mine * 8);
    if (rc) return rc;
    arvx_ctx *c = *out;
    c->ze0 = 0;  // no halo planes: neighbours in z belong to other ranks
    c->ze1 = mine * 8;
    c->nvox_ext = c->nvox;
    c->stripe_world = world;
    c->stripe_rank = rank;
    return ARVX_OK;
}

int arvx_ctx_create(arvx_ctx **out, int device, int X, int Y, int Z, float voxel_size) {
    return arvx_ctx_create_slab(out, device, X, Y, Z, voxel_size, 0, Z);
}

int arvx_ctx_destroy(arvx_ctx *ctx) {
    if (!ctx) return ARVX_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    ctx->free_views();
    ctx->free_color();
    ctx->free_mc();
    ctx->release_pools();
    if (ctx->d_flood) (void)hipFree(ctx->d_flood);
    if (ctx->d_flood_rec) (void)hipFree(ctx->d_flood_rec);
    if (ctx->d_rec) (void)hipFree(ctx->d_rec);
    if (ctx->d_state) (void)hipFree(ctx->d_state);
    if (ctx->d_stats) (void)hipFree(ctx->d_stats);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_coarse) (void)hipFree(ctx->d_coarse);
    if (ctx->d_stream) (void)hipFree(ctx->d_stream);
    if (ctx->h_fault) (void)hipHostFree(ctx->h_fault);
    if (ctx->own_stream) {
        (void)hipStreamSynchronize(ctx->own_stream);
        release_stream(ctx->device, ctx->own_stream);
    }
    delete ctx;
    return ARVX_OK;
}

int arvx_ctx_set_stream(arvx_ctx *ctx, void *hip_stream) {
    ARVX_CHECK_CTX(ctx);
    ARVX_SYNC(ctx);
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    // the carve's alternating list counters are zeroed by the PREVIOUS carve in stream order:
    // on another stream they start over (launch_carve zeroes both when the layout is unset)
    ctx->carve_layout = 0;
    return ARVX_OK;
}

int arvx_ctx_set_exchange_stream(arvx_ctx *ctx, void *hip_stream) {
    ARVX_CHECK_CTX(ctx);
    ctx->xstream = (hipStream_t)hip_stream;
    return ARVX_OK;
}

#ifdef ARVX_EXPERIMENTS
// Experiment builds only (tests/test_fault_gpu.py): leave the mark a kernel leaves when it gives up
// waiting for another workgroup, as if the launches since the last synchronisation had done so.
extern "C" int arvx_experiment_mark_fault(arvx_ctx *ctx, unsigned mark) {
    if (!ctx || !ctx->h_fault) return fail(ARVX_ERR_INVALID, "null context");
    *ctx->h_fault = mark;
    return ARVX_OK;
}
#endif

int arvx_ctx_synchronize(arvx_ctx *ctx) {
    ARVX_CHECK_CTX(ctx);
    ARVX_HIP(hipStreamSynchronize(ctx->stream));
    return check_fault(ctx);
}

int arvx_ctx_voxels(const arvx_ctx *ctx, int64_t *count) {
    if (!ctx || !count) return fail(ARVX_ERR_INVALID, "null argument");
    *count = (int64_t)ctx->nvox;
    return ARVX_OK;
}

// ---- views -------------------------------------------------------------------

static int views_common(Ctx *ctx, int V, const float *M, const float *campos, int W, int H,
                        int C) {
    if (V < 1 || V > 4096) return fail(ARVX_ERR_INVALID, "V=%d out of range [1,4096]", V);
    if (!M) return fail(ARVX_ERR_INVALID, "null M");
    if (W < 1 || H < 1 || W > arvx::kMaxImageDim || H > arvx::kMaxImageDim)
        return fail(ARVX_ERR_INVALID, "image size %dx%d out of range", W, H);
    if (C < 1 || C > 4) return fail(ARVX_ERR_INVALID, "channels=%d out of range [1,4]", C);
    // the same number and size of views as before: keep every buffer (a pipeline sends its
    // views once per stage, src/main.cpp:262-284)
    const bool same = ctx->d_M && ctx->V == V && ctx->W == W && ctx->H == H;
    ctx->views_ready = false;
    ctx->cameras_ready = false;
    ctx->free_surface();  // colour results belong to the previous views
    if (!same) {
        ctx->free_views();
        ctx->free_color();
    }
    ctx->images_ready = false;
    ctx->V = V;
    ctx->W = W;
    ctx->H = H;
    // one word more than the pixels need: bit 32 * (bgWords - 1) is a background bit that is
    // always 0, which the block-mapped exact kernel reads for voxels outside the image
    // (padded to whole 128-byte lines per view: views_strip_kernel writes a view's plane as
    // aligned 8-byte words and whole lines; only the last word is ever read as "always zero")
    ctx->bgWords = (int)(((((size_t)W * H + 31) / 32 + 1) + 31) / 32 * 32);
    // summed-area table (views_kernels.h): (H + 1) rows of W + 1 entries, the rows padded to
    // whole 128-byte lines so that the table kernel's stores are line-aligned
    ctx->satW = (W + 1 + 63) / 64 * 64;  // (two-byte entries: 64 per 128-byte line)
    ctx->satH = H + 1;
    ctx->satStride = ctx->satW * ctx->satH;
    if (!same) {
        hipError_t e = hipMalloc(&ctx->d_M, (size_t)V * 12 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&ctx->d_campos, (size_t)V * 3 * sizeof(float));
        if (e == hipSuccess) e = hipMalloc(&ctx->d_bg, (size_t)V * ctx->bgWords * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&ctx->d_sat, (size_t)V * ctx->satStride * sizeof(uint16_t));
        if (e != hipSuccess) {
            ctx->free_views();  // whatever was allocated before the failure
            return arvx::fail_hip(e, "hipMalloc(views)", __FILE__, __LINE__);
        }
    }
    // matrices and camera positions: sent only when they differ from what the device holds
    // (a pipeline re-derives its views per batch of masks with the same cameras; the two
    // small copies cost more than a kernel of the step each)
    const bool same_M = same && ctx->h_M.size() == (size_t)V * 12 &&
                        memcmp(ctx->h_M.data(), M, (size_t)V * 12 * sizeof(float)) == 0;
    if (!same_M) {
        ctx->h_M.assign(M, M + (size_t)V * 12);
        ARVX_HIP(hipMemcpyAsync(ctx->d_M, ctx->h_M.data(), (size_t)V * 12 * sizeof(float),
                                hipMemcpyHostToDevice, ctx->stream));
    }
    ctx->has_campos = campos != nullptr;
    if (campos) {
        const bool same_c = same && ctx->h_campos.size() == (size_t)V * 3 &&
                            memcmp(ctx->h_campos.data(), campos, (size_t)V * 3 * sizeof(float)) == 0;
        if (!same_c) {
            ctx->h_campos.assign(campos, campos + (size_t)V * 3);
            ARVX_HIP(hipMemcpyAsync(ctx->d_campos, ctx->h_campos.data(),
                                    (size_t)V * 3 * sizeof(float), hipMemcpyHostToDevice,
                                    ctx->stream));
        }
    } else {
        ctx->h_campos.clear();
    }
    return ARVX_OK;
}

static int views_preprocess(Ctx *ctx, const uint8_t *d_masks, int C) {
    const int npix = ctx->W * ctx->H;
    ARVX_HIP(hipGetLastError());  // anything stale would be blamed on the launches below
    // one- or three-channel masks with rows of whole 64-pixel tiles: ONE launch, a workgroup per strip
    // of 64 rows (views_kernels.h, views_strip_kernel); every other format: the three launches below
    static const bool three_launches = experiment_flag("ARVX_VIEWS_THREE_LAUNCHES");
    if ((C == 1 || C == 3) && ctx->W % 64 == 0 && ctx->W / 64 + 1 <= arvx::kStripMaxWaves && ctx->H <= 4096 &&
        ((uintptr_t)d_masks & 15u) == 0 && !three_launches) {
        const int TIs = (ctx->H + 63) / 64;
        // ticket counters (one per view, never reset) + the strips' published column counts;
        // zeroed once per layout (the tickets count launches from there)
        const size_t G = (size_t)ctx->W / 2;
        const size_t off_gran = ((size_t)ctx->V * sizeof(unsigned long long) + 255) / 256 * 256;
        const size_t vbytes = off_gran + (size_t)ctx->V * TIs * G * sizeof(unsigned long long);
        const size_t key = ((size_t)ctx->V << 40) ^ ((size_t)TIs << 24) ^ G;
        if (ctx->vstrip_key != key || ctx->pool_vstrip.cap < vbytes) {
            ARVX_HIP(ctx->pool_vstrip.reserve(vbytes));
            ARVX_HIP(hipMemsetAsync(ctx->pool_vstrip.p, 0, vbytes, ctx->stream));
            ctx->vstrip_key = key;
        }
        unsigned long long *ctr = (unsigned long long *)ctx->pool_vstrip.p;
        unsigned long long *gran = (unsigned long long *)((uint8_t *)ctx->pool_vstrip.p + off_gran);
        if (C == 1)
            hipLaunchKernelGGL(arvx::views_strip_kernel<1>, dim3(TIs, ctx->V), dim3(64 * (ctx->W / 64 + 1)),
                               0, ctx->stream, d_masks, ctx->W, ctx->H, TIs, ctx->d_bg, ctx->bgWords,
                               ctx->d_sat, ctx->satStride, ctx->satW, ctr, gran, ctx->d_fault);
        else
            hipLaunchKernelGGL(arvx::views_strip_kernel<3>, dim3(TIs, ctx->V), dim3(64 * (ctx->W / 64 + 1)),
                               0, ctx->stream, d_masks, ctx->W, ctx->H, TIs, ctx->d_bg, ctx->bgWords,
                               ctx->d_sat, ctx->satStride, ctx->satW, ctr, gran, ctx->d_fault);
        ARVX_HIP(hipGetLastError());
        ctx->views_ready = true;
        ctx->cameras_ready = true;
        return ARVX_OK;
    }
    if (C == 1 && npix % 32 == 0 && ((uintptr_t)d_masks & 15u) == 0) {
        hipLaunchKernelGGL(arvx::views_bits16_kernel, dim3((npix / 16 + 255) / 256, ctx->V),
                           dim3(256), 0, ctx->stream, d_masks, npix, ctx->d_bg, ctx->bgWords);
    } else if (C == 1 || C == 3) {
        const dim3 g1(((npix + 3) / 4 + 255) / 256, ctx->V);
        if (C == 1)
            hipLaunchKernelGGL(arvx::views_bits_kernel<1>, g1, dim3(256), 0, ctx->stream, d_masks,
                               npix, ctx->d_bg, ctx->bgWords);
        else
            hipLaunchKernelGGL(arvx::views_bits_kernel<3>, g1, dim3(256), 0, ctx->stream, d_masks,
                               npix, ctx->d_bg, ctx->bgWords);
    } else {
        const dim3 g1((npix + 255) / 256, ctx->V);
        hipLaunchKernelGGL(arvx::views_bits_generic_kernel, g1, dim3(256), 0, ctx->stream, d_masks,
                           C, npix, ctx->d_bg, ctx->bgWords);
    }
    ARVX_HIP(hipGetLastError());
    const int W = ctx->W, H = ctx->H, V = ctx->V;
    // tiles of 64 table columns x 64 image rows (views_kernels.h)
    const int TJ = (W + 1 + 63) / 64, TI = (H + arvx::kTileRows - 1) / arvx::kTileRows;
    const size_t n_rs = (size_t)V * H * TJ, n_T = (size_t)V * TI * TJ * 64, n_ts = (size_t)V * TI * TJ;
    if (int rc = ensure_scratch(ctx, (n_rs + n_T + n_ts) * sizeof(int) + 64)) return rc;
    int *d_rs = (int *)ctx->d_scratch, *d_T = d_rs + n_rs, *d_ts = d_T + n_T;
    const unsigned tgrid = (unsigned)(((size_t)TJ * TI * V + 3) / 4);
    hipLaunchKernelGGL(arvx::views_tile_sums_kernel, dim3(tgrid), dim3(256), 0, ctx->stream,
                       ctx->d_bg, ctx->bgWords, W, H, TJ, TI, V, d_rs, d_T, d_ts);
    hipLaunchKernelGGL(arvx::views_table_kernel, dim3(tgrid), dim3(256), 0, ctx->stream,
                       ctx->d_bg, ctx->bgWords, W, H, TJ, TI, V, d_rs, d_T, d_ts, ctx->d_sat,
                       ctx->satStride, ctx->satW);
    ARVX_HIP(hipGetLastError());
    ctx->views_ready = true;
    ctx->cameras_ready = true;
    return ARVX_OK;
}

int arvx_set_views(arvx_ctx *ctx, int V, const float *M, const float *campos,
                   const uint8_t *const *masks, int W, int H, int C, size_t stride) {
    ARVX_CHECK_CTX(ctx);
    if (!masks) {  // cameras only: enough for arvx_color
        if (int rc0 = views_common(ctx, V, M, campos, W, H, C > 0 ? C : 1)) return rc0;
        ctx->cameras_ready = true;
        ARVX_SYNC(ctx);
        return ARVX_OK;
    }
    if (W >= 1 && C >= 1 && stride < (size_t)W * C)
        return fail(ARVX_ERR_INVALID, "stride %zu < W*C", stride);
    for (int i = 0; i < V; ++i)
        if (!masks[i]) return fail(ARVX_ERR_INVALID, "null mask %d", i);
    int rc = views_common(ctx, V, M, campos, W, H, C);
    if (rc) return rc;
    const size_t img = (size_t)W * H * C;
    ARVX_HIP(ctx->pool_raw_masks.reserve(img * V));
    uint8_t *d_raw = (uint8_t *)ctx->pool_raw_masks.p;
    bool packed = stride == (size_t)W * C;  // all views back to back in host memory: one copy
    for (int i = 1; i < V && packed; ++i) packed = masks[i] == masks[i - 1] + img;
    if (packed) {
        ARVX_HIP(hipMemcpyAsync(d_raw, masks[0], img * V, hipMemcpyHostToDevice, ctx->stream));
    } else {
        for (int i = 0; i < V; ++i)
            ARVX_HIP(hipMemcpy2DAsync(d_raw + img * i, (size_t)W * C, masks[i], stride,
                                      (size_t)W * C, H, hipMemcpyHostToDevice, ctx->stream));
    }
    rc = views_preprocess(ctx, d_raw, C);
    if (rc) return rc;
    ARVX_SYNC(ctx);  // host buffers may go away
    return ARVX_OK;
}

int arvx_set_views_device(arvx_ctx *ctx, int V, const float *M, const float *campos,
                          const void *dev_masks, int W, int H, int C) {
    ARVX_CHECK_CTX(ctx);
    if (!dev_masks) return fail(ARVX_ERR_INVALID, "null dev_masks");
    if (int rc = views_common(ctx, V, M, campos, W, H, C)) return rc;
    return views_preprocess(ctx, (const uint8_t *)dev_masks, C);
}

static int undistort_params(const double K[9], const double *dist, int ndist, int W, int H,
                            int C, arvx::UndistortParams &p) {
    if (!K || (!dist && ndist)) return fail(ARVX_ERR_INVALID, "null calibration");
    if (ndist != 0 && ndist != 4 && ndist != 5 && ndist != 8)
        return fail(ARVX_ERR_INVALID, "distortion coefficients: 4, 5 or 8 (got %d)", ndist);
    if (W < 1 || H < 1 || C < 1 || C > 4) return fail(ARVX_ERR_INVALID, "bad image format");
    if (K[0] == 0 || K[4] == 0) return fail(ARVX_ERR_INVALID, "singular camera matrix");
    memset(&p, 0, sizeof p);
    p.fx = K[0];
    p.fy = K[4];
    p.cx = K[2];
    p.cy = K[5];
    p.ir0 = 1.0 / p.fx;  // inv(A) of an upper-triangular A with a unit last row
    p.ir2 = -p.cx / p.fx;
    p.ir4 = 1.0 / p.fy;
    p.ir5 = -p.cy / p.fy;
    double k[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < ndist; ++i) k[i] = dist[i];
    p.k1 = k[0];
    p.k2 = k[1];
    p.p1 = k[2];
    p.p2 = k[3];
    p.k3 = k[4];
    p.k4 = k[5];
    p.k5 = k[6];
    p.k6 = k[7];
    p.W = W;
    p.H = H;
    p.C = C;
    return ARVX_OK;
}

int arvx_undistort_device(arvx_ctx *ctx, int V, const void *dev_src, int W, int H, int C,
                          const double K[9], const double *dist, int ndist, void *dev_dst) {
    ARVX_CHECK_CTX(ctx);
    if (!dev_src || !dev_dst || dev_src == dev_dst || V < 1)
        return fail(ARVX_ERR_INVALID, "bad argument (src and dst must differ)");
    arvx::UndistortParams p;
    if (int rc = undistort_params(K, dist, ndist, W, H, C, p)) return rc;
    hipLaunchKernelGGL(arvx::undistort_kernel, dim3((W + 63) / 64, (H + 3) / 4, V), dim3(256), 0,
                       ctx->stream, (const uint8_t *)dev_src, (uint8_t *)dev_dst, p);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

int arvx_undistort(arvx_ctx *ctx, int V, const uint8_t *const *src, int W, int H, int C,
                   size_t stride, const double K[9], const double *dist, int ndist,
                   uint8_t *const *dst) {
    ARVX_CHECK_CTX(ctx);
    if (!src || !dst || V < 1) return fail(ARVX_ERR_INVALID, "bad argument");
    arvx::UndistortParams p;
    if (int rc = undistort_params(K, dist, ndist, W, H, C, p)) return rc;
    const size_t rowb = (size_t)W * C, img = rowb * H;
    if (stride < rowb) return fail(ARVX_ERR_INVALID, "stride %zu < W*C", stride);
    for (int i = 0; i < V; ++i)
        if (!src[i] || !dst[i]) return fail(ARVX_ERR_INVALID, "null image %d", i);
    if (int rc = ensure_scratch(ctx, 2 * img * V + 64)) return rc;
    uint8_t *d_src = (uint8_t *)ctx->d_scratch, *d_dst = d_src + img * V;
    for (int i = 0; i < V; ++i)
        ARVX_HIP(hipMemcpy2DAsync(d_src + img * i, rowb, src[i], stride, rowb, H,
                                  hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(arvx::undistort_kernel, dim3((W + 63) / 64, (H + 3) / 4, V), dim3(256), 0,
                       ctx->stream, d_src, d_dst, p);
    ARVX_HIP(hipGetLastError());
    for (int i = 0; i < V; ++i)
        ARVX_HIP(hipMemcpy2DAsync(dst[i], stride, d_dst + img * i, rowb, rowb, H,
                                  hipMemcpyDeviceToHost, ctx->stream));
    ARVX_SYNC(ctx);
    return ARVX_OK;
}

// ---- state -------------------------------------------------------------------

// grid geometry of a context as the kernels see it (planes ze0..ze1-1: owned + halo)
static void carve_geometry(const Ctx *ctx, arvx::CarveParams &p) {
    memset(&p, 0, sizeof p);
    p.X = ctx->X;
    p.Y = ctx->Y;
    p.Z = ctx->ze1 - ctx->ze0;  // owned planes plus halo (recomputed, never exchanged)
    p.zoff = ctx->ze0;
    p.zstride = ctx->stripe_world;
    p.zphase = ctx->stripe_rank;
    p.s = ctx->s;
    p.tilesX = (p.X + arvx::kTileX - 1) / arvx::kTileX;
    p.tilesY = (p.Y + arvx::kTileY - 1) / arvx::kTileY;
    p.tilesZ = (p.Z + arvx::kTileZ - 1) / arvx::kTileZ;
    // coarse tile 64x32x32; striped slabs: 64x64x8, so that it stays inside one stripe
    p.cyShift = ctx->stripe_world > 1 ? 3 : 2;
    p.czShift = ctx->stripe_world > 1 ? 0 : 2;
    p.coarseX = (p.X + arvx::kCoarseX - 1) / arvx::kCoarseX;
    p.coarseY = (p.Y + (8 << p.cyShift) - 1) / (8 << p.cyShift);
    p.coarseZ = (p.Z + (8 << p.czShift) - 1) / (8 << p.czShift);
    p.ccode = ctx->lazy ? (const uint8_t *)ctx->pool_ccode.p : nullptr;
}

// a record buffer for this context's grid, every record "finished" when it is new
static int ensure_records(Ctx *ctx, void **buf, size_t *cap) {
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    const size_t need = arvx::rec_count(g) * arvx::kRecU16 * sizeof(uint16_t);
    if (*cap >= need) return ARVX_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    ARVX_HIP(hipMalloc(buf, need));
    *cap = need;
    hipLaunchKernelGGL(arvx::rec_init_kernel, dim3(2048), dim3(256), 0, ctx->stream,
                       (uint32_t *)*buf, need / 4);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

// The state as records: what every stage works on.  lazy_ok: the caller's kernels read the
// coarse tiles a fresh carve settled as a whole from their codes (CarveParams::ccode); everybody
// else gets those tiles written out first -- the fill the carve itself skipped.
static int need_rec(Ctx *ctx, bool lazy_ok) {
    void *buf = ctx->d_rec;
    if (int rc = ensure_records(ctx, &buf, &ctx->rec_bytes)) return rc;
    ctx->d_rec = (uint16_t *)buf;
    arvx::CarveParams g;
    if (ctx->rec_valid) {
        if (ctx->lazy && !lazy_ok) {
            carve_geometry(ctx, g);
            g.rec = ctx->d_rec;
            g.ccode = nullptr;
            g.coarseCarved = (uint8_t *)ctx->pool_ccode.p;
            g.flags = 4u;  // codes 2 and 3 are those of a fresh model: written as such
            hipLaunchKernelGGL(arvx::carve_fill_kernel,
                               dim3((unsigned)((size_t)g.coarseX * g.coarseY * g.coarseZ)), dim3(256),
                               0, ctx->stream, g);
            ARVX_HIP(hipGetLastError());
            ctx->lazy = false;
        }
        return ARVX_OK;
    }
    ctx->lazy = false;
    carve_geometry(ctx, g);
    g.rec = ctx->d_rec;
    g.flags = 4u | 16u;  // a fresh model: all occupied, none seen
    hipLaunchKernelGGL(arvx::carve_fill_kernel,
                       dim3((unsigned)((size_t)g.coarseX * g.coarseY * g.coarseZ)), dim3(256), 0,
                       ctx->stream, g);
    ARVX_HIP(hipGetLastError());
    ctx->fresh_pending = false;
    ctx->rec_valid = true;
    return ARVX_OK;
}

// every call that changes occupied / seen bits goes through here: results derived from the
// old state are dropped
static void state_changes(Ctx *ctx, bool keeps_paint) {
    ++ctx->state_seq;
    ctx->color_ready = false;
    ctx->closure_ready = false;
    if (!keeps_paint) ctx->paint_valid = false;
}
// ... and one that may occupy or un-see voxels (everything but carving, the greedy carve and
// handleUnseen, which leave a tile that is carved and seen / seen as a whole as it is) also
// drops what earlier carves settled for whole coarse tiles (CarveParams::cstate)
static void state_rewritten(Ctx *ctx) { ctx->cstate_tiles = 0; }

// Bytes of local planes [zl0, zl0 + nz), already in the staging buffer, into the records
// (+ bit2 into the paint plane).  Ends with a host synchronisation.
static int bytes_into_records(Ctx *ctx, int zl0, int nz) {
    if (int rc = need_rec(ctx)) return rc;
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    g.rec = ctx->d_rec;
    const size_t pw = (size_t)((ctx->X + 63) / 64) * ctx->Y * (size_t)(ctx->ze1 - ctx->ze0);
    const bool had = ctx->pool_paint.cap >= pw * 8 && ctx->paint_valid;
    ARVX_HIP(ctx->pool_paint.reserve(pw * 8));
    if (!had) ARVX_HIP(hipMemsetAsync(ctx->pool_paint.p, 0, pw * 8, ctx->stream));
    int *d_any = (int *)(ctx->d_stats + 8);
    ARVX_HIP(hipMemsetAsync(d_any, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(arvx::rec_from_bytes_kernel,
                       dim3((unsigned)((size_t)g.tilesX * g.tilesY * g.tilesZ)), dim3(256), 0,
                       ctx->stream, g, (const uint8_t *)ctx->d_state, zl0, nz,
                       (unsigned long long *)ctx->pool_paint.p, d_any);
    ARVX_HIP(hipGetLastError());
    int any = 0;
    ARVX_HIP(hipMemcpyAsync(&any, d_any, sizeof any, hipMemcpyDeviceToHost, ctx->stream));
    ARVX_SYNC(ctx);  // (also: the host bytes may go away)
    ctx->paint_valid = had || any != 0;
    return ARVX_OK;
}

// the byte staging buffer holds the current state of local planes [zl0, zl0 + nz)
static int records_into_bytes(Ctx *ctx, int zl0, int nz) {
    if (int rc = need_rec(ctx, true)) return rc;  // (rec_to_bytes_kernel reads lazy tiles)
    if (!ctx->d_state) ARVX_HIP(hipMalloc(&ctx->d_state, ctx->nvox_ext));
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    g.rec = ctx->d_rec;
    hipLaunchKernelGGL(arvx::rec_to_bytes_kernel,
                       dim3((unsigned)((size_t)g.tilesX * g.tilesY * g.tilesZ)), dim3(256), 0,
                       ctx->stream, g, ctx->d_state, zl0, nz, paint_plane(ctx));
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

int arvx_state_reset(arvx_ctx *ctx) {
    ARVX_CHECK_CTX(ctx);
    state_changes(ctx, false);
    state_rewritten(ctx);
    ctx->fresh_pending = true;  // materialised by need_rec, or never (a carve of a fresh model
    ctx->rec_valid = false;     // writes every record)
    ctx->lazy = false;
    return ARVX_OK;
}

int arvx_state_upload(arvx_ctx *ctx, const uint8_t *state) {
    ARVX_CHECK_CTX(ctx);
    if (!state) return fail(ARVX_ERR_INVALID, "null state");
    if (!ctx->d_state) ARVX_HIP(hipMalloc(&ctx->d_state, ctx->nvox_ext));
    // paint of the owned planes is replaced; what the halo planes hold stays
    const bool halo_paint = ctx->paint_valid && ctx->nvox_ext != ctx->nvox;
    state_changes(ctx, halo_paint);
    state_rewritten(ctx);
    ARVX_HIP(hipMemcpyAsync(ctx->owned(), state, ctx->nvox, hipMemcpyHostToDevice, ctx->stream));
    return bytes_into_records(ctx, ctx->z0 - ctx->ze0, ctx->z1 - ctx->z0);
}

int arvx_state_upload_halo(arvx_ctx *ctx, const uint8_t *plane_below, const uint8_t *plane_above) {
    ARVX_CHECK_CTX(ctx);
    const size_t plane = (size_t)ctx->X * ctx->Y;
    if (ctx->stripe_world > 1) return fail(ARVX_ERR_STATE, "striped slabs keep no halo planes");
    if (!ctx->d_state) ARVX_HIP(hipMalloc(&ctx->d_state, ctx->nvox_ext));
    state_changes(ctx, true);
    state_rewritten(ctx);
    if (plane_below && ctx->ze0 < ctx->z0) {  // plane z0 - 1
        ARVX_HIP(hipMemcpyAsync(ctx->owned() - plane, plane_below, plane, hipMemcpyHostToDevice,
                                ctx->stream));
        if (int rc = bytes_into_records(ctx, ctx->z0 - ctx->ze0 - 1, 1)) return rc;
    }
    if (plane_above && ctx->ze1 > ctx->z1) {
        ARVX_HIP(hipMemcpyAsync(ctx->owned() + ctx->nvox, plane_above, plane,
                                hipMemcpyHostToDevice, ctx->stream));
        if (int rc = bytes_into_records(ctx, ctx->z1 - ctx->ze0, 1)) return rc;
    }
    return ARVX_OK;
}

int arvx_state_download(arvx_ctx *ctx, uint8_t *state) {
    ARVX_CHECK_CTX(ctx);
    if (!state) return fail(ARVX_ERR_INVALID, "null state");
    if (int mrc = records_into_bytes(ctx, ctx->z0 - ctx->ze0, ctx->z1 - ctx->z0)) return mrc;
    ARVX_HIP(hipMemcpyAsync(state, ctx->owned(), ctx->nvox, hipMemcpyDeviceToHost, ctx->stream));
    ARVX_SYNC(ctx);
    return ARVX_OK;
}

int arvx_state_upload_planes(arvx_ctx *ctx, const uint32_t *occ, const uint32_t *seen) {
    ARVX_CHECK_CTX(ctx);
    if (!occ || !seen) return fail(ARVX_ERR_INVALID, "null plane");
    if (int mrc = need_rec(ctx)) return mrc;  // (halo planes keep what they hold)
    state_changes(ctx, false);
    state_rewritten(ctx);
    const int nz = ctx->z1 - ctx->z0;
    const size_t nwords = (size_t)((ctx->X + 31) / 32) * ctx->Y * nz;
    if (int rc = ensure_scratch(ctx, 2 * nwords * sizeof(uint32_t) + 64)) return rc;
    uint32_t *d_occ = (uint32_t *)ctx->d_scratch, *d_seen = d_occ + nwords;
    ARVX_HIP(hipMemcpyAsync(d_occ, occ, nwords * 4, hipMemcpyHostToDevice, ctx->stream));
    ARVX_HIP(hipMemcpyAsync(d_seen, seen, nwords * 4, hipMemcpyHostToDevice, ctx->stream));
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    g.rec = ctx->d_rec;
    hipLaunchKernelGGL(arvx::rec_from_planes_kernel, dim3((unsigned)((nwords + 255) / 256)),
                       dim3(256), 0, ctx->stream, g, ctx->z0 - ctx->ze0, nz, d_occ, d_seen);
    ARVX_HIP(hipGetLastError());
    ARVX_SYNC(ctx);  // the host planes may go away
    return ARVX_OK;
}

int arvx_state_download_planes(arvx_ctx *ctx, uint32_t *occ, uint32_t *seen) {
    ARVX_CHECK_CTX(ctx);
    if (!occ || !seen) return fail(ARVX_ERR_INVALID, "null plane");
    if (int mrc = need_rec(ctx, true)) return mrc;  // (planes_from_rec_kernel reads lazy tiles)
    const int nz = ctx->z1 - ctx->z0;
    const size_t nwords = (size_t)((ctx->X + 31) / 32) * ctx->Y * nz;
    if (int rc = ensure_scratch(ctx, 2 * nwords * sizeof(uint32_t) + 64)) return rc;
    uint32_t *d_occ = (uint32_t *)ctx->d_scratch, *d_seen = d_occ + nwords;
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    g.rec = ctx->d_rec;
    hipLaunchKernelGGL(arvx::planes_from_rec_kernel, dim3((unsigned)((nwords + 255) / 256)),
                       dim3(256), 0, ctx->stream, g, ctx->z0 - ctx->ze0, nz, d_occ, d_seen);
    ARVX_HIP(hipGetLastError());
    ARVX_HIP(hipMemcpyAsync(occ, d_occ, nwords * 4, hipMemcpyDeviceToHost, ctx->stream));
    ARVX_HIP(hipMemcpyAsync(seen, d_seen, nwords * 4, hipMemcpyDeviceToHost, ctx->stream));
    ARVX_SYNC(ctx);
    return ARVX_OK;
}

int arvx_state_packet_geometry(arvx_ctx *ctx, int64_t *words64, int64_t *header_words) {
    if (!ctx || !words64 || !header_words) return fail(ARVX_ERR_INVALID, "null argument");
    const size_t plane = (size_t)ctx->X * ctx->Y;
    if (ctx->X % 32 || plane % 64)
        return fail(ARVX_ERR_INVALID, "state packets need X %% 32 == 0 and X*Y %% 64 == 0 (X=%d, Y=%d)", ctx->X,
                    ctx->Y);
    const long long n = (long long)(plane / 64) * (ctx->z1 - ctx->z0);
    if (2 * n >= (1ll << 32)) return fail(ARVX_ERR_INVALID, "slab too large for state packets (%lld words)", n);
    *words64 = n;
    *header_words = arvx::occ_packet_header(n);
    return ARVX_OK;
}

int arvx_state_download_packets(arvx_ctx *ctx, uint64_t *occ_packet, int64_t occ_cap, uint64_t *seen_packet,
                                int64_t seen_cap, int64_t *occ_need, int64_t *seen_need) {
    ARVX_CHECK_CTX(ctx);
    if (!occ_packet || !seen_packet || !occ_need || !seen_need || occ_cap < 0 || seen_cap < 0)
        return fail(ARVX_ERR_INVALID, "bad argument");
    int64_t n64 = 0, H64 = 0;
    if (int rc = arvx_state_packet_geometry(ctx, &n64, &H64)) return rc;
    const long long n = n64, H = H64, nb = (n + 63) / 64;
    // the two packets at their worst-case size stay on the device until the state changes: a caller
    // whose buffers were too small asks again and only the copies run
    const size_t S = (size_t)(H + n);
    if (!(ctx->packets_seq == ctx->state_seq && ctx->packets_valid && ctx->pool_state_packets.cap >= 2 * S * 8)) {
        if (int mrc = need_rec(ctx, true)) return mrc;
        ARVX_HIP(ctx->pool_state_packets.reserve(2 * S * 8));
        arvx::CarveParams g;
        carve_geometry(ctx, g);
        g.rec = ctx->d_rec;
        const int zl0 = ctx->z0 - ctx->ze0;
        const int nwg = (int)((nb + arvx::kOccGroupsPerWg - 1) / arvx::kOccGroupsPerWg);
        arvx::OccGeom og;
        og.wpr = arvx::fast_div((unsigned)(ctx->X / 32));
        og.Y = arvx::fast_div((unsigned)ctx->Y);
        og.P64 = arvx::fast_div((unsigned)((size_t)ctx->X * ctx->Y / 64));
        if (int rc = ensure_scratch(ctx, 2 * (size_t)nwg * sizeof(int) + 64)) return rc;
        int *d_wgsum = (int *)ctx->d_scratch;
        unsigned long long *pk = (unsigned long long *)ctx->pool_state_packets.p;
        hipLaunchKernelGGL(arvx::occ_pack_classify_kernel<false>, dim3(nwg), dim3(256), 0, ctx->stream, g, og,
                           zl0, n, pk, d_wgsum, (unsigned long long *)nullptr);
        hipLaunchKernelGGL(arvx::occ_pack_classify_kernel<true>, dim3(nwg), dim3(256), 0, ctx->stream, g, og,
                           zl0, n, pk + S, d_wgsum + nwg, (unsigned long long *)nullptr);
        hipLaunchKernelGGL(arvx::occ_pack_write_kernel<false>, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0,
                           ctx->stream, g, og, zl0, n, n, d_wgsum, nwg, pk);
        hipLaunchKernelGGL(arvx::occ_pack_write_kernel<true>, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0,
                           ctx->stream, g, og, zl0, n, n, d_wgsum + nwg, nwg, pk + S);
        ARVX_HIP(hipGetLastError());
        // the two counts first (16 bytes through the page-locked totals: slots 4 and 5)
        ARVX_HIP(hipMemcpyAsync(ctx->h_totals + 4, pk, 8, hipMemcpyDeviceToHost, ctx->stream));
        ARVX_HIP(hipMemcpyAsync(ctx->h_totals + 5, pk + S, 8, hipMemcpyDeviceToHost, ctx->stream));
        ARVX_SYNC(ctx);
        ctx->packet_need[0] = ctx->h_totals[4];
        ctx->packet_need[1] = ctx->h_totals[5];
        ctx->packets_seq = ctx->state_seq;
        ctx->packets_valid = true;
    }
    const unsigned long long *pk = (const unsigned long long *)ctx->pool_state_packets.p;
    *occ_need = ctx->packet_need[0];
    *seen_need = ctx->packet_need[1];
    ARVX_HIP(hipMemcpyAsync(occ_packet, pk, (size_t)(H + std::min<long long>(*occ_need, occ_cap)) * 8,
                            hipMemcpyDeviceToHost, ctx->stream));
    ARVX_HIP(hipMemcpyAsync(seen_packet, pk + S, (size_t)(H + std::min<long long>(*seen_need, seen_cap)) * 8,
                            hipMemcpyDeviceToHost, ctx->stream));
    ARVX_SYNC(ctx);
    return ARVX_OK;
}

int arvx_handle_unseen(arvx_ctx *ctx) {
    ARVX_CHECK_CTX(ctx);
    // (coarse tiles that exist only as their code stay codes: "carved and seen", "untouched and
    // seen" and "untouched, not seen" are all unchanged by occ |= ~seen -- the records behind a
    // code are not read by anybody)
    if (int mrc = need_rec(ctx, true)) return mrc;
    if (ctx->planes_ok && ctx->planes_seq == ctx->state_seq) {
        // the colour pass's planes stay usable: occupied is now their occupancy | never-seen
        ctx->planes_seq = ctx->state_seq + 1;
        ctx->planes_unseen = true;
    }
    ++ctx->state_seq;
    ctx->closure_ready = false;  // (colours and paint stay: only never-seen voxels change)
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    const size_t nrec = arvx::rec_count(g);
    hipLaunchKernelGGL(arvx::rec_handle_unseen_kernel, dim3((unsigned)((nrec * 32 + 255) / 256)),
                       dim3(256), 0, ctx->stream, (uint32_t *)ctx->d_rec, nrec, g.ccode,
                       g.cyShift + g.czShift + 2);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

int arvx_host_register(void *ptr, size_t bytes) {
    if (!ptr || !bytes) return fail(ARVX_ERR_INVALID, "null host range");
    ARVX_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return ARVX_OK;
}

int arvx_host_unregister(void *ptr) {
    if (!ptr) return fail(ARVX_ERR_INVALID, "null host pointer");
    ARVX_HIP(hipHostUnregister(ptr));
    return ARVX_OK;
}

int arvx_state_device_ptr(arvx_ctx *ctx, void **ptr, size_t *bytes) {
    if (!ctx || !ptr) return fail(ARVX_ERR_INVALID, "null argument");
    ARVX_HIP(hipSetDevice(ctx->device));
    if (int mrc = records_into_bytes(ctx, ctx->z0 - ctx->ze0, ctx->z1 - ctx->z0)) return mrc;
    *ptr = ctx->owned();
    if (bytes) *bytes = ctx->nvox;
    return ARVX_OK;
}

// X % 64 == 0 and an 8-byte aligned destination: the tile-wise pack (state_kernels.h)
static int launch_pack_tiles(Ctx *ctx, int global, void *dev_words) {
    if (int mrc = need_rec(ctx, true)) return mrc;
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    g.rec = ctx->d_rec;
    const int zl0 = ctx->z0 - ctx->ze0, nz = ctx->z1 - ctx->z0;
    const int ntz = ((zl0 + nz - 1) >> 3) - (zl0 >> 3) + 1;
    const size_t tiles = (size_t)g.tilesX * g.tilesY * ntz;
    hipLaunchKernelGGL(arvx::pack_occupancy_tile_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0,
                       ctx->stream, g, zl0, nz, global, (unsigned long long *)dev_words);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

int arvx_pack_occupancy(arvx_ctx *ctx, void *dev_words) {
    ARVX_CHECK_CTX(ctx);
    ExchangeStreamScope exchange_scope(ctx);
    if (!dev_words) return fail(ARVX_ERR_INVALID, "null dev_words");
    if (ctx->X % 64 == 0 && (uintptr_t)dev_words % 8 == 0) return launch_pack_tiles(ctx, 0, dev_words);
    if (ctx->X % 32 == 0 && (uintptr_t)dev_words % 4 == 0) {
        // straight from the records: 2 bits per voxel in, 1 out
        if (int mrc = need_rec(ctx, true)) return mrc;
        arvx::CarveParams g;
        carve_geometry(ctx, g);
        g.rec = ctx->d_rec;
        const int nz = ctx->z1 - ctx->z0;
        const size_t nwords = (size_t)(ctx->X / 32) * ctx->Y * nz;
        hipLaunchKernelGGL(arvx::pack_occupancy_rec_kernel, dim3((unsigned)((nwords + 255) / 256)),
                           dim3(256), 0, ctx->stream, g, ctx->z0 - ctx->ze0, nz, 0,
                           (uint32_t *)dev_words);
        ARVX_HIP(hipGetLastError());
        return ARVX_OK;
    }
    if (ctx->X % 8 == 0 && ((size_t)ctx->X * ctx->Y) % 32 == 0 && (uintptr_t)dev_words % 4 == 0) {
        if (int mrc = need_rec(ctx, true)) return mrc;
        arvx::CarveParams g;
        carve_geometry(ctx, g);
        g.rec = ctx->d_rec;
        const int nz = ctx->z1 - ctx->z0;
        const size_t nwords = ((size_t)ctx->X * ctx->Y * nz / 8 + 3) / 4;
        hipLaunchKernelGGL(arvx::pack_occupancy_rec8_kernel, dim3((unsigned)((nwords + 255) / 256)),
                           dim3(256), 0, ctx->stream, g, ctx->z0 - ctx->ze0, nz, 0,
                           (uint32_t *)dev_words);
        ARVX_HIP(hipGetLastError());
        return ARVX_OK;
    }
    // odd row lengths: through the byte form
    if (int mrc = records_into_bytes(ctx, ctx->z0 - ctx->ze0, ctx->z1 - ctx->z0)) return mrc;
    if (ctx->nvox % 32 == 0 && (uintptr_t)ctx->owned() % 16 == 0 && (uintptr_t)dev_words % 4 == 0) {
        const size_t nwords = ctx->nvox / 32;
        hipLaunchKernelGGL(arvx::pack_occupancy32_kernel, dim3((unsigned)((nwords + 255) / 256)),
                           dim3(256), 0, ctx->stream, ctx->owned(), nwords, (uint32_t *)dev_words);
    } else if (ctx->nvox % 32 == 0 && (uintptr_t)ctx->owned() % 8 == 0) {
        const size_t nbytes = ctx->nvox / 8;
        hipLaunchKernelGGL(arvx::pack_occupancy8_kernel, dim3((unsigned)((nbytes + 255) / 256)),
                           dim3(256), 0, ctx->stream, ctx->owned(), nbytes, (uint8_t *)dev_words);
    } else {
        const size_t nround = (ctx->nvox + 255) / 256;
        const unsigned grid = (unsigned)(nround < 8192 ? (nround ? nround : 1) : 8192);
        hipLaunchKernelGGL(arvx::pack_occupancy_kernel, dim3(grid), dim3(256), 0, ctx->stream,
                           ctx->owned(), ctx->nvox, (uint32_t *)dev_words);
    }
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

int arvx_pack_occupancy_global(arvx_ctx *ctx, void *dev_global_words) {
    ARVX_CHECK_CTX(ctx);
    ExchangeStreamScope exchange_scope(ctx);
    if (!dev_global_words) return fail(ARVX_ERR_INVALID, "null dev_global_words");
    const size_t plane = (size_t)ctx->X * ctx->Y;
    if (plane % 64) return fail(ARVX_ERR_INVALID, "X*Y must be a multiple of 64");
    if (ctx->X % 64 == 0 && (uintptr_t)dev_global_words % 8 == 0)
        return launch_pack_tiles(ctx, 1, dev_global_words);
    if (ctx->X % 32 == 0) {
        if (int mrc = need_rec(ctx, true)) return mrc;
        arvx::CarveParams g;
        carve_geometry(ctx, g);
        g.rec = ctx->d_rec;
        const int nz = ctx->z1 - ctx->z0;
        const size_t nwords = (size_t)(ctx->X / 32) * ctx->Y * nz;
        hipLaunchKernelGGL(arvx::pack_occupancy_rec_kernel, dim3((unsigned)((nwords + 255) / 256)),
                           dim3(256), 0, ctx->stream, g, ctx->z0 - ctx->ze0, nz, 1,
                           (uint32_t *)dev_global_words);
        ARVX_HIP(hipGetLastError());
        return ARVX_OK;
    }
    if (ctx->X % 8 == 0) {  // (X * Y % 64 == 0 above)
        if (int mrc = need_rec(ctx, true)) return mrc;
        arvx::CarveParams g;
        carve_geometry(ctx, g);
        g.rec = ctx->d_rec;
        const int nz = ctx->z1 - ctx->z0;
        const size_t nwords = (size_t)ctx->X * ctx->Y * nz / 32;
        hipLaunchKernelGGL(arvx::pack_occupancy_rec8_kernel, dim3((unsigned)((nwords + 255) / 256)),
                           dim3(256), 0, ctx->stream, g, ctx->z0 - ctx->ze0, nz, 1,
                           (uint32_t *)dev_global_words);
        ARVX_HIP(hipGetLastError());
        return ARVX_OK;
    }
    if (int mrc = records_into_bytes(ctx, ctx->z0 - ctx->ze0, ctx->z1 - ctx->z0)) return mrc;
    // plane % 64 == 0 and an aligned allocation: every plane starts on an 8-byte boundary
    const size_t nbytes = ctx->nvox / 8;
    hipLaunchKernelGGL(arvx::pack_occupancy_global8_kernel, dim3((unsigned)((nbytes + 255) / 256)),
                       dim3(256), 0, ctx->stream, ctx->owned(), plane, ctx->z1 - ctx->z0,
                       ctx->stripe_world > 1 ? 0 : ctx->z0, ctx->stripe_world, ctx->stripe_rank,
                       (uint8_t *)dev_global_words);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

// ---- compressed occupancy exchange -----------------------------------------------------

int64_t arvx_occupancy_packet_words(int64_t n_words64, int64_t cap_words64) {
    if (n_words64 < 0 || cap_words64 < 0) return -1;
    return (int64_t)arvx::occ_packet_header(n_words64) + cap_words64;
}

int arvx_occupancy_compress(arvx_ctx *ctx, const void *dev_words, int64_t n_words64,
                            void *dev_packet, int64_t cap_words64) {
    ARVX_CHECK_CTX(ctx);
    ExchangeStreamScope exchange_scope(ctx);
    if (!dev_words || !dev_packet || n_words64 <= 0 || cap_words64 < 0)
        return fail(ARVX_ERR_INVALID, "bad argument");
    if (((uintptr_t)dev_words | (uintptr_t)dev_packet) & 7u)
        return fail(ARVX_ERR_INVALID, "buffers must be 8-byte aligned");
    const long long n = n_words64, nb = (n + 63) / 64;
    const int nwg = (int)((nb + arvx::kOccGroupsPerWg - 1) / arvx::kOccGroupsPerWg);
    // a buffer of its own: this call may run on the exchange stream beside the views / carve of
    // the next job, whose tables live in d_scratch (arvx_ctx_set_exchange_stream).  Growing it
    // frees the old one, which hipFree orders after everything the device has in flight.
    ARVX_HIP(ctx->pool_xscratch.reserve((size_t)(nwg + 1) * sizeof(long long) +
                                        (size_t)nwg * sizeof(int) + 64));
    long long *d_wgoff = (long long *)ctx->pool_xscratch.p;  // nwg + 1 offsets, last = total
    int *d_wgsum = (int *)(d_wgoff + nwg + 1);
    hipLaunchKernelGGL(arvx::occ_classify_kernel, dim3(nwg), dim3(256), 0, ctx->stream,
                       (const unsigned long long *)dev_words, n, (unsigned long long *)dev_packet,
                       d_wgsum);
    hipLaunchKernelGGL(arvx::mc_scan_blocks_kernel, dim3(1), dim3(256), 0, ctx->stream, d_wgsum, nwg,
                       d_wgoff);
    hipLaunchKernelGGL(arvx::occ_write_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0,
                       ctx->stream, (const unsigned long long *)dev_words, n, (long long)cap_words64,
                       d_wgoff, nwg, (unsigned long long *)dev_packet);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

int arvx_occupancy_pack_compress(arvx_ctx *ctx, void *dev_packet, int64_t cap_words64,
                                 void *dev_full_words) {
    ARVX_CHECK_CTX(ctx);
    ExchangeStreamScope exchange_scope(ctx);
    if (!dev_packet || cap_words64 < 0) return fail(ARVX_ERR_INVALID, "bad argument");
    if (((uintptr_t)dev_packet | (uintptr_t)dev_full_words) & 7u)
        return fail(ARVX_ERR_INVALID, "buffers must be 8-byte aligned");
    const size_t plane = (size_t)ctx->X * ctx->Y;
    if (ctx->X % 32 || plane % 64)
        return fail(ARVX_ERR_INVALID, "needs X %% 32 == 0 and X*Y %% 64 == 0 (X=%d, Y=%d)", ctx->X, ctx->Y);
    if (int mrc = need_rec(ctx, true)) return mrc;
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    g.rec = ctx->d_rec;
    const int zl0 = ctx->z0 - ctx->ze0, nz = ctx->z1 - ctx->z0;
    const long long n = (long long)(plane / 64) * nz, nb = (n + 63) / 64;
    if (2 * n >= (1ll << 32)) return fail(ARVX_ERR_INVALID, "slab too large for the fused packet (%lld words)", n);
    const int nwg = (int)((nb + arvx::kOccGroupsPerWg - 1) / arvx::kOccGroupsPerWg);
    arvx::OccGeom og;
    og.wpr = arvx::fast_div((unsigned)(ctx->X / 32));
    og.Y = arvx::fast_div((unsigned)ctx->Y);
    og.P64 = arvx::fast_div((unsigned)(plane / 64));
    // (a buffer of its own: the call may run on the exchange stream beside the next job's views)
    ARVX_HIP(ctx->pool_xscratch.reserve((size_t)(nwg + 1) * sizeof(long long) +
                                        (size_t)nwg * sizeof(int) + 64));
    int *d_wgsum = (int *)((long long *)ctx->pool_xscratch.p + nwg + 1);
    hipLaunchKernelGGL(arvx::occ_pack_classify_kernel, dim3(nwg), dim3(256), 0, ctx->stream, g, og, zl0, n,
                       (unsigned long long *)dev_packet, d_wgsum, (unsigned long long *)dev_full_words);
    hipLaunchKernelGGL(arvx::occ_pack_write_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0,
                       ctx->stream, g, og, zl0, n, (long long)cap_words64, d_wgsum, nwg,
                       (unsigned long long *)dev_packet);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

int arvx_occupancy_expand(arvx_ctx *ctx, const void *dev_packets, int world, int self_rank,
                          int64_t n_words64, int64_t cap_words64, void *dev_full_words,
                          int *dev_overflow) {
    ARVX_CHECK_CTX(ctx);
    ExchangeStreamScope exchange_scope(ctx);
    if (!dev_packets || !dev_full_words || !dev_overflow || world < 1 || self_rank < 0 ||
        self_rank >= world || n_words64 <= 0 || cap_words64 < 0)
        return fail(ARVX_ERR_INVALID, "bad argument");
    if (((uintptr_t)dev_packets & 7u) || ((uintptr_t)dev_full_words & 15u))
        return fail(ARVX_ERR_INVALID, "packets must be 8-byte, the word plane 16-byte aligned");
    const long long n = n_words64, nb = (n + 63) / 64;
    const long long S = arvx::occ_packet_header(n) + cap_words64;
    const long long per = (nb + 2 * arvx::kExpandChunks - 1) / (2 * arvx::kExpandChunks);
    hipLaunchKernelGGL(arvx::occ_expand_kernel, dim3((unsigned)((per * world + 3) / 4)), dim3(256), 0,
                       ctx->stream, (const unsigned long long *)dev_packets, S, world, self_rank, n,
                       (long long)cap_words64, (unsigned long long *)dev_full_words, dev_overflow,
                       0ll);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

int arvx_occupancy_expand_striped(arvx_ctx *ctx, const void *dev_packets, int world,
                                  int64_t n_words64, int64_t cap_words64, int64_t words_per_group,
                                  void *dev_full_words, int *dev_overflow) {
    return arvx_occupancy_expand_striped_others(ctx, dev_packets, world, -1, n_words64, cap_words64,
                                                words_per_group, dev_full_words, dev_overflow);
}

int arvx_occupancy_expand_striped_others(arvx_ctx *ctx, const void *dev_packets, int world,
                                         int self_rank, int64_t n_words64, int64_t cap_words64,
                                         int64_t words_per_group, void *dev_full_words,
                                         int *dev_overflow) {
    ARVX_CHECK_CTX(ctx);
    ExchangeStreamScope exchange_scope(ctx);
    if (self_rank < -1 || self_rank >= world) return fail(ARVX_ERR_INVALID, "self_rank %d of %d", self_rank, world);
    if (!dev_packets || !dev_full_words || !dev_overflow || world < 1 || n_words64 <= 0 ||
        cap_words64 < 0 || words_per_group < 2 || (words_per_group & 1) ||
        n_words64 % words_per_group)
        return fail(ARVX_ERR_INVALID, "bad argument (words per group must be even and divide n)");
    if (((uintptr_t)dev_packets & 7u) || ((uintptr_t)dev_full_words & 15u))
        return fail(ARVX_ERR_INVALID, "packets must be 8-byte, the word plane 16-byte aligned");
    const long long n = n_words64, nb = (n + 63) / 64;
    const long long S = arvx::occ_packet_header(n) + cap_words64;
    const long long per = (nb + 2 * arvx::kExpandChunks - 1) / (2 * arvx::kExpandChunks);
    hipLaunchKernelGGL(arvx::occ_expand_kernel, dim3((unsigned)((per * world + 3) / 4)), dim3(256), 0,
                       ctx->stream, (const unsigned long long *)dev_packets, S, world, self_rank, n,
                       (long long)cap_words64, (unsigned long long *)dev_full_words, dev_overflow,
                       (long long)words_per_group);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

// ---- carve -------------------------------------------------------------------

// Launches the carve over planes [ze0, ze1) (owned + halo) of the records at `rec`.
#ifdef ARVX_EXPERIMENTS
// A fresh model carved by ONE persistent launch (carve_stream_kernels.h).  p: geometry, views and
// flags filled in by launch_carve.
static int launch_carve_stream(Ctx *ctx, arvx::CarveParams p, int ncu) {
    const size_t ncoarse = (size_t)p.coarseX * p.coarseY * p.coarseZ;
    const unsigned G = (unsigned)ncu * 4u;  // resident workgroups (128 VGPRs: 4 per compute unit)
    // (the chunks of 64 views are a template parameter of the kernel: 1, or all four)
    if (p.nchunks > 1) p.nchunks = arvx::kMaxChunks;
    // a coarse unit (one wave): up to four coarse tiles, so that every wave gets one
    p.cwA = (int)std::min<size_t>(arvx::kStreamTilesA, std::max<size_t>(1, (ncoarse + 4 * G - 1) / (4 * G)));
    p.nA = (int)((ncoarse + p.cwA - 1) / p.cwA);
    // a sub-tile unit (one wave): 16 sub-tiles, a quarter of a coarse tile (half on striped slabs)
    p.splitLog2 = p.cyShift + p.czShift - 2;
    p.listStride = 1 + 2 * p.nchunks;
    p.itemStride = 2 + 4 * p.nchunks;
    // A unit u appends to list part u % 8: a part gets at most every 8th unit's tiles
    p.listCap = (p.nA / arvx::kStreamLists + 1) * p.cwA;
    // sub-tile i (of 4 per tile) goes to list i % 8 of one of eight weight classes: a list never
    // gets more than every 8th sub-tile
    p.workCap = (int)(((size_t)p.tilesX * p.tilesY * p.tilesZ * 4 + 7) / 8);
    const size_t off_list = ((size_t)arvx::kStreamLines * arvx::kCounterStride * sizeof(int) + 255) / 256 * 256;
    const size_t off_items =
        off_list + ((size_t)arvx::kStreamLists * p.listCap * p.listStride * 8 + 255) / 256 * 256;
    const size_t need = off_items + (size_t)arvx::kWorkLists * p.workCap * p.itemStride * 8 + 256;
    if (ctx->stream_bytes < need) {
        if (ctx->d_stream) (void)hipFree(ctx->d_stream);
        ctx->d_stream = nullptr;
        ctx->stream_bytes = 0;
        ctx->stream_layout = 0;
        ARVX_HIP(hipMalloc(&ctx->d_stream, need));
        ctx->stream_bytes = need;
        // (stale granules are told by their tag; but the memory must not hold a FUTURE tag)
        ARVX_HIP(hipMemsetAsync(ctx->d_stream, 0, need, ctx->stream));
        ctx->carve_epoch = 0;
    }
    if (ctx->stream_layout != need) {  // the control block: all zero before a launch
        ARVX_HIP(hipMemsetAsync(ctx->d_stream, 0, off_list, ctx->stream));
        ctx->stream_layout = need;
    }
    if (++ctx->carve_epoch == 0u) {  // (2^32 launches: start the tags over on clean memory)
        ARVX_HIP(hipMemsetAsync(ctx->d_stream, 0, need, ctx->stream));
        ctx->carve_epoch = 1u;
    }
    p.sctl = (int *)ctx->d_stream;
    p.poolNext = p.sctl + (size_t)arvx::kSC_Pool * arvx::kCounterStride;
    p.listG = (unsigned long long *)((uint8_t *)ctx->d_stream + off_list);
    p.itemG = (unsigned long long *)((uint8_t *)ctx->d_stream + off_items);
    p.epoch = ctx->carve_epoch;
    p.fault = ctx->d_fault;
    p.nwaves = (int)G * 4;
    ARVX_HIP(ctx->pool_ccode.reserve(ncoarse + 64));
    ARVX_HIP(ctx->pool_cstate.reserve(ncoarse + 64));
    p.coarseCarved = (uint8_t *)ctx->pool_ccode.p;
    p.cstate = (uint8_t *)ctx->pool_cstate.p;
    p.flags |= 128u;
    ctx->cstate_tiles = 0;
#ifdef ARVX_TIMELINE
    if (ctx->d_timeline) (void)hipFree(ctx->d_timeline);
    ctx->d_timeline = nullptr;
    ctx->timeline_n = (int64_t)G * 4;  // one record per wave
    ctx->timeline_rec = 128;
    ARVX_HIP(hipMalloc(&ctx->d_timeline, (size_t)G * 4 * 128));
    ARVX_HIP(hipMemsetAsync(ctx->d_timeline, 0, (size_t)G * 4 * 128, ctx->stream));
    p.timeline = (unsigned long long *)ctx->d_timeline;
#endif
    // (the chunks of 64 views are a template parameter: up to 64 views, or up to 256)
    const bool left = ctx->assoc == ARVX_ASSOC_LEFT, one = p.nchunks == 1;
    if (left && one)
        hipLaunchKernelGGL((arvx::carve_stream_kernel<true, 1>), dim3(G), dim3(256), 0, ctx->stream, p);
    else if (left)
        hipLaunchKernelGGL((arvx::carve_stream_kernel<true, arvx::kMaxChunks>), dim3(G), dim3(256), 0, ctx->stream, p);
    else if (one)
        hipLaunchKernelGGL((arvx::carve_stream_kernel<false, 1>), dim3(G), dim3(256), 0, ctx->stream, p);
    else
        hipLaunchKernelGGL((arvx::carve_stream_kernel<false, arvx::kMaxChunks>), dim3(G), dim3(256), 0, ctx->stream, p);
    ARVX_HIP(hipGetLastError());
    ctx->lazy = true;
    ctx->cstate_tiles = ncoarse;
    return ARVX_OK;
}
#endif

// `fresh`: the model is all-occupied/unseen and exists only as that flag: nothing is read,
// every record of the grid is written.
// `foreign_code` (records that are not the context's own, fresh): the decided coarse tiles are not
// written there either -- their codes go to that array (ncoarse bytes) and its reader takes them
// from it (arvx_fast_carve).
static int launch_carve(Ctx *ctx, uint16_t *rec, int first, int count, unsigned flags,
                        bool fresh, uint8_t *foreign_code = nullptr) {
    arvx::CarveParams p;
    carve_geometry(ctx, p);
    p.rec = rec;
    p.M = ctx->d_M;
    p.bg = ctx->d_bg;
    p.sat = ctx->d_sat;
    p.stats = ctx->d_stats;
    p.W = ctx->W;
    p.H = ctx->H;
    p.bgWords = ctx->bgWords;
    p.satStride = ctx->satStride;
    p.satW = ctx->satW;
    p.v0 = first;
    p.v1 = first + count;
    p.flags = (flags & 3u) | (fresh ? 4u : 0u);
    p.ccode = nullptr;  // (the carve reads records only where it has written them)
    p.cstate = nullptr;
    p.nchunks = (count + 63) / 64;
    if (flags & ARVX_CARVE_STATS) ARVX_HIP(hipMemsetAsync(ctx->d_stats, 0, 64, ctx->stream));
    // rows of tiles (along x) are dealt to the XCDs cyclically: see carve_fused_kernel
    const size_t rows8 = ((size_t)p.tilesY * p.tilesZ + 7) / 8 * 8;
    const unsigned grid = (unsigned)(rows8 * p.tilesX);
    const bool cull = !(flags & ARVX_CARVE_NO_CULL);
    const bool split = cull && !(flags & ARVX_CARVE_FUSED) && p.nchunks <= arvx::kMaxChunks;
    static const bool two_launches = experiment_flag("ARVX_COARSE_SPLIT");
    const size_t ncoarse = (size_t)p.coarseX * p.coarseY * p.coarseZ;
    if (ctx->ncu <= 0) {
        ctx->ncu = 256;
        (void)hipDeviceGetAttribute(&ctx->ncu, hipDeviceAttributeMultiprocessorCount, ctx->device);
        if (ctx->ncu <= 0) ctx->ncu = 256;
    }
    const int ncu = ctx->ncu;
    size_t layout_when_done = 0;
    bool lazy = false;
    if (rec == ctx->d_rec) ctx->lazy = false;  // (set again below once the launches are out)
    // a fresh model, the context's own records, up to 256 views: one persistent launch where the
    // caller asks for it (carve_stream_kernels.h; EXPERIMENTS.md, round 4: its sub-tile phase is
    // slower inside a launch of 128-register waves than as a launch of its own, so the three
    // launches below stay the default)
#ifdef ARVX_EXPERIMENTS
    static const bool stream_default = experiment_flag("ARVX_STREAM_DEFAULT");
    if (fresh && split && rec == ctx->d_rec && !(flags & (ARVX_CARVE_STATS | ARVX_CARVE_NO_STREAM)) &&
        ((flags & ARVX_CARVE_STREAM) || (stream_default && (size_t)p.X * p.Y * p.Z >= ((size_t)1 << 26))) &&
        arvx::rec_count(p) < ((size_t)1 << 30) && p.tilesX < 65536 && p.tilesY < 65536 && p.tilesZ < 65536)
        return launch_carve_stream(ctx, p, ncu);
#endif
    if (cull) {
        const size_t words = ncoarse * p.nchunks;
        // coarse masks | coarse codes | undecided list | two list counters
        const size_t off_list = (2 * words * sizeof(unsigned long long) + ncoarse + 255) / 256 * 256;
        const size_t off_work = off_list + ((ncoarse + 2 * 64) * sizeof(int) + 255) / 256 * 256;
        const size_t nctr = (size_t)arvx::kWorkLists * arvx::kCounterStride;
        // sub-tile i (of 4 per tile) goes to list i % 8 of one of eight weight classes: a
        // list never gets more than every 8th sub-tile
        const size_t cap = ((size_t)p.tilesX * p.tilesY * p.tilesZ * 4 + 7) / 8;
        const size_t nitems = cap * arvx::kWorkLists;
        // one persistent workgroup per workgroup slot of the chip (4 per CU at 128 VGPRs)
        static const int exact_wgs = experiment_int("ARVX_EXACT_WGS_PER_CU");  // (A/B builds)
        // (the small grids' instantiation -- items shared between waves -- holds 149 registers, 3 per CU;
        // launched 4 per CU all the same: the fourth takes over as the first ends, C1 / C2 3 / 9 %
        // faster than with 3, EXPERIMENTS.md round 5)
        const unsigned pgrid = (unsigned)ncu * (exact_wgs > 0 ? (unsigned)exact_wgs : (unsigned)ARVX_EXACT_WAVES_PER_SIMD);
        const size_t nwaves = (size_t)pgrid * 4;
        const size_t nctr_pool = (size_t)arvx::kPoolCounters * arvx::kCounterStride;
        const size_t ints = nctr + nctr_pool;  // list fill counters, pool ticket counters
        const size_t off_items = off_work + (ints * sizeof(int) + 255) / 256 * 256;
        const size_t need =
            off_items + nitems * (1 + 2 * (size_t)p.nchunks) * sizeof(unsigned long long) + 64;
        if (ctx->coarse_bytes < need) {
            if (ctx->d_coarse) (void)hipFree(ctx->d_coarse);
            ctx->d_coarse = nullptr;
            ctx->coarse_bytes = 0;
            ARVX_HIP(hipMalloc(&ctx->d_coarse, need));
            ctx->coarse_bytes = need;
        }
        p.coarseMixed = (unsigned long long *)ctx->d_coarse;
        p.coarseFg = p.coarseMixed + words;
        p.coarseCarved = (uint8_t *)(p.coarseFg + words);
        // Lazy state: a fresh model carved by the split launch into the context's own records
        // does not write the coarse tiles it settles as a whole -- their code, kept in the
        // context, is their state (arvx_device.h; need_rec writes them out for the stages that
        // want records).
#ifndef ARVX_NO_LAZY  // (A/B builds: always write the decided tiles)
        lazy = fresh && split && (rec == ctx->d_rec || foreign_code);
#endif
        if (lazy && rec == ctx->d_rec) {
            ARVX_HIP(ctx->pool_ccode.reserve(ncoarse + 64));
            p.coarseCarved = (uint8_t *)ctx->pool_ccode.p;
            p.flags |= 128u;
        } else if (lazy) {
            p.coarseCarved = foreign_code;
            p.flags |= 128u;
        }
        // the model's own records: what this and earlier carves settle for whole coarse tiles
        // is remembered (CarveParams::cstate); a fresh carve rewrites every entry
        if (split && rec == ctx->d_rec) {
            ARVX_HIP(ctx->pool_cstate.reserve(ncoarse + 64));
            if (!fresh && ctx->cstate_tiles != ncoarse)
                ARVX_HIP(hipMemsetAsync(ctx->pool_cstate.p, 0, ncoarse, ctx->stream));
            p.cstate = (uint8_t *)ctx->pool_cstate.p;
            ctx->cstate_tiles = 0;  // (valid again once the launches are out)
        }
        if (split) {
            int *lst = (int *)((uint8_t *)ctx->d_coarse + off_list);
            // two counters, 64 ints apart, used alternately (carve_coarse_kernel): both zero
            // before the first carve that uses this layout
            if (ctx->carve_layout != need) {
                ARVX_HIP(hipMemsetAsync(lst, 0, 2 * 64 * sizeof(int), ctx->stream));
                ctx->carve_layout = need;
                ctx->carve_seq = 0;
            }
            p.undecidedCount = lst + 64 * (ctx->carve_seq & 1);
            p.undecidedCountNext = lst + 64 * ((ctx->carve_seq + 1) & 1);
            p.undecidedList = lst + 2 * 64;
            // (carve_seq advances once all launches of this carve are enqueued; a failure in
            // between leaves the layout unset, so that the next carve zeroes both counters)
            ctx->carve_layout = 0;
            layout_when_done = need;
            int *base = (int *)((uint8_t *)ctx->d_coarse + off_work);
            p.workCount = base;
            p.poolNext = base + nctr;
            p.nwaves = (int)nwaves;
            p.workCap = (int)cap;
            p.itemInfo = (unsigned long long *)((uint8_t *)ctx->d_coarse + off_items);
            p.itemMasks = p.itemInfo + nitems;
            // (carve_coarse_kernel zeroes the `ints` counters at base)
        }
        if (!split || two_launches || lazy) {
            // (lazy: one WAVE per coarse tile classifies it and lists it if it is undecided;
            // no workgroup per tile is needed when nothing is filled)
            // (16 tiles per workgroup where that still leaves a workgroup per compute unit:
            // fewer appends to the list's counter; 4 on small grids)
            if (p.nchunks > arvx::kMaxChunks) {  // (only the fused kernel takes that many views)
                hipLaunchKernelGGL(arvx::carve_coarse_wave_kernel, dim3((unsigned)((ncoarse + 3) / 4)),
                                   dim3(256), 0, ctx->stream, p);
                ARVX_HIP(hipGetLastError());
            } else {
                const int cw = ncoarse >= (size_t)16 * ncu ? arvx::kCoarseWaves : 4;
                p.coarsePerWg = cw;
                // (tile, view) pairs dealt to the lanes densely: as many waves as the pairs need
                const size_t pairs = (size_t)cw * (size_t)(p.v1 - p.v0);
                const unsigned threads =
                    (unsigned)std::min<size_t>(64 * arvx::kCoarseWaves, (pairs + 63) / 64 * 64);
                hipLaunchKernelGGL(arvx::carve_coarse_kernel,
                                   dim3((unsigned)((ncoarse + cw - 1) / cw)),
                                   dim3(std::max(threads, 64u)), 0, ctx->stream, p);
                ARVX_HIP(hipGetLastError());
            }
        }
    }
    // the statistics counters live in the row-mapped variant
    static const bool row_map = experiment_flag("ARVX_EXACT_ROWS");
    const bool blocks = split && !row_map && !(flags & ARVX_CARVE_STATS);
    // few sub-tiles per wave (small grids, slabs): the exact kernel may hand an item's views
    // to several waves (decided in the kernel from the length of the work lists)
    static const bool no_item_split = experiment_flag("ARVX_NO_ITEM_SPLIT");
    // (up to 2^26 voxels: above that there are more items than half the waves and the kernel
    // never splits; ARVX_ITEM_SPLIT_LOG2 in an experiment build moves the limit)
    static const int split_log2 = experiment_int("ARVX_ITEM_SPLIT_LOG2");
    if (blocks && !no_item_split &&
        (size_t)p.X * p.Y * p.Z <= ((size_t)1 << (split_log2 > 0 ? split_log2 : 26)))
        p.flags |= 8u;
    static const bool force_split = experiment_flag("ARVX_FORCE_SPLIT2");  // every item in 2 parts
    if (blocks && force_split) p.flags |= 8u | 64u;
    static const bool no_block_tests = experiment_flag("ARVX_NO_BLOCK_TESTS");
    if (no_block_tests) p.flags |= 32u;
    if (split) {
        // coarse tiles: classified, and the decided ones written as constant records, one
        // workgroup each
        if (lazy) {
            // (carve_coarse_kernel above did everything)
        } else if (two_launches)
            hipLaunchKernelGGL(arvx::carve_fill_kernel, dim3((unsigned)ncoarse), dim3(256), 0,
                               ctx->stream, p);
        else
            hipLaunchKernelGGL(arvx::carve_coarse_fill_kernel, dim3((unsigned)ncoarse), dim3(256),
                               0, ctx->stream, p);
        ARVX_HIP(hipGetLastError());
        // the others: a fixed grid walks the list (8 waves per SIMD)
        static const int cgrid_env = experiment_int("ARVX_CLASSIFY_WGS");
        const unsigned cgrid = cgrid_env > 0 ? (unsigned)cgrid_env : (unsigned)ncu * 8u;
#ifndef ARVX_CLASSIFY_SPARSE  // (A/B builds: the wave-per-sub-tile kernel everywhere)
        // (the statistics counters live in the other kernel; below 2^26 voxels there are too few
        // listed coarse tiles for a workgroup each: 256^3 +2.5 % with the dense kernel, 512^3
        // -1.5 %, 768^3 -5 %, 1024^3 -11 %)
        const bool dense = !(flags & ARVX_CARVE_STATS) && (size_t)p.X * p.Y * p.Z >= ((size_t)1 << 26);
#else
        const bool dense = false;
#endif
        if (dense)  // one workgroup per listed coarse tile and turn
            hipLaunchKernelGGL(arvx::carve_classify_dense_kernel, dim3((unsigned)ncu * (unsigned)ARVX_DENSE_WGS_PER_CU), dim3(256),
                               0, ctx->stream, p);
        else
            hipLaunchKernelGGL(arvx::carve_classify_kernel, dim3(cgrid), dim3(256), 0, ctx->stream, p);
        ARVX_HIP(hipGetLastError());
        const unsigned pgrid = (unsigned)(p.nwaves / 4);
#ifdef ARVX_TIMELINE
        if (ctx->d_timeline) (void)hipFree(ctx->d_timeline);
        ctx->d_timeline = nullptr;
        ctx->timeline_n = (int64_t)pgrid * 4;  // one record per WAVE of the persistent kernel
        ctx->timeline_rec = 64;
        ARVX_HIP(hipMalloc(&ctx->d_timeline, (size_t)pgrid * 4 * 64));
        ARVX_HIP(hipMemsetAsync(ctx->d_timeline, 0, (size_t)pgrid * 4 * 64, ctx->stream));
        p.timeline = (unsigned long long *)ctx->d_timeline;
#endif
        const bool left = ctx->assoc == ARVX_ASSOC_LEFT;
        const bool may_split = p.flags & 8u;
#ifdef ARVX_EXPERIMENTS
        // the fp32 filter in front of the exact projection (carve_kernels.h, filtered_view_blocks):
        // for callers whose masks leave most blocks of an item to be projected (ARVX_CARVE_FILTER)
        const bool filter = (flags & ARVX_CARVE_FILTER) != 0;
        if (blocks && filter && left && may_split)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<true, true, false, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks && filter && left && fresh)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<true, false, true, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks && filter && left)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<true, false, false, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks && filter && may_split)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<false, true, false, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks && filter && fresh)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<false, false, true, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks && filter)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<false, false, false, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else
#endif
        if (blocks && left && may_split)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<true, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks && left && fresh)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<true, false, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks && left)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<true, false>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks && may_split)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<false, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks && fresh)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<false, false, true>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (blocks)
            hipLaunchKernelGGL((arvx::carve_exact_blocks_kernel<false, false>), dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else if (left)
            hipLaunchKernelGGL(arvx::carve_exact_kernel<true>, dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        else
            hipLaunchKernelGGL(arvx::carve_exact_kernel<false>, dim3(pgrid), dim3(256), 0,
                               ctx->stream, p);
        ARVX_HIP(hipGetLastError());
        ctx->carve_layout = layout_when_done;
        ++ctx->carve_seq;
        if (lazy && rec == ctx->d_rec) ctx->lazy = true;
        if (p.cstate) ctx->cstate_tiles = ncoarse;
        return ARVX_OK;
    }
#ifdef ARVX_TIMELINE
    if (ctx->d_timeline) (void)hipFree(ctx->d_timeline);
    ctx->d_timeline = nullptr;
    ctx->timeline_n = grid;
    ctx->timeline_rec = 32;
    ARVX_HIP(hipMalloc(&ctx->d_timeline, (size_t)grid * 32));
    ARVX_HIP(hipMemsetAsync(ctx->d_timeline, 0, (size_t)grid * 32, ctx->stream));
    p.timeline = (unsigned long long *)ctx->d_timeline;
#endif
    if (ctx->assoc == ARVX_ASSOC_LEFT)
        hipLaunchKernelGGL(arvx::carve_fused_kernel<true>, dim3(grid), dim3(256), 0, ctx->stream, p);
    else
        hipLaunchKernelGGL(arvx::carve_fused_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream, p);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}

int arvx_carve_views(arvx_ctx *ctx, int first, int count, unsigned flags) {
    ARVX_CHECK_CTX(ctx);
    if (!ctx->views_ready) return fail(ARVX_ERR_STATE, "arvx_set_views has not been called");
    if (first < 0 || count < 0 || first + count > ctx->V)
        return fail(ARVX_ERR_INVALID, "view range [%d,%d) outside [0,%d)", first, first + count,
                    ctx->V);
    if (count == 0) return ARVX_OK;
#ifndef ARVX_EXPERIMENTS
    if (flags & ARVX_CARVE_STREAM)  // (refused before anything about the context changes)
        return fail(ARVX_ERR_INVALID, "ARVX_CARVE_STREAM: the one-launch carve is only in -DARVX_EXPERIMENTS "
                                      "builds (libarvx_experiments.so)");
    if (flags & ARVX_CARVE_FILTER)
        return fail(ARVX_ERR_INVALID, "ARVX_CARVE_FILTER: the fp32 projection filter is only in "
                                      "-DARVX_EXPERIMENTS builds (libarvx_experiments.so)");
#endif
    state_changes(ctx, false);
    const bool fresh = ctx->fresh_pending;
    if (fresh) {  // the kernels write every record of the grid; nothing is read
        void *buf = ctx->d_rec;
        if (int rc = ensure_records(ctx, &buf, &ctx->rec_bytes)) return rc;
        ctx->d_rec = (uint16_t *)buf;
    } else if (int rc = need_rec(ctx)) {
        return rc;
    }
    ctx->fresh_pending = false;
    ctx->rec_valid = true;
    return launch_carve(ctx, ctx->d_rec, first, count, flags, fresh);
}

int arvx_carve(arvx_ctx *ctx, unsigned flags) {
    if (!ctx) return fail(ARVX_ERR_INVALID, "null context");
    return arvx_carve_views(ctx, 0, ctx->V, flags);
}

static int surf_host(Ctx *ctx);
static int clo_host(Ctx *ctx);
static void owned_part(const Ctx *ctx, const std::vector<int> &idx, size_t &lo, size_t &hi,
                       long long &base);
// a context without halo planes owns every entry of its lists: their lengths need no host copy
static bool owns_all_planes(const Ctx *ctx) { return ctx->z0 == ctx->ze0 && ctx->z1 == ctx->ze1; }

int arvx_get_stats(arvx_ctx *ctx, arvx_stats *out) {
    ARVX_CHECK_CTX(ctx);
    if (!out) return fail(ARVX_ERR_INVALID, "null out");
    unsigned long long h[8];
    ARVX_HIP(hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    ARVX_SYNC(ctx);
    memset(out, 0, sizeof *out);
    out->subtiles = h[0];
    out->subtiles_carved = h[1];
    out->subtile_views_mixed = h[2];
    out->subtile_views_total = h[3];
    out->surface_voxels = 0;
    if (ctx->color_ready && owns_all_planes(ctx)) {
        out->surface_voxels = (unsigned long long)ctx->surf_count;
    } else if (ctx->color_ready) {  // the owned part of the colour pass's list (arvx_color)
        if (int rc = surf_host(ctx)) return rc;
        size_t lo, hi;
        long long base;
        owned_part(ctx, ctx->h_surf_index, lo, hi, base);
        out->surface_voxels = hi - lo;
    }
    out->reserved[0] = h[5];  // 256-voxel slices evaluated exactly
    out->reserved[1] = h[6];  // voxels among them that were not yet carved+seen
    out->reserved[2] = h[7];  // mixed pairs whose open voxels all got the same answer
    out->host_total_fallbacks = ctx->host_total_fallbacks;
    return ARVX_OK;
}

// Which planes (global z, [lo, hi)) a context's stages work on.  Colours: every plane whose two
// neighbour planes the records hold (a voxel is on the surface or not by its six neighbours).
// Closure with radius r: every plane whose 2 r neighbour planes have colours.  A whole-grid
// context and the sides of a slab that touch the grid's faces lose nothing.
static void stage_ranges(const Ctx *ctx, int radius, int &c_lo, int &c_hi, int &f_lo, int &f_hi) {
    c_lo = ctx->ze0 > 0 ? ctx->ze0 + 1 : 0;
    c_hi = ctx->ze1 < ctx->Z ? ctx->ze1 - 1 : ctx->Z;
    f_lo = c_lo > 0 ? c_lo + radius : 0;
    f_hi = c_hi < ctx->Z ? c_hi - radius : ctx->Z;
}
// the entries [lo, hi) of an ascending index list over the context's planes that lie in the
// owned planes; base = what to subtract to number them over the owned planes
static void owned_part(const Ctx *ctx, const std::vector<int> &idx, size_t &lo, size_t &hi,
                       long long &base) {
    const long long plane = (long long)ctx->X * ctx->Y;
    base = plane * (ctx->z0 - ctx->ze0);
    const long long end = plane * (ctx->z1 - ctx->ze0);
    lo = (size_t)(std::lower_bound(idx.begin(), idx.end(), base,
                                   [](int a, long long b) { return (long long)a < b; }) - idx.begin());
    hi = (size_t)(std::lower_bound(idx.begin(), idx.end(), end,
                                   [](int a, long long b) { return (long long)a < b; }) - idx.begin());
}

// ---- colour pass -----------------------------------------------------------------

int arvx_set_images(arvx_ctx *ctx, const uint8_t *const *images, size_t stride) {
    ARVX_CHECK_CTX(ctx);
    if (!ctx->cameras_ready) return fail(ARVX_ERR_STATE, "arvx_set_views must come first");
    if (!images) return fail(ARVX_ERR_INVALID, "null images");
    const size_t rowb = (size_t)ctx->W * 3;
    if (stride < rowb) return fail(ARVX_ERR_INVALID, "stride %zu < W*3", stride);
    for (int i = 0; i < ctx->V; ++i)
        if (!images[i]) return fail(ARVX_ERR_INVALID, "null image %d", i);
    const size_t img = rowb * ctx->H;
    if (!ctx->d_images) ARVX_HIP(hipMalloc(&ctx->d_images, img * ctx->V));
    bool packed = stride == rowb;
    for (int i = 1; i < ctx->V && packed; ++i) packed = images[i] == images[i - 1] + img;
    if (packed) {
        ARVX_HIP(hipMemcpyAsync(ctx->d_images, images[0], img * ctx->V, hipMemcpyHostToDevice,
                                ctx->stream));
    } else {
        for (int i = 0; i < ctx->V; ++i)
            ARVX_HIP(hipMemcpy2DAsync(ctx->d_images + img * i, rowb, images[i], stride, rowb,
                                      ctx->H, hipMemcpyHostToDevice, ctx->stream));
    }
    ARVX_SYNC(ctx);
    ctx->images_ready = true;
    ctx->color_ready = false;
    ctx->closure_ready = false;
    return ARVX_OK;
}

int arvx_color(arvx_ctx *ctx, int mode) {
    ARVX_CHECK_CTX(ctx);
    if (!ctx->cameras_ready) return fail(ARVX_ERR_STATE, "arvx_set_views has not been called");
    if (!ctx->images_ready) return fail(ARVX_ERR_STATE, "arvx_set_images has not been called");
    if (!ctx->has_campos) return fail(ARVX_ERR_STATE, "arvx_set_views was given no campos");
    if (mode != ARVX_COLOR_CLOSEST && mode != ARVX_COLOR_AVERAGE)
        return fail(ARVX_ERR_INVALID, "colour mode %d", mode);
    if (ctx->stripe_world > 1)
        return fail(ARVX_ERR_STATE, "the colour pass needs contiguous slabs (neighbour planes)");
    ctx->free_surface();
    // surface = occupied and not inner, on bit planes over the context's planes; colours are
    // voted for the planes [c_lo, c_hi) (stage_ranges): the owned ones and the halo planes
    // whose own neighbours the records hold
    const int XW = (ctx->X + 63) / 64;
    const int Zext = ctx->ze1 - ctx->ze0;
    int c_lo, c_hi, f_lo, f_hi;
    stage_ranges(ctx, 0, c_lo, c_hi, f_lo, f_hi);
    const arvx::BitGrid gext{ctx->X, ctx->Y, Zext, XW};
    const size_t row_words = (size_t)XW * ctx->Y;
    const size_t nw_ext = row_words * Zext;
    // the surface plane stays with the context: with its ranks it is the index of the colour
    // list (closure and mesh look colours up through it)
    ARVX_HIP(ctx->pool_col_bits.reserve(nw_ext * sizeof(unsigned long long)));
    ARVX_HIP(ctx->pool_col_rank.reserve(nw_ext * sizeof(arvx::SparseWord)));
    // ... and so do the occupancy plane and the plane of the voxels no view has seen: a closure that
    // follows -- with nothing but handleUnseen in between, src/main.cpp:282-299 -- starts from them
    // instead of converting the records again (state_planes)
    ARVX_HIP(ctx->pool_state_planes.reserve(2 * nw_ext * sizeof(unsigned long long)));  // (one allocation)
    unsigned long long *d_occ = (unsigned long long *)ctx->pool_state_planes.p;
    unsigned long long *d_surf = (unsigned long long *)ctx->pool_col_bits.p;
    ctx->planes_ok = false;
    if (int rc = launch_bit_pack(ctx, gext, 0, 1, d_occ, d_occ + nw_ext)) return rc;
    ctx->planes_words = nw_ext;
    ctx->planes_ok = true;
    ctx->planes_seq = ctx->state_seq;
    ctx->planes_unseen = false;
    // the surface plane of the planes [c_lo, c_hi) (zeros elsewhere) and, per chunk of the
    // compaction, its number of set bits
    int *d_counts = nullptr;
    if (int rc = chunk_counts(ctx, nw_ext, &d_counts)) return rc;
    hipLaunchKernelGGL(arvx::bit_surface_count_kernel, dim3((unsigned)((nw_ext + 255) / 256)), dim3(256), 0,
                       ctx->stream, d_occ, gext, c_lo - ctx->ze0, c_hi - ctx->ze0, d_surf, d_counts);
    ARVX_HIP(hipGetLastError());
    // The list's length is not known before the compaction has run: the buffers are sized for what
    // the last pass needed (first call: a surface's share of the voxels), the kernels stop at that
    // capacity, and the true length is read at the call's ONE synchronisation; a list that outgrew
    // its buffers is compacted and voted again with room for all of it.
    long long cap = (long long)(ctx->pool_surf_index.cap / sizeof(int));
    if (cap <= 0) {
        const double v = (double)ctx->X * ctx->Y * Zext;
        cap = (long long)std::min<double>(v, 8.0 * std::cbrt(v) * std::cbrt(v) + 4096.0);
    }
    long long total = 0;
    for (int attempt = 0;; ++attempt) {
        ARVX_HIP(ctx->pool_surf_index.reserve((size_t)cap * sizeof(int)));
        ctx->d_surf_index = (int *)ctx->pool_surf_index.p;
        ARVX_HIP(ctx->pool_surf_rgb.reserve((size_t)cap * sizeof(float4)));
        ctx->d_surf_rgba = (float4 *)ctx->pool_surf_rgb.p;
        ARVX_HIP(ctx->pool_surf_depth.reserve((size_t)cap * sizeof(float)));
        ctx->d_surf_depth = (float *)ctx->pool_surf_depth.p;
        ARVX_HIP(ctx->pool_surf_has.reserve((size_t)cap));
        ctx->d_surf_has = (uint8_t *)ctx->pool_surf_has.p;
        const long long *d_total = nullptr;
        if (int rc = bit_compact(ctx, d_surf, nw_ext, gext, cap, ctx->d_surf_index,
                                 (arvx::SparseWord *)ctx->pool_col_rank.p, 0, &d_total))
            return rc;
        arvx::VoteParams vp;
        vp.index = ctx->d_surf_index;
        vp.n = cap;
        vp.n_dev = d_total;
        vp.X = ctx->X;
        vp.Y = ctx->Y;
        vp.zglob0 = ctx->ze0;  // (list indices run over the context's planes)
        vp.s = ctx->s;
        vp.V = ctx->V;
        vp.W = ctx->W;
        vp.H = ctx->H;
        vp.M = ctx->d_M;
        vp.campos = ctx->d_campos;
        vp.images = ctx->d_images;
        vp.mode = mode;
        vp.rgba = ctx->d_surf_rgba;
        vp.depth = ctx->d_surf_depth;
        vp.has = ctx->d_surf_has;
        if (ctx->assoc == ARVX_ASSOC_LEFT)
            hipLaunchKernelGGL(arvx::color_vote_kernel<true>, dim3((unsigned)((cap + 255) / 256)),
                               dim3(256), 0, ctx->stream, vp);
        else
            hipLaunchKernelGGL(arvx::color_vote_kernel<false>, dim3((unsigned)((cap + 255) / 256)),
                               dim3(256), 0, ctx->stream, vp);
        ARVX_HIP(hipGetLastError());
        ARVX_SYNC(ctx);
        total = host_total(ctx, 0);
        if (total < 0) return fail(ARVX_ERR_HIP, "the compaction left no count");
        if (total <= cap || attempt) break;
        cap = total + total / 8;  // (once more, with room for all)
    }
    ctx->surf_count = total;
    ctx->surf_host_count = -1;  // the host copies of the list are fetched when somebody asks
    if (total == 0) {
        ctx->d_surf_index = nullptr;
        ctx->d_surf_rgba = nullptr;
        ctx->d_surf_depth = nullptr;
        ctx->d_surf_has = nullptr;
    }
    ctx->color_ready = true;
    // (arvx_get_stats' surface_voxels: the owned part of the list, counted when it is asked for)
    return ARVX_OK;
}

int arvx_color_samples(arvx_ctx *ctx, int64_t n, const int64_t *index, int views,
                       arvx_color_sample *out) {
    ARVX_CHECK_CTX(ctx);
    static_assert(sizeof(arvx_color_sample) == 8, "r, g, b, valid, depth");
    if (!ctx->cameras_ready) return fail(ARVX_ERR_STATE, "arvx_set_views has not been called");
    if (views != ctx->V)
        return fail(ARVX_ERR_INVALID, "out holds %d samples per voxel, the context has %d views", views,
                    ctx->V);
    if (!ctx->images_ready) return fail(ARVX_ERR_STATE, "arvx_set_images has not been called");
    if (!ctx->has_campos) return fail(ARVX_ERR_STATE, "arvx_set_views was given no campos");
    if (n < 0 || (n && (!index || !out))) return fail(ARVX_ERR_INVALID, "null index / out or n < 0");
    if (n == 0) return ARVX_OK;
    const int64_t nown = (int64_t)ctx->X * ctx->Y * (ctx->z1 - ctx->z0);
    for (int64_t k = 0; k < n; ++k)
        if (index[k] < 0 || index[k] >= nown)
            return fail(ARVX_ERR_INVALID, "voxel index %lld outside [0,%lld)", (long long)index[k],
                        (long long)nown);
    const size_t total = (size_t)n * ctx->V;
    if (int rc = ensure_scratch(ctx, (size_t)n * sizeof(long long) + total * sizeof(uint2) + 64)) return rc;
    uint2 *d_out = (uint2 *)ctx->d_scratch;
    long long *d_idx = (long long *)(d_out + total);
    ARVX_HIP(hipMemcpyAsync(d_idx, index, (size_t)n * sizeof(long long), hipMemcpyHostToDevice,
                            ctx->stream));
    arvx::VoteParams vp{};
    vp.n = n;
    vp.X = ctx->X;
    vp.Y = ctx->Y;
    vp.zglob0 = ctx->z0;  // (indices run over the owned planes)
    vp.s = ctx->s;
    vp.V = ctx->V;
    vp.W = ctx->W;
    vp.H = ctx->H;
    vp.M = ctx->d_M;
    vp.campos = ctx->d_campos;
    vp.images = ctx->d_images;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (ctx->assoc == ARVX_ASSOC_LEFT)
        hipLaunchKernelGGL(arvx::color_samples_kernel<true>, dim3(grid), dim3(256), 0, ctx->stream, vp,
                           d_idx, d_out);
    else
        hipLaunchKernelGGL(arvx::color_samples_kernel<false>, dim3(grid), dim3(256), 0, ctx->stream, vp,
                           d_idx, d_out);
    ARVX_HIP(hipGetLastError());
    ARVX_HIP(hipMemcpyAsync(out, d_out, total * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream));
    ARVX_SYNC(ctx);
    return ARVX_OK;
}

// The colour list's indices and has-flags on the host: fetched when somebody asks (a pipeline
// that goes on to the closure and the mesh on the device never does).
static int surf_host(Ctx *ctx) {
    if (ctx->surf_host_count == ctx->surf_count) return ARVX_OK;
    const size_t n = (size_t)ctx->surf_count;
    ctx->h_surf_index.resize(n);
    ctx->h_surf_has.resize(n);
    if (n) {
        ARVX_HIP(hipMemcpyAsync(ctx->h_surf_index.data(), ctx->d_surf_index, n * sizeof(int),
                                hipMemcpyDeviceToHost, ctx->stream));
        ARVX_HIP(hipMemcpyAsync(ctx->h_surf_has.data(), ctx->d_surf_has, n, hipMemcpyDeviceToHost,
                                ctx->stream));
        ARVX_SYNC(ctx);
    }
    ctx->surf_host_count = ctx->surf_count;
    return ARVX_OK;
}

int arvx_surface_count(arvx_ctx *ctx, int64_t *count) {
    if (!ctx || !count) return fail(ARVX_ERR_INVALID, "null argument");
    if (!ctx->color_ready) return fail(ARVX_ERR_STATE, "no colour result (call arvx_color)");
    ARVX_HIP(hipSetDevice(ctx->device));
    if (int rc = surf_host(ctx)) return rc;
    size_t lo, hi;
    long long base;
    owned_part(ctx, ctx->h_surf_index, lo, hi, base);
    int64_t n = 0;
    for (size_t e = lo; e < hi; ++e) n += ctx->h_surf_has[e];
    *count = n;
    return ARVX_OK;
}

static int surface_fetch(Ctx *ctx, std::vector<float> &rgb, std::vector<float> &depth) {
    rgb.resize((size_t)ctx->surf_count * 4);  // (r, g, b, has) per entry
    depth.resize((size_t)ctx->surf_count);
    if (ctx->surf_count) {
        ARVX_HIP(hipMemcpyAsync(rgb.data(), ctx->d_surf_rgba, rgb.size() * sizeof(float),
                                hipMemcpyDeviceToHost, ctx->stream));
        ARVX_HIP(hipMemcpyAsync(depth.data(), ctx->d_surf_depth, depth.size() * sizeof(float),
                                hipMemcpyDeviceToHost, ctx->stream));
        ARVX_SYNC(ctx);
    }
    return ARVX_OK;
}

int arvx_surface_download(arvx_ctx *ctx, int64_t *index, float *rgb) {
    ARVX_CHECK_CTX(ctx);
    if (!index || !rgb) return fail(ARVX_ERR_INVALID, "null argument");
    if (!ctx->color_ready) return fail(ARVX_ERR_STATE, "no colour result (call arvx_color)");
    std::vector<float> hrgb, hdepth;
    int rc = surface_fetch(ctx, hrgb, hdepth);
    if (rc) return rc;
    if (int rc2 = surf_host(ctx)) return rc2;
    size_t lo, hi;
    long long base;
    owned_part(ctx, ctx->h_surf_index, lo, hi, base);
    size_t k = 0;
    for (size_t e = lo; e < hi; ++e) {
        if (!ctx->h_surf_has[e]) continue;
        index[k] = ctx->h_surf_index[e] - base;
        rgb[3 * k] = hrgb[4 * e];
        rgb[3 * k + 1] = hrgb[4 * e + 1];
        rgb[3 * k + 2] = hrgb[4 * e + 2];
        ++k;
    }
    return ARVX_OK;
}

int arvx_surface_depth_download(arvx_ctx *ctx, float *depth) {
    ARVX_CHECK_CTX(ctx);
    if (!depth) return fail(ARVX_ERR_INVALID, "null argument");
    if (!ctx->color_ready) return fail(ARVX_ERR_STATE, "no colour result (call arvx_color)");
    std::vector<float> hrgb, hdepth;
    int rc = surface_fetch(ctx, hrgb, hdepth);
    if (rc) return rc;
    if (int rc2 = surf_host(ctx)) return rc2;
    size_t lo, hi;
    long long base;
    owned_part(ctx, ctx->h_surf_index, lo, hi, base);
    size_t k = 0;
    for (size_t e = lo; e < hi; ++e)
        if (ctx->h_surf_has[e]) depth[k++] = hdepth[e];
    return ARVX_OK;
}

// Model::voxels of the owned voxels, built on the device in chunks and copied out.
int arvx_export_model(arvx_ctx *ctx, float *rgba, int apply_unseen) {
    ARVX_CHECK_CTX(ctx);
    if (int mrc = need_rec(ctx)) return mrc;
    if (!rgba) return fail(ARVX_ERR_INVALID, "null rgba");
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    g.rec = ctx->d_rec;
    const int zown = ctx->z0 - ctx->ze0;
    if (ctx->closure_ready && (apply_unseen != 0) != (ctx->closure_unseen != 0))
        return fail(ARVX_ERR_STATE, "arvx_closure was computed with apply_unseen=%d",
                    ctx->closure_unseen);
    const size_t chunk = (size_t)1 << 24;  // voxels per chunk: 256 MiB of float4
    const size_t nchunk = std::min(chunk, ctx->nvox);
    float4 *d_out = nullptr;
    ARVX_HIP(hipMalloc(&d_out, nchunk * sizeof(float4)));
    int rc = ARVX_OK;
    for (size_t i0 = 0; i0 < ctx->nvox && rc == ARVX_OK; i0 += chunk) {
        const size_t n = std::min(chunk, ctx->nvox - i0);
        const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 16384);
        hipLaunchKernelGGL(arvx::export_fill_kernel, dim3(grid), dim3(256), 0, ctx->stream, g,
                           zown, i0, n, paint_plane(ctx), d_out, apply_unseen);
        // entries of an ascending list (numbered over the context's planes) in owned voxels
        // [i0, i0 + n)
        const long long own_base = (long long)ctx->X * ctx->Y * zown;
        auto list_range = [&](const std::vector<int> &idx, long long &first, long long &last) {
            auto below = [](int a, long long b) { return (long long)a < b; };
            first = std::lower_bound(idx.begin(), idx.end(), own_base + (long long)i0, below) - idx.begin();
            last = std::lower_bound(idx.begin(), idx.end(), own_base + (long long)(i0 + n), below) -
                   idx.begin();
        };
        if (ctx->color_ready && ctx->surf_count > 0) {
            if (int rc = surf_host(ctx)) return rc;
            long long first, last;
            list_range(ctx->h_surf_index, first, last);
            if (last > first)
                hipLaunchKernelGGL(arvx::export_scatter_kernel,
                                   dim3((unsigned)((last - first + 255) / 256)), dim3(256), 0,
                                   ctx->stream, ctx->d_surf_index, ctx->d_surf_rgba, first, last,
                                   g, zown, paint_plane(ctx), i0, d_out, apply_unseen);
        }
        if (ctx->closure_ready && ctx->clo_count > 0) {
            if (int rc = clo_host(ctx)) return rc;
            long long first, last;
            list_range(ctx->h_clo_index, first, last);
            if (last > first)
                hipLaunchKernelGGL(arvx::export_overlay_kernel,
                                   dim3((unsigned)((last - first + 255) / 256)), dim3(256), 0,
                                   ctx->stream, ctx->d_clo_index, (const float4 *)ctx->d_clo_rgba,
                                   first, last, (size_t)own_base, i0, d_out);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess)
            e = hipMemcpyAsync(rgba + 4 * i0, d_out, n * sizeof(float4), hipMemcpyDeviceToHost,
                               ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = arvx::fail_hip(e, "export chunk", __FILE__, __LINE__);
    }
    (void)hipFree(d_out);
    return rc;
}

int arvx_selftest_divide(arvx_ctx *ctx, int64_t n, const float *a0, const float *a1,
                         const float *b, float *out) {
    ARVX_CHECK_CTX(ctx);
    if (n < 1 || !a0 || !a1 || !b || !out) return fail(ARVX_ERR_INVALID, "bad argument");
    float *d = nullptr;
    ARVX_HIP(hipMalloc(&d, (size_t)n * 7 * sizeof(float)));
    hipError_t e = hipMemcpyAsync(d, a0, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d + n, a1, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d + 2 * n, b, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(arvx::selftest_divide_kernel, dim3((unsigned)((n + 255) / 256)),
                           dim3(256), 0, ctx->stream, d, d + n, d + 2 * n, (size_t)n, d + 3 * n);
        e = hipGetLastError();
    }
    if (e == hipSuccess)
        e = hipMemcpyAsync(out, d + 3 * n, (size_t)n * 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return arvx::fail_hip(e, "selftest_divide", __FILE__, __LINE__);
    return ARVX_OK;
}

int arvx_selftest_project(arvx_ctx *ctx, int64_t n, const float M[12], float voxel_size,
                          const int32_t *xyz, float *rows_uv) {
    ARVX_CHECK_CTX(ctx);
    if (!M) return fail(ARVX_ERR_INVALID, "null M");
    arvx::SelftestMatrix m;
    memcpy(m.m, M, sizeof m.m);
    const unsigned grid = (unsigned)((n + 255) / 256);
    return selftest_xyz(ctx, n, xyz, (size_t)n * 5, rows_uv, [&](int *d_xyz, float *d_out) {
        if (ctx->assoc == ARVX_ASSOC_LEFT)
            hipLaunchKernelGGL(arvx::selftest_project_kernel<true>, dim3(grid), dim3(256), 0,
                               ctx->stream, m, voxel_size, d_xyz, (long long)n, d_out, d_out + 3 * n);
        else
            hipLaunchKernelGGL(arvx::selftest_project_kernel<false>, dim3(grid), dim3(256), 0,
                               ctx->stream, m, voxel_size, d_xyz, (long long)n, d_out, d_out + 3 * n);
    });
}

int arvx_selftest_depth(arvx_ctx *ctx, int64_t n, const float campos[3], float voxel_size,
                        const int32_t *xyz, float *depth) {
    ARVX_CHECK_CTX(ctx);
    if (!campos) return fail(ARVX_ERR_INVALID, "null campos");
    arvx::SelftestMatrix m;
    memset(&m, 0, sizeof m);
    memcpy(m.m, campos, 3 * sizeof(float));
    return selftest_xyz(ctx, n, xyz, (size_t)n, depth, [&](int *d_xyz, float *d_out) {
        hipLaunchKernelGGL(arvx::selftest_depth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256),
                           0, ctx->stream, m, voxel_size, d_xyz, (long long)n, d_out);
    });
}

int arvx_selftest_view_tables(arvx_ctx *ctx, int view, uint32_t *bg_bits, uint16_t *table, int *ld) {
    ARVX_CHECK_CTX(ctx);
    if (!ctx->views_ready) return fail(ARVX_ERR_STATE, "arvx_set_views has not been called");
    if (view < 0 || view >= ctx->V) return fail(ARVX_ERR_INVALID, "view %d outside [0,%d)", view, ctx->V);
    if (!ld) return fail(ARVX_ERR_INVALID, "null ld");
    *ld = ctx->satW;
    const size_t words = ((size_t)ctx->W * ctx->H + 31) / 32;
    if (bg_bits)
        ARVX_HIP(hipMemcpyAsync(bg_bits, ctx->d_bg + (size_t)view * ctx->bgWords, words * sizeof(uint32_t),
                                hipMemcpyDeviceToHost, ctx->stream));
    if (table)
        ARVX_HIP(hipMemcpyAsync(table, ctx->d_sat + (size_t)view * ctx->satStride,
                                (size_t)ctx->satStride * sizeof(uint16_t), hipMemcpyDeviceToHost,
                                ctx->stream));
    ARVX_SYNC(ctx);
    return ARVX_OK;
}

int arvx_selftest_round(arvx_ctx *ctx, int64_t *mismatches) {
    ARVX_CHECK_CTX(ctx);
    if (!mismatches) return fail(ARVX_ERR_INVALID, "null mismatches");
    *mismatches = -1;
    if (int rc = ensure_scratch(ctx, 64)) return rc;
    unsigned long long *d_bad = (unsigned long long *)ctx->d_scratch;
    ARVX_HIP(hipMemsetAsync(d_bad, 0, sizeof *d_bad, ctx->stream));
    // every float in [+0, 2^24] and in (-0.5, -0]: all quotients that can be inside an image
    const unsigned ranges[2][2] = {{0x00000000u, 0x4B800000u}, {0x80000000u, 0xBEFFFFFFu}};
    for (const auto &r : ranges) {
        const unsigned long long n = (unsigned long long)r[1] - r[0] + 1;
        hipLaunchKernelGGL(arvx::selftest_round_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256),
                           0, ctx->stream, r[0], r[1], d_bad);
        ARVX_HIP(hipGetLastError());
    }
    unsigned long long bad = 0;
    ARVX_HIP(hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream));
    ARVX_SYNC(ctx);
    *mismatches = (int64_t)bad;
    return ARVX_OK;
}

#ifdef ARVX_TIMELINE
// diagnostic builds only: per-workgroup {start, end (100 MHz ticks), xcc id, 0} of the last carve
extern "C" int arvx_debug_timeline(arvx_ctx *ctx, unsigned long long *out, int64_t *n) {
    ARVX_CHECK_CTX(ctx);
    *n = ctx->timeline_n;
    if (out && ctx->d_timeline) {
        ARVX_SYNC(ctx);
        ARVX_HIP(hipMemcpy(out, ctx->d_timeline, (size_t)ctx->timeline_n * ctx->timeline_rec,
                           hipMemcpyDeviceToHost));
    }
    return ARVX_OK;
}
#endif

// ---- closure -------------------------------------------------------------------------

int arvx_colors_upload(arvx_ctx *ctx, int64_t n, const int64_t *index, const float *rgb) {
    ARVX_CHECK_CTX(ctx);
    if (n < 0 || (n > 0 && (!index || !rgb))) return fail(ARVX_ERR_INVALID, "bad colour list");
    ctx->free_surface();
    std::vector<int> idx((size_t)n);
    const long long own_base = (long long)ctx->X * ctx->Y * (ctx->z0 - ctx->ze0);
    for (int64_t k = 0; k < n; ++k) {
        if (index[k] < 0 || (size_t)index[k] >= ctx->nvox || (k && index[k] <= index[k - 1]))
            return fail(ARVX_ERR_INVALID, "colour indices must be ascending and inside the grid");
        idx[(size_t)k] = (int)(index[k] + own_base);  // numbered over the context's planes
    }
    ctx->surf_count = n;
    ctx->surf_host_count = n;
    ctx->h_surf_index = idx;
    ctx->h_surf_has.assign((size_t)n, 1);
    if (n > 0) {
        ARVX_HIP(ctx->pool_surf_index.reserve((size_t)n * sizeof(int)));
        ctx->d_surf_index = (int *)ctx->pool_surf_index.p;
        ARVX_HIP(ctx->pool_surf_rgb.reserve((size_t)n * sizeof(float4)));
        ctx->d_surf_rgba = (float4 *)ctx->pool_surf_rgb.p;
        ARVX_HIP(ctx->pool_surf_depth.reserve((size_t)n * sizeof(float)));
        ctx->d_surf_depth = (float *)ctx->pool_surf_depth.p;
        ARVX_HIP(ctx->pool_surf_has.reserve((size_t)n));
        ctx->d_surf_has = (uint8_t *)ctx->pool_surf_has.p;
        ARVX_HIP(hipMemcpyAsync(ctx->d_surf_index, idx.data(), (size_t)n * sizeof(int),
                                hipMemcpyHostToDevice, ctx->stream));
        // three floats per voxel from the host, (r, g, b, has = 1) on the device
        if (int rc = ensure_scratch(ctx, (size_t)n * 3 * sizeof(float) + 64)) return rc;
        ARVX_HIP(hipMemcpyAsync(ctx->d_scratch, rgb, (size_t)n * 3 * sizeof(float),
                                hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(arvx::rgb_to_rgba_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           ctx->stream, (const float *)ctx->d_scratch, (long long)n,
                           ctx->d_surf_rgba);
        ARVX_HIP(hipGetLastError());
        ARVX_SYNC(ctx);  // (the scratch buffer is reused below)
        ARVX_HIP(hipMemsetAsync(ctx->d_surf_depth, 0, (size_t)n * sizeof(float), ctx->stream));
        ARVX_HIP(hipMemsetAsync(ctx->d_surf_has, 1, (size_t)n, ctx->stream));
        // the list's plane + ranks (what arvx_color leaves behind)
        const int XW = (ctx->X + 63) / 64;
        const arvx::BitGrid gown{ctx->X, ctx->Y, ctx->ze1 - ctx->ze0, XW};  // the context's planes
        const size_t nw = (size_t)XW * gown.Y * gown.Z;
        const int nblk = (int)((nw + arvx::kBitBlock - 1) / arvx::kBitBlock);
        ARVX_HIP(ctx->pool_col_bits.reserve(nw * sizeof(unsigned long long)));
        ARVX_HIP(ctx->pool_col_rank.reserve(nw * sizeof(arvx::SparseWord)));
        if (int rc = ensure_scratch(ctx, (size_t)(nblk + 1) * sizeof(long long) +
                                             (size_t)nblk * sizeof(int) + 64))
            return rc;
        unsigned long long *bits = (unsigned long long *)ctx->pool_col_bits.p;
        long long *d_off = (long long *)ctx->d_scratch;
        int *d_cnt = (int *)(d_off + nblk + 1);
        ARVX_HIP(hipMemsetAsync(bits, 0, nw * sizeof(unsigned long long), ctx->stream));
        hipLaunchKernelGGL(arvx::bits_from_index_kernel, dim3((unsigned)((n + 255) / 256)),
                           dim3(256), 0, ctx->stream, ctx->d_surf_index, (long long)n, gown, bits);
        ARVX_HIP(hipGetLastError());
        long long total = 0;
        if (int rc = bit_compact_count(ctx, bits, nw, d_cnt, d_off, &total)) return rc;
        if (int rc = bit_compact_write(ctx, bits, nw, gown, d_off, nullptr,
                                       (arvx::SparseWord *)ctx->pool_col_rank.p))
            return rc;
        ARVX_SYNC(ctx);
    }
    ctx->color_ready = true;
    return ARVX_OK;
}

// the context's sparse lists as plane + rank (empty: nulls)
static arvx::SparseList colour_list(const Ctx *ctx) {
    if (!ctx->color_ready || ctx->surf_count <= 0) return arvx::SparseList{nullptr};
    return arvx::SparseList{(const arvx::SparseWord *)ctx->pool_col_rank.p};
}
static arvx::SparseList closure_list(const Ctx *ctx) {
    if (!ctx->closure_ready || ctx->clo_count <= 0) return arvx::SparseList{nullptr};
    return arvx::SparseList{(const arvx::SparseWord *)ctx->pool_clo_rank.p};
}

int arvx_closure(arvx_ctx *ctx, int kernel_size, int apply_unseen) {
    ARVX_CHECK_CTX(ctx);
    if (kernel_size < 1 || kernel_size % 2 != 1 || kernel_size > 9)
        return fail(ARVX_ERR_INVALID, "kernel size %d (odd, 1..9)", kernel_size);
    if (ctx->stripe_world > 1)
        return fail(ARVX_ERR_STATE, "arvx_closure needs contiguous slabs (neighbour planes)");
    if (ctx->closure_ready)
        return fail(ARVX_ERR_STATE, "closure already applied to this model state");
    const int radius = (kernel_size - 1) / 2;
    // the planes that get filled here: all owned ones, plus the halo planes whose box lies
    // inside the planes that carry colours (stage_ranges)
    int c_lo, c_hi, f_lo, f_hi;
    stage_ranges(ctx, radius, c_lo, c_hi, f_lo, f_hi);
    if (f_lo > ctx->z0 || f_hi < ctx->z1)
        return fail(ARVX_ERR_STATE,
                    "a slab needs %d halo planes for a closure of size %d (it has %d): "
                    "arvx_ctx_create_slab_halo", radius + 1, kernel_size, ctx->halo);
    // (the planes are built from the lazy form; the tiles that exist only as a code are written out
    // further down, just before rec_or_bitgrid_kernel needs every record to exist -- written out
    // first, their non-temporal stores had emptied the caches of the records the planes are built from)
    if (int mrc = need_rec(ctx, true)) return mrc;
    ctx->free_closure();
    // filled = dilate(occupied, box of radius r) and not occupied, on bit planes over the
    // context's planes
    const int XW = (ctx->X + 63) / 64;
    const int Zext = ctx->ze1 - ctx->ze0;
    const arvx::BitGrid g{ctx->X, ctx->Y, Zext, XW};
    const size_t row_words = (size_t)XW * g.Y;
    const size_t nwords = row_words * g.Z;
    const bool paints = apply_unseen || ctx->paint_valid;  // somebody is UNSEEN_COLOR
    if (int rc = ensure_scratch(ctx, 3 * nwords * sizeof(unsigned long long) + 64)) return rc;
    ARVX_HIP(ctx->pool_clo_bits.reserve(nwords * sizeof(unsigned long long)));
    ARVX_HIP(ctx->pool_clo_rank.reserve(nwords * sizeof(arvx::SparseWord)));
    unsigned long long *d_occ = (unsigned long long *)ctx->d_scratch;
    unsigned long long *d_unseen = d_occ + nwords, *d_b = d_unseen + nwords;
    unsigned long long *d_fill = (unsigned long long *)ctx->pool_clo_bits.p;
    const unsigned gw = (unsigned)((nwords + 255) / 256);
    if (ctx->planes_ok && ctx->planes_seq == ctx->state_seq && !ctx->paint_valid && ctx->planes_words == nwords &&
        ctx->pool_state_planes.cap >= 2 * nwords * 8) {
        // the colour pass's planes are the state's (nothing but handleUnseen ran since): what the
        // closure calls occupied is their occupancy, with the never-seen voxels once those are occupied
        // (handleUnseen ran, or the caller says apply_unseen); the UNSEEN_COLOR plane is the never-seen one
        const unsigned long long *p_occ = (const unsigned long long *)ctx->pool_state_planes.p;
        const unsigned long long *p_nseen = p_occ + nwords;
        const bool merge = ctx->planes_unseen || apply_unseen;
        hipLaunchKernelGGL(arvx::bit_dilate_xy_kernel, dim3(gw), dim3(256), 0, ctx->stream, p_occ,
                           merge ? p_nseen : (const unsigned long long *)nullptr, g, radius, d_b, d_occ);
        if (!merge) d_occ = const_cast<unsigned long long *>(p_occ);
        d_unseen = const_cast<unsigned long long *>(p_nseen);  // (read only where `paints`)
    } else {
        if (int rc = launch_bit_pack(ctx, g, 1, apply_unseen ? 1 : 0, d_occ, paints ? d_unseen : nullptr))
            return rc;
        hipLaunchKernelGGL(arvx::bit_dilate_xy_kernel, dim3(gw), dim3(256), 0, ctx->stream,
                           (const unsigned long long *)d_occ, (const unsigned long long *)nullptr, g, radius, d_b,
                           (unsigned long long *)nullptr);
    }
    // (halo planes outside [f_lo, f_hi): their boxes reach planes this context knows nothing
    // about -- not filled here, their owners do it: zeros)
    int *d_counts = nullptr;
    if (int rc = chunk_counts(ctx, nwords, &d_counts)) return rc;
    arvx::CarveParams rp;
    carve_geometry(ctx, rp);
    rp.rec = ctx->d_rec;
    // Coarse tiles that exist only as their code: the few that receive a voxel are marked by the fill
    // plane's producer and written out by rec_or_bitgrid_lazy_kernel, the others stay codes (grids
    // whose rows and planes fill whole tiles; else every tile is written out first, as in round 4)
    const bool lazy_or = ctx->lazy && ctx->Y % 8 == 0 && Zext % 8 == 0;
    arvx::CoarseMark mark{lazy_or ? (uint8_t *)ctx->pool_ccode.p : nullptr, rp.coarseX, rp.coarseY, rp.cyShift,
                          rp.czShift};
    hipLaunchKernelGGL(arvx::bit_dilate_z_count_kernel, dim3(gw), dim3(256), 0, ctx->stream, d_b, g,
                       radius, (const unsigned long long *)d_occ, f_lo - ctx->ze0, f_hi - ctx->ze0, d_fill,
                       d_counts, mark);
    ARVX_HIP(hipGetLastError());
    // The filled voxels are occupied from now on (their w is count / count = 1; the fill kernel reads
    // the occupancy from the bit planes, not from the records).  Launched HERE, right behind the
    // producer that marked the tiles: from this launch on the marks and the records agree, whatever
    // happens to the rest of the call.
    if (!lazy_or)
        if (int mrc = need_rec(ctx)) return mrc;  // every record exists from here on
    ++ctx->state_seq;
    state_rewritten(ctx);  // (voxels of tiles an earlier carve emptied may be occupied again)
    if (lazy_or)
        hipLaunchKernelGGL(arvx::rec_or_bitgrid_lazy_kernel, dim3(gw), dim3(256), 0, ctx->stream, rp, g.Z,
                           (const unsigned long long *)d_fill, (const uint8_t *)ctx->pool_ccode.p);
    else
        hipLaunchKernelGGL(arvx::rec_or_bitgrid_kernel, dim3(gw), dim3(256), 0, ctx->stream, rp, 0, g.Z,
                           d_fill);
    ARVX_HIP(hipGetLastError());
    // The list of the filled voxels: compacted in one launch into buffers sized for what the last
    // closure needed (first call: a shell's share of the voxels); the true length is read at the
    // call's ONE synchronisation, and a list that outgrew its buffers is written again.
    long long cap = (long long)(ctx->pool_clo_index.cap / sizeof(int));
    if (cap <= 0) {
        const double v = (double)nwords * 64.0;
        cap = (long long)std::min<double>(v, 8.0 * std::cbrt(v) * std::cbrt(v) + 4096.0);
    }
    long long total = 0;
    for (int attempt = 0;; ++attempt) {
        ARVX_HIP(ctx->pool_clo_index.reserve((size_t)cap * sizeof(int)));
        ctx->d_clo_index = (int *)ctx->pool_clo_index.p;
        ARVX_HIP(ctx->pool_clo_rgba.reserve((size_t)cap * sizeof(float4)));
        ctx->d_clo_rgba = (void *)ctx->pool_clo_rgba.p;
        const long long *d_total = nullptr;
        if (int rc = bit_compact(ctx, d_fill, nwords, g, cap, ctx->d_clo_index,
                                 (arvx::SparseWord *)ctx->pool_clo_rank.p, 1, &d_total))
            return rc;
        arvx::ClosureParams cp;
        cp.g = g;
        cp.occ = d_occ;
        cp.unseen = paints ? d_unseen : nullptr;
        cp.radius = radius;
        cp.col = colour_list(ctx);
        cp.col_rgba = ctx->d_surf_rgba;
        if (radius == 1)
            hipLaunchKernelGGL(arvx::closure_fill_kernel<true>, dim3((unsigned)((cap + 255) / 256)),
                               dim3(256), 0, ctx->stream, cp, ctx->d_clo_index, cap, d_total,
                               (float4 *)ctx->d_clo_rgba);
        else
            hipLaunchKernelGGL(arvx::closure_fill_kernel<false>, dim3((unsigned)((cap + 255) / 256)),
                               dim3(256), 0, ctx->stream, cp, ctx->d_clo_index, cap, d_total,
                               (float4 *)ctx->d_clo_rgba);
        ARVX_HIP(hipGetLastError());
        ARVX_SYNC(ctx);
        total = host_total(ctx, 1);
        if (total < 0) return fail(ARVX_ERR_HIP, "the compaction left no count");
        if (total <= cap || attempt) break;
        cap = total + total / 8;  // (once more, with room for all)
    }
    ctx->clo_count = total;
    ctx->clo_host_count = -1;  // the list's host copy is fetched when somebody asks
    if (total == 0) {
        ctx->d_clo_index = nullptr;
        ctx->d_clo_rgba = nullptr;
    }
    ctx->closure_ready = true;
    ctx->closure_unseen = apply_unseen ? 1 : 0;
    ctx->closure_radius = radius;
    return ARVX_OK;
}

// the closure list's indices on the host: fetched when somebody asks
static int clo_host(Ctx *ctx) {
    if (ctx->clo_host_count == ctx->clo_count) return ARVX_OK;
    const size_t n = (size_t)ctx->clo_count;
    ctx->h_clo_index.resize(n);
    if (n) {
        ARVX_HIP(hipMemcpyAsync(ctx->h_clo_index.data(), ctx->d_clo_index, n * sizeof(int),
                                hipMemcpyDeviceToHost, ctx->stream));
        ARVX_SYNC(ctx);
    }
    ctx->clo_host_count = ctx->clo_count;
    return ARVX_OK;
}

int arvx_closure_count(arvx_ctx *ctx, int64_t *count) {
    if (!ctx || !count) return fail(ARVX_ERR_INVALID, "null argument");
    if (!ctx->closure_ready) return fail(ARVX_ERR_STATE, "no closure result (call arvx_closure)");
    ARVX_HIP(hipSetDevice(ctx->device));
    if (owns_all_planes(ctx)) {  // (the list itself stays on the device until somebody asks for it)
        *count = (int64_t)ctx->clo_count;
        return ARVX_OK;
    }
    if (int rc = clo_host(ctx)) return rc;
    size_t lo, hi;
    long long base;
    owned_part(ctx, ctx->h_clo_index, lo, hi, base);
    *count = (int64_t)(hi - lo);  // (the filled voxels of the owned planes)
    return ARVX_OK;
}

int arvx_closure_download(arvx_ctx *ctx, int64_t *index, float *rgba) {
    ARVX_CHECK_CTX(ctx);
    if (!index || !rgba) return fail(ARVX_ERR_INVALID, "null argument");
    if (!ctx->closure_ready) return fail(ARVX_ERR_STATE, "no closure result (call arvx_closure)");
    if (int rc_h = clo_host(ctx)) return rc_h;
    size_t lo, hi;
    long long base;
    owned_part(ctx, ctx->h_clo_index, lo, hi, base);
    for (size_t k = lo; k < hi; ++k) index[k - lo] = ctx->h_clo_index[k] - base;
    if (hi > lo) {
        ARVX_HIP(hipMemcpyAsync(rgba, (const float4 *)ctx->d_clo_rgba + lo,
                                (hi - lo) * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        ARVX_SYNC(ctx);
    }
    return ARVX_OK;
}

// ---- marching-cubes hand-off ----------------------------------------------------------

// The cell list, launched without a synchronisation: up to `cap` cells into the context's buffer,
// the list's true length in device word *d_total (and ctx->h_totals[2] after the next sync).
static int mc_cells_launch(Ctx *ctx, long long cap, const long long **d_total) {
    if (int mrc = need_rec(ctx, true)) return mrc;  // (mc_zpack_rec_kernel reads lazy tiles)
    arvx::McParams mp;
    mp.X = ctx->X;
    mp.Y = ctx->Y;
    mp.Z = ctx->Z;
    mp.ze0 = ctx->ze0;
    mp.ze1 = ctx->ze1;
    // a cell belongs to the slab that owns its upper plane; the last slab also takes
    // the cells whose upper plane is outside the grid
    mp.cz0 = ctx->z0 - 1;
    mp.cz1 = (ctx->z1 == ctx->Z) ? ctx->Z : ctx->z1 - 1;
    mp.ZW = (mp.cz1 - mp.cz0 + 1 + 63) / 64;
    const size_t nzw = (size_t)mp.ZW * ctx->X * ctx->Y;  // z-packed occupancy words
    const long long ncol = (long long)(ctx->X + 1) * (ctx->Y + 1);
    if (int rc = ensure_scratch(ctx, nzw * sizeof(unsigned long long) +
                                         2 * (size_t)(ncol + 1) * sizeof(int) + 64))
        return rc;
    mp.zbits = (unsigned long long *)ctx->d_scratch;
    int *d_off = (int *)(mp.zbits + nzw);  // ncol column offsets
    int *d_cnt = d_off + ncol + 1;
    {
        arvx::CarveParams g;
        carve_geometry(ctx, g);
        g.rec = ctx->d_rec;
        hipLaunchKernelGGL(arvx::mc_zpack_rec_kernel,
                           dim3((unsigned)((size_t)g.tilesX * g.tilesY * mp.ZW)), dim3(256), 0,
                           ctx->stream, g, mp);
        ARVX_HIP(hipGetLastError());
    }
    const unsigned nblk = (unsigned)((ncol + 255) / 256);
    hipLaunchKernelGGL(arvx::mc_count_kernel, dim3(nblk), dim3(256), 0, ctx->stream, mp, d_cnt);
    ARVX_HIP(hipGetLastError());
    // the columns' offsets in the list: one launch
    if (int rc = scan_counts(ctx, d_cnt, nullptr, ncol, nullptr, d_off, 2, d_total)) return rc;
    ARVX_HIP(ctx->pool_mc_cells.reserve((size_t)cap * sizeof(int4)));
    ctx->d_mc_cells = (void *)ctx->pool_mc_cells.p;
    hipLaunchKernelGGL(arvx::mc_write_kernel, dim3(nblk), dim3(256), 0, ctx->stream, mp, d_off, cap,
                       (int4 *)ctx->d_mc_cells);
    ARVX_HIP(hipGetLastError());
    return ARVX_OK;
}
// room for the cell list before its length is known: what the last list needed, or a surface's share
static long long mc_cells_cap(const Ctx *ctx) {
    long long cap = (long long)(ctx->pool_mc_cells.cap / sizeof(int4));
    if (cap <= 0) {
        const double v = (double)ctx->nvox_ext;
        cap = (long long)std::min<double>(v + 1e6, 8.0 * std::cbrt(v) * std::cbrt(v) + 4096.0);
    }
    return cap;
}

int arvx_mc_cells(arvx_ctx *ctx, int64_t *count) {
    ARVX_CHECK_CTX(ctx);
    if (!count) return fail(ARVX_ERR_INVALID, "null argument");
    if (ctx->stripe_world > 1)
        return fail(ARVX_ERR_STATE, "the cell walk needs contiguous slabs (neighbour planes)");
    ctx->free_mc();
    long long cap = mc_cells_cap(ctx), total = 0;
    for (int attempt = 0;; ++attempt) {  // ONE synchronisation; a second round only if the list outgrew its room
        const long long *d_total = nullptr;
        if (int rc = mc_cells_launch(ctx, cap, &d_total)) return rc;
        ARVX_SYNC(ctx);
        total = host_total(ctx, 2);
        if (total < 0) return fail(ARVX_ERR_HIP, "the scan left no count");
        if (total <= cap || attempt) break;
        cap = total + total / 8;
    }
    if (total == 0) ctx->d_mc_cells = nullptr;
    ctx->mc_count = total;
    ctx->mc_ready = true;
    *count = total;
    return ARVX_OK;
}

int arvx_mc_cells_download(arvx_ctx *ctx, int32_t *cells) {
    ARVX_CHECK_CTX(ctx);
    if (!cells) return fail(ARVX_ERR_INVALID, "null argument");
    if (!ctx->mc_ready) return fail(ARVX_ERR_STATE, "no cell list (call arvx_mc_cells)");
    if (ctx->mc_count) {
        ARVX_HIP(hipMemcpyAsync(cells, ctx->d_mc_cells, (size_t)ctx->mc_count * sizeof(int4),
                                hipMemcpyDeviceToHost, ctx->stream));
        ARVX_SYNC(ctx);
    }
    return ARVX_OK;
}

int arvx_mc_mesh(arvx_ctx *ctx, int apply_unseen, int64_t *triangles) {
    ARVX_CHECK_CTX(ctx);
    if (!triangles) return fail(ARVX_ERR_INVALID, "null argument");
    if (ctx->stripe_world > 1)
        return fail(ARVX_ERR_STATE, "arvx_mc_mesh needs contiguous slabs (neighbour planes)");
    if (ctx->closure_ready && (apply_unseen != 0) != (ctx->closure_unseen != 0))
        return fail(ARVX_ERR_STATE, "arvx_closure was computed with apply_unseen=%d",
                    ctx->closure_unseen);
    {
        // a slab lists the cells whose upper plane it owns: their corners lie in planes
        // z0 - 1 .. z1 - 1, and plane z0 - 1 (a halo plane) must carry whatever the owned
        // planes carry -- colours, the closure's fills -- or the cells at the slab's lower
        // face would be coloured differently from the whole-grid result
        int c_lo, c_hi, f_lo, f_hi;
        stage_ranges(ctx, ctx->closure_radius, c_lo, c_hi, f_lo, f_hi);
        const int low = ctx->z0 > 0 ? ctx->z0 - 1 : 0;
        if (ctx->color_ready && ctx->surf_count > 0 && c_lo > low)
            return fail(ARVX_ERR_STATE, "a slab's mesh needs 2 halo planes once colours exist "
                                        "(it has %d): arvx_ctx_create_slab_halo", ctx->halo);
        if (ctx->closure_ready && f_lo > low)
            return fail(ARVX_ERR_STATE, "a slab's mesh needs %d halo planes after a closure of "
                                        "radius %d (it has %d): arvx_ctx_create_slab_halo",
                        ctx->closure_radius + 2, ctx->closure_radius, ctx->halo);
    }
    // cells -> triangles per cell -> triangles, all launched before the call's ONE synchronisation:
    // the lists' lengths are not known when their buffers are sized (what the last mesh needed, or
    // a surface's share of the voxels), the kernels stop at the room they have, and a mesh that
    // outgrew it is built once more with room for all.
    ctx->free_mc();
    ctx->mesh_tris = 0;
    *triangles = 0;
    long long ccap = mc_cells_cap(ctx);
    long long tcap = (long long)(ctx->pool_mesh_verts.cap / (9 * sizeof(float)));
    if (tcap <= 0) tcap = 2 * ccap;
    long long ncells = 0, total = 0;
    for (int attempt = 0;; ++attempt) {
        const long long *d_ncells = nullptr;
        if (int rc = mc_cells_launch(ctx, ccap, &d_ncells)) return rc;
        ARVX_HIP(ctx->pool_mesh_verts.reserve((size_t)tcap * 9 * sizeof(float)));
        ARVX_HIP(ctx->pool_mesh_rgb.reserve((size_t)tcap * 6 * sizeof(unsigned)));  // face records
        arvx::McMeshParams mp;
        carve_geometry(ctx, mp.g);
        mp.g.rec = ctx->d_rec;
        mp.paint = paint_plane(ctx);
        mp.apply_unseen = apply_unseen ? 1 : 0;
        mp.col = colour_list(ctx);
        mp.col_rgba = ctx->d_surf_rgba;
        mp.clo = closure_list(ctx);
        mp.clo_rgba = (const float4 *)ctx->d_clo_rgba;
        // (the scratch buffer holds mc_cells_launch's arrays: the triangle offsets get their own)
        ARVX_HIP(ctx->pool_mesh_off.reserve((size_t)(ccap + 1) * sizeof(int)));
        int *d_off = (int *)ctx->pool_mesh_off.p;
        const long long *d_ntris = nullptr;
        if (int rc = scan_counts(ctx, nullptr, (const int4 *)ctx->d_mc_cells, ccap, d_ncells, d_off, 3, &d_ntris))
            return rc;
        hipLaunchKernelGGL(arvx::mc_mesh_kernel, dim3((unsigned)((ccap + 255) / 256)), dim3(256), 0,
                           ctx->stream, mp, (const int4 *)ctx->d_mc_cells, ccap, d_ncells, tcap, d_off,
                           (float *)ctx->pool_mesh_verts.p, (unsigned *)ctx->pool_mesh_rgb.p);
        ARVX_HIP(hipGetLastError());
        ARVX_SYNC(ctx);
        ncells = host_total(ctx, 2);
        total = host_total(ctx, 3);
        if (ncells < 0 || total < 0) return fail(ARVX_ERR_HIP, "the scans left no count");
        if ((ncells <= ccap && total <= tcap) || attempt) break;
        // (with too few cells the triangle count is of the cells that fitted: five per cell at most)
        if (ncells > ccap) {
            tcap = std::max(tcap, total + 5 * (ncells - ccap));
            ccap = ncells + ncells / 8;
        }
        if (total > tcap) tcap = total + total / 8;
    }
    if (ncells == 0) ctx->d_mc_cells = nullptr;
    ctx->mc_count = ncells;
    ctx->mc_ready = true;
    ctx->mesh_tris = total;
    *triangles = total;
    return ARVX_OK;
}

int arvx_mc_mesh_download(arvx_ctx *ctx, float *verts, uint32_t *face_rgb) {
    ARVX_CHECK_CTX(ctx);
    if (!verts || !face_rgb) return fail(ARVX_ERR_INVALID, "null argument");
    if (ctx->mesh_tris > 0) {
        ARVX_HIP(hipMemcpyAsync(verts, ctx->pool_mesh_verts.p, (size_t)ctx->mesh_tris * 36,
                                hipMemcpyDeviceToHost, ctx->stream));
        // r, g, b of the 24-byte face records
        ARVX_HIP(hipMemcpy2DAsync(face_rgb, 12, (const uint8_t *)ctx->pool_mesh_rgb.p + 12, 24, 12,
                                  (size_t)ctx->mesh_tris, hipMemcpyDeviceToHost, ctx->stream));
        ARVX_SYNC(ctx);
    }
    return ARVX_OK;
}

int arvx_mc_mesh_download_faces(arvx_ctx *ctx, float *verts, uint32_t *faces) {
    ARVX_CHECK_CTX(ctx);
    if (!verts || !faces) return fail(ARVX_ERR_INVALID, "null argument");
    if (ctx->mesh_tris > 0) {
        ARVX_HIP(hipMemcpyAsync(verts, ctx->pool_mesh_verts.p, (size_t)ctx->mesh_tris * 36,
                                hipMemcpyDeviceToHost, ctx->stream));
        ARVX_HIP(hipMemcpyAsync(faces, ctx->pool_mesh_rgb.p, (size_t)ctx->mesh_tris * 24,
                                hipMemcpyDeviceToHost, ctx->stream));
        ARVX_SYNC(ctx);
    }
    return ARVX_OK;
}

int arvx_closure_download32(arvx_ctx *ctx, int32_t *index, float *rgba) {
    ARVX_CHECK_CTX(ctx);
    if (!index || !rgba) return fail(ARVX_ERR_INVALID, "null argument");
    if (!ctx->closure_ready) return fail(ARVX_ERR_STATE, "no closure result (call arvx_closure)");
    if (int rc_h = clo_host(ctx)) return rc_h;
    size_t lo, hi;
    long long base;
    owned_part(ctx, ctx->h_clo_index, lo, hi, base);
    if (hi > lo) {
        if (base == 0) memcpy(index, ctx->h_clo_index.data() + lo, (hi - lo) * sizeof(int32_t));
        else for (size_t k = lo; k < hi; ++k) index[k - lo] = (int32_t)(ctx->h_clo_index[k] - base);
        ARVX_HIP(hipMemcpyAsync(rgba, (const float4 *)ctx->d_clo_rgba + lo,
                                (hi - lo) * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
        ARVX_SYNC(ctx);
    }
    return ARVX_OK;
}

// ---- greedy carve -------------------------------------------------------------------

int arvx_fast_carve(arvx_ctx *ctx) {
    ARVX_CHECK_CTX(ctx);
    if (!ctx->views_ready) return fail(ARVX_ERR_STATE, "arvx_set_views has not been called");
    if (ctx->z0 != 0 || ctx->z1 != ctx->Z || ctx->stripe_world > 1)
        return fail(ARVX_ERR_STATE,
                    "arvx_fast_carve needs the whole grid in one context (connectivity is global)");
    state_changes(ctx, false);
    arvx::FloodParams fp;
    // a fresh model (the usual case: src/main.cpp calls fastCarve on a new Model) is neither
    // filled nor read: the kernels know its records, and flood_apply_rec_kernel writes them all
    fp.fresh = ctx->fresh_pending ? 1 : 0;
    if (fp.fresh) {
        void *buf = ctx->d_rec;
        if (int rc = ensure_records(ctx, &buf, &ctx->rec_bytes)) return rc;
        ctx->d_rec = (uint16_t *)buf;
    } else if (int mrc = need_rec(ctx)) {
        return mrc;
    }
    fp.X = ctx->X;
    fp.Y = ctx->Y;
    fp.Z = ctx->Z;
    fp.XW = (ctx->X + 63) / 64;
    const size_t nwords = (size_t)fp.XW * fp.Y * fp.Z;
    const int tilesY = (fp.Y + 15) / 16, tilesZ = (fp.Z + 15) / 16;
    const int tile_rows = tilesY * tilesZ;
    const unsigned gflood = (unsigned)((size_t)fp.XW * tile_rows);
    const bool prepass = fp.XW <= 64 && tile_rows <= arvx::kFloodMaxTileRows;

    // one work buffer, kept by the context: open, reach bit planes | whole-tile rows (full,
    // reached) | wake flags (2 x tiles) | changed flag
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t o_bits = 0;
    const size_t o_tiles = o_bits + up(2 * nwords * sizeof(unsigned long long));
    const size_t o_dirty = o_tiles + up(2 * (size_t)tile_rows * sizeof(unsigned long long));
    const size_t o_changed = o_dirty + up(2 * (size_t)gflood);
    const size_t need = o_changed + 256;
    if (ctx->flood_bytes < need) {
        if (ctx->d_flood) (void)hipFree(ctx->d_flood);
        ctx->d_flood = nullptr;
        ctx->flood_bytes = 0;
        ARVX_HIP(hipMalloc(&ctx->d_flood, need));
        ctx->flood_bytes = need;
    }
    uint8_t *base = (uint8_t *)ctx->d_flood;
    fp.open = (unsigned long long *)(base + o_bits);
    fp.reach = fp.open + nwords;
    unsigned long long *d_tiles = (unsigned long long *)(base + o_tiles);
    uint8_t *d_dirty = base + o_dirty;
    fp.changed = (int *)(base + o_changed);
    fp.dirty_cur = fp.dirty_next = nullptr;  // set per launch of flood_step_kernel

    // carvable = what the dense carve clears on a fresh model: carved into records of its own
    if (int rc = ensure_records(ctx, &ctx->d_flood_rec, &ctx->flood_rec_bytes)) return rc;
    arvx::CarveParams g;
    carve_geometry(ctx, g);
    g.rec = ctx->d_rec;
    // (the coarse tiles that carve settles as a whole exist only as their codes, as in the model's
    // own lazy state: no N / 4 bytes of constants written here and read back by the conversion)
    const size_t ncoarse_f = (size_t)g.coarseX * g.coarseY * g.coarseZ;
    ARVX_HIP(ctx->pool_flood_code.reserve(ncoarse_f + 64));
    uint8_t *d_carv_code = (uint8_t *)ctx->pool_flood_code.p;
    if (int rc = launch_carve(ctx, (uint16_t *)ctx->d_flood_rec, 0, ctx->V, 0, true, d_carv_code)) return rc;
    // (both conversions: one workgroup per row of tiles, 64 rows x (tiles along x + 1) words
    // of LDS, twice that for the way back)
    const int chunk = std::min(arvx::kFloodChunk, fp.XW);
    const size_t lds_words = (size_t)64 * (chunk + 1);
    hipLaunchKernelGGL(arvx::flood_open_from_rec_kernel, dim3((unsigned)(g.tilesY * g.tilesZ)),
                       dim3(256), lds_words * sizeof(unsigned long long), ctx->stream, g,
                       (const uint16_t *)ctx->d_flood_rec, (const uint8_t *)d_carv_code, fp);
    ARVX_HIP(hipGetLastError());
    // whole-tile pre-pass (fast_carve_kernels.h): seeds every completely open tile
    // that is connected to the origin tile through completely open tiles
    if (prepass) {
        const size_t tile_bytes = 2 * (size_t)tile_rows * sizeof(unsigned long long);
        hipLaunchKernelGGL(arvx::flood_tile_full_kernel, dim3(tile_rows), dim3(256), 0,
                           ctx->stream, fp, d_tiles, d_dirty);
        hipLaunchKernelGGL(arvx::flood_tile_fill_kernel, dim3(1), dim3(1024), tile_bytes,
                           ctx->stream, d_tiles, d_tiles + tile_rows, tilesY, tilesZ);
        hipLaunchKernelGGL(arvx::flood_tile_seed_kernel, dim3(tile_rows), dim3(256), 0,
                           ctx->stream, fp, d_tiles + tile_rows);
        ARVX_HIP(hipGetLastError());
    }
    // per-tile wake flags, two buffers swapped every launch; without the pre-pass
    // (which sets the first buffer) all tiles are awake at first
    if (!prepass) ARVX_HIP(hipMemsetAsync(d_dirty, 1, gflood, ctx->stream));
    ARVX_HIP(hipMemsetAsync(d_dirty + gflood, 0, gflood, ctx->stream));
    // every launch that is not the last grows at least one word; the launches of a batch write
    // one flag each, read back after 4, 4, 8, 8, ... launches: the fill has converged when the
    // LAST launch of a batch changed nothing (a launch with no tile awake costs microseconds)
    const long max_launches = 128 + (long)gflood * 256;
    long launched = 0;
    int *const flags = fp.changed;
    for (int batch = 4, round = 0;; ++round) {
        int changed[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        ARVX_HIP(hipMemsetAsync(flags, 0, 8 * sizeof(int), ctx->stream));
        for (int k = 0; k < batch; ++k, ++launched) {
            fp.dirty_cur = d_dirty + (launched & 1) * (size_t)gflood;
            fp.dirty_next = d_dirty + ((launched + 1) & 1) * (size_t)gflood;
            fp.changed = flags + k;
            hipLaunchKernelGGL(arvx::flood_step_kernel, dim3(gflood), dim3(256), 0, ctx->stream,
                               fp);
            ARVX_HIP(hipGetLastError());
        }
        ARVX_HIP(hipMemcpyAsync(changed, flags, 8 * sizeof(int), hipMemcpyDeviceToHost,
                                ctx->stream));
        ARVX_SYNC(ctx);
        if (!changed[batch - 1]) break;
        if (launched >= max_launches) return fail(ARVX_ERR_HIP, "flood fill did not converge");
        if (round >= 1 && batch < 8) batch *= 2;
    }
    {
        const unsigned rows = (unsigned)((g.coarseY << g.cyShift) * (g.coarseZ << g.czShift));
        hipLaunchKernelGGL(arvx::flood_apply_rec_kernel, dim3(rows), dim3(256),
                           2 * lds_words * sizeof(unsigned long long), ctx->stream, g, fp);
    }
    ARVX_HIP(hipGetLastError());
    ctx->fresh_pending = false;  // the records now hold every voxel's state
    ctx->rec_valid = true;
    ctx->lazy = false;
    ARVX_SYNC(ctx);
    return ARVX_OK;
}

}  // extern "C"
