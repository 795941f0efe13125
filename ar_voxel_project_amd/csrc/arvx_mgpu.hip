// arvx_mgpu.hip -- libarvx_mgpu.so: include/arvx/arvx_mgpu.h over the public C-ABI of
// libarvx.so and RCCL.  One process, one striped context + stream per device, one collective.
#include "arvx/arvx_mgpu.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    std::fprintf(stderr, "arvx_mgpu: %s\n", msg.c_str());
    return code;
}

#define MG_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(ARVX_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define MG_NCCL(call)                                                                       \
    do {                                                                                    \
        ncclResult_t r_ = (call);                                                           \
        if (r_ != ncclSuccess)                                                              \
            return fail(ARVX_ERR_RCCL, std::string(#call) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define MG_ARVX(call)                                                                    \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != ARVX_OK) return fail(rc_, std::string(#call) + ": " + arvx_last_error()); \
    } while (0)

struct Rank {
    int device = -1;
    arvx_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t *d_full = nullptr;     // merged occupancy of the whole grid
    uint32_t *d_local = nullptr;    // this device's planes, local order (compressed merge)
    uint64_t *d_packet = nullptr;   // its compressed packet
    uint64_t *d_packets = nullptr;  // all packets after the all-gather
    int *d_overflow = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
};

}  // namespace

struct arvx_mgpu {
    int n = 0, X = 0, Y = 0, Z = 0;
    float s = 0.f;
    size_t full_words = 0;    // 32-bit words of the whole grid's packed occupancy
    size_t local_words64 = 0; // 64-bit words of one device's planes
    int64_t cap64 = 0, packet_words = 0;
    std::vector<Rank> r;
    float carve_ms = 0.f, merge_ms = 0.f;
    // a collective was enqueued on some devices and not on others (an error in the middle of a
    // group): the communicators cannot be trusted any more, every later carve is refused
    bool broken = false;
};

namespace {
// ncclGroupStart ... ncclGroupEnd around the per-device calls of ONE collective: the group is
// always closed, and a failure inside it marks the handle unusable instead of leaving some
// devices waiting in a collective the others never entered.
template <class PerRank>
int grouped(arvx_mgpu *m, PerRank per_rank) {
    ncclResult_t r = ncclGroupStart();
    if (r != ncclSuccess)
        return fail(ARVX_ERR_RCCL, std::string("ncclGroupStart: ") + ncclGetErrorString(r));
    ncclResult_t bad = ncclSuccess;
    for (Rank &k : m->r) {
        bad = per_rank(k);
        if (bad != ncclSuccess) break;
    }
    r = ncclGroupEnd();
    if (bad != ncclSuccess || r != ncclSuccess) {
        m->broken = true;
        for (Rank &k : m->r)
            if (k.comm) (void)ncclCommAbort(k.comm), k.comm = nullptr;
        return fail(ARVX_ERR_RCCL, std::string("collective failed (handle is unusable now): ") +
                                       ncclGetErrorString(bad != ncclSuccess ? bad : r));
    }
    return ARVX_OK;
}
}  // namespace

extern "C" {

int arvx_mgpu_destroy(arvx_mgpu *m) {
    if (!m) return ARVX_OK;
    for (Rank &k : m->r) {
        if (k.device < 0) continue;
        (void)hipSetDevice(k.device);
        if (k.stream) (void)hipStreamSynchronize(k.stream);
        if (k.comm) (void)ncclCommDestroy(k.comm);
        if (k.ctx) arvx_ctx_destroy(k.ctx);
        for (void *p : {(void *)k.d_full, (void *)k.d_local, (void *)k.d_packet,
                        (void *)k.d_packets, (void *)k.d_overflow})
            if (p) (void)hipFree(p);
        for (hipEvent_t e : {k.e0, k.e1, k.e2})
            if (e) (void)hipEventDestroy(e);
        if (k.stream) (void)hipStreamDestroy(k.stream);
    }
    delete m;
    return ARVX_OK;
}

int arvx_mgpu_create(arvx_mgpu **out, const int *devices, int n, int X, int Y, int Z,
                     float voxel_size) {
    if (!out) return fail(ARVX_ERR_INVALID, "null out");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return fail(ARVX_ERR_INVALID, "bad device list");
    if (X < 1 || Y < 1 || Z < 1 || ((size_t)X * Y) % 64 || Z % (8 * n))
        return fail(ARVX_ERR_INVALID, "X*Y must be a multiple of 64 and Z a multiple of 8*n");
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return fail(ARVX_ERR_INVALID, "duplicate device");
    arvx_mgpu *m = new (std::nothrow) arvx_mgpu();
    if (!m) return fail(ARVX_ERR_NOMEM, "out of host memory");
    m->n = n;
    m->X = X;
    m->Y = Y;
    m->Z = Z;
    m->s = voxel_size;
    m->full_words = ((size_t)X * Y * Z + 31) / 32;
    m->local_words64 = (size_t)X * Y * (Z / n) / 64;
    m->cap64 = (int64_t)m->local_words64;  // worst case: every word mixed, cannot overflow
    m->packet_words = arvx_occupancy_packet_words((int64_t)m->local_words64, m->cap64);
    m->r.resize(n);
    std::vector<ncclComm_t> comms(n);
    ncclResult_t nr = ncclCommInitAll(comms.data(), n, devices);
    if (nr != ncclSuccess) {
        delete m;
        return fail(ARVX_ERR_RCCL, std::string("ncclCommInitAll: ") + ncclGetErrorString(nr));
    }
    for (int i = 0; i < n; ++i) {
        m->r[i].device = devices[i];
        m->r[i].comm = comms[i];
    }
    auto init = [&]() -> int {
        for (int i = 0; i < n; ++i) {
            Rank &k = m->r[i];
            MG_HIP(hipSetDevice(k.device));
            MG_HIP(hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking));
            MG_ARVX(arvx_ctx_create_striped(&k.ctx, k.device, X, Y, Z, voxel_size, n, i));
            MG_ARVX(arvx_ctx_set_stream(k.ctx, k.stream));
            MG_HIP(hipMalloc(&k.d_full, m->full_words * 4));
            MG_HIP(hipMalloc(&k.d_local, m->local_words64 * 8));
            MG_HIP(hipMalloc(&k.d_packet, (size_t)m->packet_words * 8));
            MG_HIP(hipMalloc(&k.d_packets, (size_t)m->packet_words * 8 * n));
            MG_HIP(hipMalloc(&k.d_overflow, sizeof(int)));
            MG_HIP(hipEventCreate(&k.e0));
            MG_HIP(hipEventCreate(&k.e1));
            MG_HIP(hipEventCreate(&k.e2));
        }
        return ARVX_OK;
    };
    if (int rc = init()) {
        arvx_mgpu_destroy(m);
        return rc;
    }
    *out = m;
    return ARVX_OK;
}

int arvx_mgpu_devices(const arvx_mgpu *m, int *n) {
    if (!m || !n) return fail(ARVX_ERR_INVALID, "null argument");
    *n = m->n;
    return ARVX_OK;
}

int arvx_mgpu_set_views(arvx_mgpu *m, int V, const float *M, const float *campos,
                        const uint8_t *const *masks, int W, int H, int C, size_t stride) {
    if (!m) return fail(ARVX_ERR_INVALID, "null handle");
    for (Rank &k : m->r) MG_ARVX(arvx_set_views(k.ctx, V, M, campos, masks, W, H, C, stride));
    return ARVX_OK;
}

int arvx_mgpu_state_reset(arvx_mgpu *m) {
    if (!m) return fail(ARVX_ERR_INVALID, "null handle");
    for (Rank &k : m->r) MG_ARVX(arvx_state_reset(k.ctx));
    return ARVX_OK;
}

// device i's planes in local order <-> the whole grid's planes: group g of device i is global
// group g * n + i; a group is 8 planes of wpr * Y words
static void stripe_copy(const arvx_mgpu *m, int i, uint32_t *local, uint32_t *global, bool to_local) {
    const size_t gw = (size_t)((m->X + 31) / 32) * m->Y * 8;
    const int groups = m->Z / 8 / m->n;
    for (int g = 0; g < groups; ++g) {
        uint32_t *a = local + (size_t)g * gw, *b = global + ((size_t)g * m->n + i) * gw;
        if (to_local) memcpy(a, b, gw * 4);
        else memcpy(b, a, gw * 4);
    }
}

int arvx_mgpu_state_upload_planes(arvx_mgpu *m, const uint32_t *occ, const uint32_t *seen) {
    if (!m || !occ || !seen) return fail(ARVX_ERR_INVALID, "null argument");
    const size_t lw = (size_t)((m->X + 31) / 32) * m->Y * (m->Z / m->n);
    std::vector<uint32_t> o(lw), s(lw);
    for (int i = 0; i < m->n; ++i) {
        stripe_copy(m, i, o.data(), const_cast<uint32_t *>(occ), true);
        stripe_copy(m, i, s.data(), const_cast<uint32_t *>(seen), true);
        MG_ARVX(arvx_state_upload_planes(m->r[i].ctx, o.data(), s.data()));
    }
    return ARVX_OK;
}

int arvx_mgpu_state_download_planes(arvx_mgpu *m, uint32_t *occ, uint32_t *seen) {
    if (!m || !occ || !seen) return fail(ARVX_ERR_INVALID, "null argument");
    const size_t lw = (size_t)((m->X + 31) / 32) * m->Y * (m->Z / m->n);
    std::vector<uint32_t> o(lw), s(lw);
    for (int i = 0; i < m->n; ++i) {
        MG_ARVX(arvx_state_download_planes(m->r[i].ctx, o.data(), s.data()));
        stripe_copy(m, i, o.data(), occ, false);
        stripe_copy(m, i, s.data(), seen, false);
    }
    return ARVX_OK;
}

static int merge_allreduce(arvx_mgpu *m) {
    for (Rank &k : m->r) {  // zero everywhere but the device's own groups
        MG_HIP(hipSetDevice(k.device));
        MG_HIP(hipMemsetAsync(k.d_full, 0, m->full_words * 4, k.stream));
        MG_ARVX(arvx_pack_occupancy_global(k.ctx, k.d_full));
    }
    return grouped(m, [&](Rank &k) {
        return ncclAllReduce(k.d_full, k.d_full, m->full_words, ncclInt32, ncclSum, k.comm,
                             k.stream);
    });
}

int arvx_mgpu_carve(arvx_mgpu *m, unsigned flags, int merge, int *fell_back) {
    if (!m) return fail(ARVX_ERR_INVALID, "null handle");
    if (merge != ARVX_MERGE_ALLREDUCE && merge != ARVX_MERGE_COMPRESSED)
        return fail(ARVX_ERR_INVALID, "unknown merge");
    if (fell_back) *fell_back = 0;
    if (m->broken)
        return fail(ARVX_ERR_RCCL, "an earlier collective failed half-way: destroy this handle");
    for (Rank &k : m->r) {  // all devices carve at once: the calls only enqueue
        MG_HIP(hipSetDevice(k.device));
        MG_HIP(hipEventRecord(k.e0, k.stream));
        MG_ARVX(arvx_carve(k.ctx, flags));
        MG_HIP(hipEventRecord(k.e1, k.stream));
    }
    bool redo = false;
    if (merge == ARVX_MERGE_ALLREDUCE) {
        if (int rc = merge_allreduce(m)) return rc;
    } else {
        const int64_t wpg = (int64_t)m->X * m->Y * 8 / 64;  // 64-bit words per 8-plane group
        for (Rank &k : m->r) {
            MG_HIP(hipSetDevice(k.device));
            MG_HIP(hipMemsetAsync(k.d_overflow, 0, sizeof(int), k.stream));
            MG_ARVX(arvx_pack_occupancy(k.ctx, k.d_local));
            MG_ARVX(arvx_occupancy_compress(k.ctx, k.d_local, (int64_t)m->local_words64, k.d_packet,
                                            m->cap64));
        }
        if (int rc = grouped(m, [&](Rank &k) {
                return ncclAllGather(k.d_packet, k.d_packets, (size_t)m->packet_words, ncclUint64,
                                     k.comm, k.stream);
            }))
            return rc;
        for (Rank &k : m->r) {
            MG_HIP(hipSetDevice(k.device));
            MG_ARVX(arvx_occupancy_expand_striped(k.ctx, k.d_packets, m->n,
                                                  (int64_t)m->local_words64, m->cap64, wpg,
                                                  k.d_full, k.d_overflow));
        }
        // the overflow flags and the packets' counts are read at the sync point that ends the
        // call anyway (no polling in between); every device received the same packets
        const int64_t S = m->packet_words;
        unsigned long long need = 0;
        for (Rank &k : m->r) {
            int of = 0;
            MG_HIP(hipSetDevice(k.device));
            MG_HIP(hipMemcpyAsync(&of, k.d_overflow, sizeof(int), hipMemcpyDeviceToHost, k.stream));
            if (&k == &m->r[0]) {
                std::vector<unsigned long long> cnt(m->n);
                for (int q = 0; q < m->n; ++q)
                    MG_HIP(hipMemcpyAsync(&cnt[q], k.d_packets + (size_t)q * S, 8,
                                          hipMemcpyDeviceToHost, k.stream));
                MG_HIP(hipStreamSynchronize(k.stream));
                for (unsigned long long c : cnt) need = c > need ? c : need;
            } else {
                MG_HIP(hipStreamSynchronize(k.stream));
            }
            redo = redo || of != 0;
        }
        if (redo) {
            if (fell_back) *fell_back = 1;
            m->cap64 = (int64_t)m->local_words64;  // back to the size that cannot overflow
            if (int rc = merge_allreduce(m)) return rc;
        } else {  // size the next packets for what this scene needed (+25 %)
            const int64_t want = (int64_t)(need + need / 4 + 16);
            m->cap64 = want < (int64_t)m->local_words64 ? want : (int64_t)m->local_words64;
        }
        m->packet_words = arvx_occupancy_packet_words((int64_t)m->local_words64, m->cap64);
    }
    float cmax = 0.f, mmax = 0.f;
    for (Rank &k : m->r) {
        MG_HIP(hipSetDevice(k.device));
        MG_HIP(hipEventRecord(k.e2, k.stream));
        MG_HIP(hipStreamSynchronize(k.stream));
        float a = 0.f, b = 0.f;
        MG_HIP(hipEventElapsedTime(&a, k.e0, k.e1));
        MG_HIP(hipEventElapsedTime(&b, k.e1, k.e2));
        cmax = a > cmax ? a : cmax;
        mmax = b > mmax ? b : mmax;
    }
    m->carve_ms = cmax;
    m->merge_ms = mmax;
    return ARVX_OK;
}

int arvx_mgpu_occupancy_device_ptr(arvx_mgpu *m, int rank, void **words, size_t *nwords) {
    if (!m || !words || rank < 0 || rank >= m->n) return fail(ARVX_ERR_INVALID, "bad argument");
    *words = m->r[rank].d_full;
    if (nwords) *nwords = m->full_words;
    return ARVX_OK;
}

int arvx_mgpu_occupancy_download(arvx_mgpu *m, uint32_t *words) {
    if (!m || !words) return fail(ARVX_ERR_INVALID, "null argument");
    Rank &k = m->r[0];
    MG_HIP(hipSetDevice(k.device));
    MG_HIP(hipMemcpyAsync(words, k.d_full, m->full_words * 4, hipMemcpyDeviceToHost, k.stream));
    MG_HIP(hipStreamSynchronize(k.stream));
    return ARVX_OK;
}

int arvx_mgpu_context(arvx_mgpu *m, int rank, arvx_ctx **ctx) {
    if (!m || !ctx || rank < 0 || rank >= m->n) return fail(ARVX_ERR_INVALID, "bad argument");
    *ctx = m->r[rank].ctx;
    return ARVX_OK;
}

int arvx_mgpu_last_times(const arvx_mgpu *m, float *carve_ms, float *merge_ms) {
    if (!m) return fail(ARVX_ERR_INVALID, "null handle");
    if (carve_ms) *carve_ms = m->carve_ms;
    if (merge_ms) *merge_ms = m->merge_ms;
    return ARVX_OK;
}

}  // extern "C"
