// arvx_ctx.h -- the opaque context behind the C-ABI (include/arvx/arvx.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace arvx {

int fail_hip(hipError_t e, const char *what, const char *file, int line);
int fail_msg(int code, const char *msg);

#define ARVX_HIP(call)                                                        \
    do {                                                                      \
        hipError_t arvx_e_ = (call);                                          \
        if (arvx_e_ != hipSuccess)                                            \
            return ::arvx::fail_hip(arvx_e_, #call, __FILE__, __LINE__);      \
    } while (0)

// grow-only device buffer: results of repeated calls reuse the allocation
// (hipMalloc / hipFree of tens of MB cost more than the kernels that fill them)
struct DevPool {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4;
        const hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

struct Ctx {
    int device = 0;
    int X = 0, Y = 0, Z = 0;  // full grid
    int z0 = 0, z1 = 0;       // slab owned here
    int ze0 = 0, ze1 = 0;     // slab plus `halo` planes each side (clipped to the grid)
    int halo = 1;             // halo planes per inner side (arvx_ctx_create_slab_halo)
    int stripe_world = 1, stripe_rank = 0;  // striped slabs (arvx_ctx_create_striped)
    float s = 0.f;
    int assoc = 1;        // grouping of the M * world row sums (ARVX_ASSOC_*, arvx_device.h)
    size_t nvox = 0;      // owned voxels
    size_t nvox_ext = 0;  // voxels in the state buffer (owned + halo)

    // Kernels that wait for other workgroups inside a launch (views_strip_kernel, the streaming
    // carve) give up after seconds and leave a mark here instead of hanging the device: one word of
    // page-locked host memory the device writes directly, read by the host after its next
    // synchronisation (arvx_capi.hip, check_fault).  Never seen set by a kernel; the host's side of
    // it -- the call fails, the next one starts from clean control blocks -- is exercised by
    // tests/test_fault_gpu.py through the experiment build's arvx_experiment_mark_fault.
    unsigned *h_fault = nullptr;  // host address
    unsigned *d_fault = nullptr;  // the same word as the device sees it
    DevPool pool_vstrip;          // views_strip_kernel: ticket counters + published column counts
    size_t vstrip_key = 0;        // layout (V, strips, granules) the pool was zeroed for

    // scan_lookback_kernel (bitplane_kernels.h): ticket counter, device totals, status granules; and
    // what the host keeps beside it.  h_totals: page-locked words behind the fault word that the
    // kernels write the lists' lengths to (read at the call's one synchronisation).
    DevPool pool_compact;
    // set bits per kBitChunk words of the plane being compacted: two buffers of counts_stride
    // ints used in turn (the compaction that reads one zeroes the other: arvx_capi.hip, chunk_counts)
    DevPool pool_chunk_counts;
    size_t counts_stride = 0;
    int counts_cur = 0;
    bool counts_clean[2] = {false, false};
    unsigned long long compact_tickets = 0;  // tickets all launches so far have taken
    uint32_t compact_epoch = 0;
    long long *h_totals = nullptr, *d_totals_host = nullptr;  // 6 slots (host / device address)
    long long surf_host_count = -1;  // entries of h_surf_index / h_surf_has that are valid (-1: fetch)
    long long clo_host_count = -1;   // ... of h_clo_index

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t xstream = nullptr;  // arvx_ctx_set_exchange_stream: the occupancy hand-off (null: stream)

    // The state lives in sub-tile RECORDS (csrc/arvx_device.h), 2 bits per voxel: what every
    // stage reads and writes.  The one-byte-per-voxel plane of the C-ABI (arvx_state_upload /
    // _download / _device_ptr) is a staging buffer, allocated on demand and converted from / to
    // the records inside those calls; bit2 of uploaded bytes (voxel painted UNSEEN_COLOR by a
    // host Model) is kept beside the records as one bit plane (`paint`).
    uint16_t *d_rec = nullptr;   // records of planes ze0..ze1-1 (+ padding to whole coarse tiles)
    size_t rec_bytes = 0;
    bool rec_valid = false;      // d_rec holds the current state (else: fresh_pending)
    // every call that changes occupied / seen bits counts here: what was derived from the state
    // (the host hand-off's packets) is current while the count it was taken at still stands
    unsigned long long state_seq = 1;
    // the colour pass's bit planes of the state (occupied; seen by no view), kept for a closure that
    // follows: valid while state_seq == planes_seq; planes_unseen: handleUnseen has run since (the
    // records' occupancy is the occupancy plane | the never-seen plane)
    DevPool pool_state_planes;  // occupancy plane, then never-seen plane (planes_words words each)
    size_t planes_words = 0;
    unsigned long long planes_seq = 0;
    bool planes_ok = false, planes_unseen = false;
    DevPool pool_state_packets;  // arvx_state_download_packets: occupancy | seen, worst-case size each
    unsigned long long packets_seq = 0;
    bool packets_valid = false;
    long long packet_need[2] = {0, 0};
    // lazy state (arvx_device.h): after the carve of a fresh model the coarse tiles it settled
    // as a whole exist only as their code in `ccode`; their records are not written
    DevPool pool_ccode;
    DevPool pool_cstate;         // per coarse tile: settled by earlier carves (CarveParams::cstate)
    size_t cstate_tiles = 0;     // ... valid for this many tiles of the current layout (0: not)
    bool lazy = false;
    uint8_t *d_state = nullptr;  // byte staging, planes ze0..ze1-1 (lazy)
    uint8_t *owned() const { return d_state + (size_t)(z0 - ze0) * X * Y; }
    DevPool pool_paint;          // bit plane (bitplane_kernels.h layout) over planes ze0..ze1-1
    bool paint_valid = false;    // some voxel is painted: the plane takes part in the stages
    int ncu = 0;                 // compute units of the device (cached)
    int carve_seq = 0;           // parity of the undecided-list counters (carve_coarse_kernel)
    size_t carve_layout = 0;     // d_coarse layout those counters were zeroed for
    // the streaming carve (carve_stream_kernels.h): control block + list entries + item queues
    void *d_stream = nullptr;
    size_t stream_bytes = 0;
    size_t stream_layout = 0;    // layout the control block was zeroed for (0: zero it again)
    unsigned carve_epoch = 0;    // tag of the latest streaming launch's granules
    void *d_timeline = nullptr;  // ARVX_TIMELINE diagnostic builds only
    int64_t timeline_n = 0;
    int timeline_rec = 32;  // bytes per record
    bool fresh_pending = false;  // a fresh model that exists only as this flag (neither form valid)
    void *d_flood_rec = nullptr;  // records of the "carvable" plane of arvx_fast_carve
    DevPool pool_flood_code;      // ... and the codes of the coarse tiles that exist only as a code
    size_t flood_rec_bytes = 0;
    void *d_coarse = nullptr;    // coarse pre-pass masks of the carve kernel
    size_t coarse_bytes = 0;
    unsigned long long *d_stats = nullptr;
    // times a list total had to be fetched from the device because the page-locked word still
    // read -1 at the synchronisation (host_total, arvx_capi.hip); expected: never
    unsigned long long host_total_fallbacks = 0;
    void *d_scratch = nullptr;  // work buffer of the calls on `stream`
    size_t scratch_bytes = 0;
    // work buffer of the hand-off calls (arvx_occupancy_compress): they may run on `xstream`
    // BESIDE a set_views / carve on `stream`, so they never touch d_scratch
    DevPool pool_xscratch;
    void *d_flood = nullptr;  // work buffer of arvx_fast_carve, kept between calls
    size_t flood_bytes = 0;

    // views
    bool views_ready = false;    // matrices + bit planes + tables: the carve can run
    bool cameras_ready = false;  // matrices (+ camera positions): enough for the colour pass
    bool has_campos = false;
    int V = 0, W = 0, H = 0;
    int bgWords = 0, satStride = 0;
    int satW = 0, satH = 0;  // summed-area table: satH = H + 1 rows of satW >= W + 1 entries
    float *d_M = nullptr;
    float *d_campos = nullptr;
    uint32_t *d_bg = nullptr;
    uint16_t *d_sat = nullptr;
    std::vector<float> h_M, h_campos;

    // colour pass
    uint8_t *d_images = nullptr;  // V x H x W x 3, BGR
    bool images_ready = false;
    int *d_surf_index = nullptr;      // compacted flat indices (slab-local)
    float4 *d_surf_rgba = nullptr;    // r, g, b, has-sample flag per surface voxel
    float *d_surf_depth = nullptr;    // minimum sample depth per surface voxel
    uint8_t *d_surf_has = nullptr;    // 1 if the voxel received >= 1 sample
    int64_t surf_count = 0;           // occupied non-inner voxels found (planes c_lo .. c_hi)
    // Index lists (d_surf_index, d_clo_index) hold flat indices over the context's planes
    // ze0 .. ze1 - 1 (x + X * (y + Y * (z - ze0))): a slab with a halo wider than one plane
    // colours / closes some halo planes too, because its OWNED cells and voxels need their
    // neighbours' results (stage_ranges in arvx_capi.hip).  What the download calls return is
    // the owned part, relative to the owned planes.
    std::vector<int> h_surf_index;    // host copy (ascending)
    std::vector<uint8_t> h_surf_has;
    bool color_ready = false;
    // the list's plane and its index (bitplane_kernels.h, SparseWord) over the owned planes
    DevPool pool_col_bits, pool_col_rank;

    // closure (dilation) result: filled voxels, ascending index
    int *d_clo_index = nullptr;
    void *d_clo_rgba = nullptr;  // float4 per filled voxel
    int64_t clo_count = 0;
    bool closure_ready = false;
    int closure_unseen = 0;
    int closure_radius = 0;
    std::vector<int> h_clo_index;
    DevPool pool_clo_bits, pool_clo_rank;  // the filled voxels' plane and index (SparseWord)

    // marching-cubes hand-off: active cells (x, y, z, cube index), reference order
    void *d_mc_cells = nullptr;
    int64_t mc_count = 0;
    bool mc_ready = false;
    // storage behind the d_surf_* / d_clo_* / d_mc_cells views (kept until destroy)
    DevPool pool_surf_index, pool_surf_rgb, pool_surf_depth, pool_surf_has, pool_clo_index,
        pool_clo_rgba, pool_mc_cells, pool_raw_masks, pool_mesh_verts, pool_mesh_rgb, pool_mesh_off;
    int64_t mesh_tris = 0;  // triangles of the last arvx_mc_mesh
    void release_pools() {
        pool_surf_index.release();
        pool_surf_rgb.release();
        pool_surf_depth.release();
        pool_surf_has.release();
        pool_clo_index.release();
        pool_clo_rgba.release();
        pool_mc_cells.release();
        pool_raw_masks.release();
        pool_flood_code.release();
        pool_mesh_verts.release();
        pool_mesh_rgb.release();
        pool_mesh_off.release();
        pool_xscratch.release();
        pool_vstrip.release();
        vstrip_key = 0;
        pool_compact.release();
        pool_chunk_counts.release();
        counts_stride = 0;
        pool_state_packets.release();
        pool_state_planes.release();
        planes_ok = false;
        packets_valid = false;
        compact_tickets = 0;
        pool_paint.release();
        pool_ccode.release();
        pool_cstate.release();
        cstate_tiles = 0;
        pool_col_bits.release();
        pool_col_rank.release();
        pool_clo_bits.release();
        pool_clo_rank.release();
    }
    void free_mc() {
        d_mc_cells = nullptr;
        mc_count = 0;
        mc_ready = false;
    }

    void free_closure() {
        d_clo_index = nullptr;
        d_clo_rgba = nullptr;
        clo_count = 0;
        closure_ready = false;
        clo_host_count = -1;
        h_clo_index.clear();
    }

    void free_views() {
        if (d_M) (void)hipFree(d_M);
        if (d_campos) (void)hipFree(d_campos);
        if (d_bg) (void)hipFree(d_bg);
        if (d_sat) (void)hipFree(d_sat);
        d_M = d_campos = nullptr;
        d_bg = nullptr;
        d_sat = nullptr;
        views_ready = false;
        cameras_ready = false;
    }
    void free_surface() {
        free_closure();
        d_surf_index = nullptr;
        d_surf_rgba = nullptr;
        d_surf_depth = nullptr;
        d_surf_has = nullptr;
        color_ready = false;
        surf_count = 0;
        surf_host_count = -1;
        h_surf_index.clear();
        h_surf_has.clear();
    }
    void free_color() {
        free_surface();
        if (d_images) (void)hipFree(d_images);
        d_images = nullptr;
        images_ready = false;
    }
};

}  // namespace arvx

struct arvx_ctx : arvx::Ctx {};
