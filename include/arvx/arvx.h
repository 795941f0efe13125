/*
 * arvx.h -- C-ABI of the MI355X (gfx950) voxel-carving library, libarvx.so.
 *
 * This is the drop-in boundary for ONE path of alxfox/AR_Voxel_Project: the
 * dense silhouette carve, the greedy carve and the per-voxel colour vote,
 * i.e. the bodies of
 *     carve(...)                    reference src/VoxelCarving.h:19
 *     fastCarve(...)                reference src/VoxelCarving.h:31
 *     reconstructClosestColor(...)  reference src/ColorReconstruction.h:131
 *     reconstructAvgColor(...)      reference src/ColorReconstruction.h:142
 *     Model::handleUnseen()         reference src/Model.cpp:36-47
 * The reference's third-party pre-processing (ChArUco pose estimation,
 * cv::undistort, cv::imread) stays on the caller's side: the boundary takes
 * the world->camera matrices and the already undistorted u8 images that the
 * reference feeds its voxel loops (src/VoxelCarving.cpp:25-36).
 *
 * Plain C types only.  Every call returns ARVX_OK or an error code and never
 * throws; arvx_last_error() gives the text.  A context owns one voxel grid (or
 * one Z slab of a grid) on one GPU; calls on one context are not thread safe.
 *
 * State plane: one byte per voxel, index x + X*(y + Y*z) (Model::flatten,
 * reference src/Model.h:104-106), bit0 = occupied (voxels[i].w != 0),
 * bit1 = seen (seen[i]).  That is the EXCHANGE form of arvx_state_upload / _download; on the
 * device the state lives as 2 bits per voxel (or use the bit planes of
 * arvx_state_upload_planes / _download_planes, 8x smaller, which is what the C++ layer does).
 *
 * Streams: every call enqueues on the context's stream (arvx_ctx_set_stream) and calls on one
 * context must not overlap -- with ONE exception: the occupancy hand-off (arvx_pack_occupancy
 * [_global], arvx_occupancy_compress, _expand[_striped]) may run on the exchange stream
 * (arvx_ctx_set_exchange_stream) while arvx_set_views_device / arvx_carve of the NEXT job run
 * on the main stream; those calls share no work buffer with the main stream.
 */
#ifndef ARVX_H
#define ARVX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARVX_VERSION 100

enum {
    ARVX_OK = 0,
    ARVX_ERR_INVALID = 1, /* bad argument (null pointer, size, range) */
    ARVX_ERR_HIP = 2,     /* a HIP runtime call failed */
    ARVX_ERR_STATE = 3,   /* call order (e.g. carve before set_views) */
    ARVX_ERR_NOMEM = 4
};

#define ARVX_OCC 1u
#define ARVX_SEEN 2u

/* flags of arvx_carve / arvx_carve_views */
#define ARVX_CARVE_NO_CULL 1u /* evaluate every voxel in every view (ablation) */
#define ARVX_CARVE_STATS 2u   /* fill the counters read by arvx_get_stats */
#define ARVX_CARVE_FUSED 8u   /* one kernel for rectangle tests and per-voxel work (A/B) */
#define ARVX_CARVE_STREAM 16u    /* a fresh model: the one-launch streaming carve.  It loses to the
                                    three-launch chain and is compiled only into
                                    -DARVX_EXPERIMENTS builds (libarvx_experiments.so, A/B and
                                    tests); the shipped library answers ARVX_ERR_INVALID */
#define ARVX_CARVE_NO_STREAM 32u /* never the streaming carve */
#define ARVX_CARVE_FILTER 64u    /* project through an fp32 filter first, exact fall-back for voxels within
                                    the filter's error of a rounding tie: the same model bit for bit, but
                                    1.16-1.23x SLOWER than the exact kernel (EXPERIMENTS.md round 5) --
                                    compiled only into -DARVX_EXPERIMENTS builds like ARVX_CARVE_STREAM */

/* colour modes: reference -color=1 / -color=2 (src/main.cpp:276-288) */
#define ARVX_COLOR_CLOSEST 0
#define ARVX_COLOR_AVERAGE 1

typedef struct arvx_ctx arvx_ctx;

typedef struct arvx_stats {
    uint64_t subtiles;        /* 16x8x8 sub-tiles visited */
    uint64_t subtiles_carved; /* decided "all carved" by one view's rectangle test */
    uint64_t subtile_views_mixed; /* (sub-tile, view) pairs evaluated per voxel */
    uint64_t subtile_views_total; /* (sub-tile, view) pairs classified */
    uint64_t surface_voxels;  /* colour pass: occupied non-inner voxels */
    uint64_t reserved[3];
    uint64_t host_total_fallbacks; /* list totals the host had to fetch from the device because the
                                      page-locked word did not show the kernel's store at the
                                      synchronisation (colour pass, closure, cell list, mesh);
                                      expected 0, counted over the context's life */
} arvx_stats;

int arvx_version(void);
const char *arvx_last_error(void);
int arvx_device_count(int *count);
/* How the four products of a row of M * world are summed -- the cv::gemm call behind
 * `intr * pose * world`, reference src/VoxelCarving.cpp:19, which takes cv::gemm's generic path
 * (GEMMSingleMul<float, double>: fp32 inputs, fp64 products and sums):
 *   ARVX_ASSOC_LEFT   ((p0 + p1) + p2) + p3   four accumulators, `(s0 + s1 + s2 + s3) * alpha`
 *   ARVX_ASSOC_RIGHT  p0 + ((p1 + p2) + p3)   `s0 += s1 + s2 + s3`
 * The two differ by double rounding in about one voxel-view in 1e8.  Which one an OpenCV build
 * uses cannot be checked in an image without OpenCV: the default (LEFT) follows the OpenCV 4.x
 * source as recalled, it is a property of a context that can be changed at run time, and
 * include/arvx/opencv_dropin.hpp settles it with one cv::gemm call at first use (see also
 * tools/pin_with_opencv.py).  arvx_projection_assoc: what new contexts start with. */
#define ARVX_ASSOC_RIGHT 0
#define ARVX_ASSOC_LEFT 1
int arvx_projection_assoc(void);
int arvx_set_projection_assoc(int assoc);

/* ---- context --------------------------------------------------------- */

/* Grid X*Y*Z voxels of edge `voxel_size` on HIP device `device`: the
 * arguments of the reference's Model constructor (src/Model.cpp:9). */
int arvx_ctx_create(arvx_ctx **out, int device, int X, int Y, int Z,
                    float voxel_size);
/* Same grid, but this context holds only planes z_begin <= z < z_end
 * (one Z slab of a multi-GPU split). State buffers cover the slab only. */
int arvx_ctx_create_slab(arvx_ctx **out, int device, int X, int Y, int Z,
                         float voxel_size, int z_begin, int z_end);
/* The same with `halo` planes kept (and recomputed by the carve: carving is a pure function of
 * the voxel position) on each side that lies inside the grid; arvx_ctx_create_slab keeps 1.
 * The stages after the carve look further: a voxel is on the surface by its six neighbours
 * (colour pass: 1 plane), the closure's box has radius r = (kernel_size - 1) / 2 and averages
 * its neighbours' COLOURS (r + 1 planes), and the mesh of a slab's cells needs the colours and
 * fills of plane z_begin - 1 (2 planes with colours, r + 2 after a closure: 3 for the
 * reference's kernel size 3).  With enough halo every slab computes its part of
 * carve -> colour -> handleUnseen -> closure -> mesh without any exchange, and the parts put
 * together are the whole-grid result (ar_voxel_project_amd/sharding.py: merge_closure,
 * merge_mesh; tests/test_sharded_stages_gpu.py).  A call that lacks halo planes says so. */
int arvx_ctx_create_slab_halo(arvx_ctx **out, int device, int X, int Y, int Z,
                              float voxel_size, int z_begin, int z_end, int halo);
/* Same grid, striped over `world` GPUs for load balance: with the planes cut
 * into groups of 8, this context holds groups rank, rank+world, rank+2*world, ...
 * back to back (Z must be a multiple of 8).  Carve, state up/download and
 * arvx_pack_occupancy_global work on striped contexts; the colour pass and
 * arvx_fast_carve need neighbouring planes and take contiguous slabs only. */
int arvx_ctx_create_striped(arvx_ctx **out, int device, int X, int Y, int Z,
                            float voxel_size, int world, int rank);
/* (The HIP stream a destroyed context owned is kept in a per-device pool and handed to the next
 * context: creating and destroying a stream costs milliseconds on this stack.  A caller that
 * resets the device (hipDeviceReset) between contexts must not have any context alive across
 * the reset -- the pooled streams die with the device and the library does not notice.) */
int arvx_ctx_destroy(arvx_ctx *ctx);
/* Launch on a caller-owned hipStream_t (borrowed); NULL = context's own. */
int arvx_ctx_set_stream(arvx_ctx *ctx, void *hip_stream);
/* A stream of its own for the occupancy hand-off (arvx_pack_occupancy[_global],
 * arvx_occupancy_compress / _expand[_striped]): a multi-GPU job runs the exchange of job k --
 * pack, compress, the collective, expand -- beside the carve of job k + 1.  The caller orders
 * the two streams with events: the pack reads the state the carve wrote (it must wait for the
 * carve, and the next carve for the pack).  NULL: back on the context's stream. */
int arvx_ctx_set_exchange_stream(arvx_ctx *ctx, void *hip_stream);
int arvx_ctx_synchronize(arvx_ctx *ctx);
/* The row-sum grouping of THIS context (see arvx_projection_assoc); takes effect with the next
 * carve / colour pass. */
int arvx_ctx_set_projection_assoc(arvx_ctx *ctx, int assoc);
int arvx_ctx_projection_assoc(const arvx_ctx *ctx, int *assoc);
/* Voxels held by this context (slab). */
int arvx_ctx_voxels(const arvx_ctx *ctx, int64_t *count);

/* ---- views ------------------------------------------------------------ */

/* M = K(3x3) * Rt(3x4), float, each element a0*b0 + a1*b1 + a2*b2 evaluated
 * left to right without FMA: what `intr * pose` yields through cv::gemm in
 * the reference (src/VoxelCarving.cpp:19).  Host helper; callers that have
 * OpenCV should pass OpenCV's own product instead. */
int arvx_compose_projection(const float K[9], const float Rt[12], float M[12]);

/* The V camera views.
 *   M      V*12 floats, row-major 3x4, M = intr * pose[0:3,:]  (world->pixel)
 *   campos V*3 floats, the translation column of the world->camera matrix
 *          (what the reference's colour pass calls `cameras[i]`,
 *          src/ColorReconstruction.h:21); may be NULL if arvx_color is unused
 *   masks  V host pointers; each an undistorted mask, H rows of `stride`
 *          bytes, C interleaved u8 channels (reference: 3, BGR); a pixel is
 *          background iff all C bytes are 0 (src/VoxelCarving.cpp:49-50)
 * Copies everything to the device; the host buffers may be freed on return.
 * masks == NULL: cameras only (M, campos, W, H) for a colour pass that follows -- arvx_color
 * never looks at the masks (the reference undistorts them there and drops them,
 * src/ColorReconstruction.h:25-28); arvx_carve / arvx_fast_carve then need a full
 * arvx_set_views first. */
int arvx_set_views(arvx_ctx *ctx, int V, const float *M, const float *campos,
                   const uint8_t *const *masks, int W, int H, int C,
                   size_t stride);
/* Same, masks already in device memory: V images back to back, tightly
 * packed (stride = W*C).  No host synchronisation. */
int arvx_set_views_device(arvx_ctx *ctx, int V, const float *M,
                          const float *campos, const void *dev_masks, int W,
                          int H, int C);
/* cv::undistort(src, dst, cameraMatrix, distCoeffs) for V images of one size on the device --
 * what the reference does to every mask and image before its voxel loops
 * (src/VoxelCarving.cpp:35-36; src/ColorReconstruction.h:22-27): map in double precision,
 * 5-bit fixed-point bilinear weights, constant-0 border (csrc/undistort_kernels.h; restated
 * from OpenCV 4.x, parity UNPINNED -- no OpenCV in this image).  K: cameraMatrix row-major;
 * dist: k1 k2 p1 p2 [k3 [k4 k5 k6]] (ndist = 4, 5 or 8).  src/dst: V host images, H rows of
 * `stride` bytes, C interleaved u8 channels; dst may equal src.  The _device form takes and
 * leaves tightly packed images in device memory (dst != src), ready for
 * arvx_set_views_device. */
int arvx_undistort(arvx_ctx *ctx, int V, const uint8_t *const *src, int W, int H, int C,
                   size_t stride, const double K[9], const double *dist, int ndist,
                   uint8_t *const *dst);
int arvx_undistort_device(arvx_ctx *ctx, int V, const void *dev_src, int W, int H, int C,
                          const double K[9], const double *dist, int ndist, void *dev_dst);
/* Undistorted colour images for arvx_color: V host pointers, BGR u8, H rows
 * of `stride` bytes, same W/H as the masks. */
int arvx_set_images(arvx_ctx *ctx, const uint8_t *const *images, size_t stride);

/* ---- state ------------------------------------------------------------ */

/* All voxels occupied, none seen: the state a fresh Model has. */
int arvx_state_reset(arvx_ctx *ctx);
int arvx_state_upload(arvx_ctx *ctx, const uint8_t *state);
int arvx_state_download(arvx_ctx *ctx, uint8_t *state);
/* The same state as two bit planes, 8x smaller than the byte plane each: occupied and seen,
 * rows padded to whole 32-bit words -- word (z, y, k) holds voxels x = 32 k .. 32 k + 31 of row
 * (y, z), bit x % 32; ceil(X / 32) * Y * (owned planes) words per plane (for X % 32 == 0 this
 * is the flat packing of arvx_pack_occupancy).  This is the form the device keeps (2 bits per
 * voxel) and the host-side arvx::Model mirrors, so a carve result crosses PCIe as N / 4 bytes. */
int arvx_state_upload_planes(arvx_ctx *ctx, const uint32_t *occ, const uint32_t *seen);
int arvx_state_download_planes(arvx_ctx *ctx, uint32_t *occ, uint32_t *seen);
/* The same two planes of the owned voxels as compressed PACKETS -- what a carved model mostly is:
 * long runs of empty space (all-zero words), solid interior (all-one words) and a thin shell of
 * mixed words.  The planes are read as 64-bit words in flat order (bit i % 64 of word i / 64 =
 * voxel i = x + X * (y + Y * z), z relative to the first owned plane; needs X % 32 == 0 and
 * X * Y % 64 == 0), and a packet is the layout of arvx_occupancy_compress (below):
 *   [0] number of mixed words   [1, 1+nb) bitmap of the all-one words   [1+nb, 1+2nb) bitmap of the
 *   mixed words   [1+2nb, H) per group of 64 words the number of mixed words before it (u32)
 *   [H, H+need) the mixed words in order           nb = ceil(n / 64), H = 1 + 2 nb + ceil(nb / 2)
 * so a reader finds word i without expanding anything: all-one / mixed by its bit in group i / 64,
 * a mixed word at H + offset[group] + popcount(mixed bits below it).  1024^3 sphere: 24 MB instead of
 * 268 MB across PCIe.  arvx_state_packet_geometry: n (words64) and H (header_words).
 * arvx_state_download_packets: each host buffer has room for H + cap words; *need = the packet's
 * number of mixed words.  A packet that needed more than its cap is copied up to the cap only: call
 * again with larger buffers -- the device keeps both packets while the state is unchanged, so the
 * second call only copies.  Host side: include/arvx/model.hpp answers get / isInner / visited
 * straight from the packets. */
int arvx_state_packet_geometry(arvx_ctx *ctx, int64_t *words64, int64_t *header_words);
int arvx_state_download_packets(arvx_ctx *ctx, uint64_t *occ_packet, int64_t occ_cap, uint64_t *seen_packet,
                                int64_t seen_cap, int64_t *occ_need, int64_t *seen_need);
/* Model::handleUnseen() on the device state (reference src/Model.cpp:36-47): every voxel that
 * no view saw becomes occupied (the reference paints it UNSEEN_COLOR, w = 1); seen bits stay. */
int arvx_handle_unseen(arvx_ctx *ctx);
/* Page-lock caller memory (hipHostRegister) so that uploads / downloads into it run at full
 * PCIe rate instead of through the runtime's staging buffers; optional.  Plain wrappers, so
 * that host code needs no HIP headers. */
int arvx_host_register(void *ptr, size_t bytes);
int arvx_host_unregister(void *ptr);
/* A device-side SNAPSHOT of the owned part of the state as one byte per voxel, for consumers
 * that read it in place.  The context keeps the state as 2-bit records; this call converts
 * them into a buffer the context owns (enqueued on the context's stream: synchronise or use
 * that stream before reading).  The pointer is valid until the context is destroyed, its
 * CONTENT only until the next call that changes the state (carve, fast carve, handleUnseen,
 * closure, uploads, reset): query again afterwards.  Writes through it are not seen by the
 * library -- use arvx_state_upload[_planes]. */
int arvx_state_device_ptr(arvx_ctx *ctx, void **ptr, size_t *bytes);
/* Slab contexts keep one halo plane on each side that lies inside the grid
 * (z_begin-1 and z_end), because the colour pass needs the six neighbours of
 * every voxel (reference Model::isInner, src/Model.h:126-132).  arvx_carve
 * recomputes the halo planes itself (carving is a pure function of the voxel
 * position), arvx_state_reset makes them fresh, arvx_state_upload leaves them
 * untouched.  A caller that uploads a pre-carved model into a slab passes the
 * neighbouring planes here (X*Y bytes each; NULL = leave as is). */
int arvx_state_upload_halo(arvx_ctx *ctx, const uint8_t *plane_below,
                           const uint8_t *plane_above);
/* Pack bit0 (occupied) of the slab into 32-bit words, voxel i -> bit i%32 of
 * word i/32, written to device memory dev_words (>= ceil(n/32) words). */
int arvx_pack_occupancy(arvx_ctx *ctx, void *dev_words);
/* Same bits, written at their place in the word plane of the WHOLE grid
 * (ceil(X*Y*Z/32) words at dev_global_words): plane z starts at word z*X*Y/32.
 * Works for contiguous and striped slabs; needs X*Y to be a multiple of 64. */
int arvx_pack_occupancy_global(arvx_ctx *ctx, void *dev_global_words);

/* Compressed form of a slab's packed occupancy for the end-of-carve exchange (SURVEY 8e:
 * over xGMI the exchange, not the carve, sets the pace of a multi-GPU step).  Most 64-bit
 * words of the packed plane are all-zero or all-one: a packet holds the number of mixed
 * words, two bitmaps (all-one, mixed), per 64 words the number of mixed words before them,
 * and the mixed words themselves -- arvx_occupancy_packet_words(n, cap) 64-bit words for a
 * slab of n words with room for cap mixed ones.
 *   arvx_occupancy_compress: device words of ONE slab -> one packet.
 *   arvx_occupancy_expand: the `world` packets of an all-gather, back to back -> the plain
 *   words of every other rank's slab at q * n in dev_full_words (equal slabs; 16-byte
 *   aligned); a packet whose
 *   slab had more than cap mixed words sets *dev_overflow = 1 and is skipped: fall back to
 *   the plain all-gather of the packed words then. */
int64_t arvx_occupancy_packet_words(int64_t n_words64, int64_t cap_words64);
int arvx_occupancy_compress(arvx_ctx *ctx, const void *dev_words, int64_t n_words64,
                            void *dev_packet, int64_t cap_words64);
int arvx_occupancy_expand(arvx_ctx *ctx, const void *dev_packets, int world, int self_rank,
                          int64_t n_words64, int64_t cap_words64, void *dev_full_words,
                          int *dev_overflow);
/* The same for striped slabs (arvx_ctx_create_striped: rank q owns the 8-plane groups q,
 * q + world, ...; a packet holds a rank's planes in its local order, arvx_pack_occupancy):
 * word i of rank q goes to ((i / wpg) * world + q) * wpg + i % wpg of the whole grid's plane,
 * wpg = words_per_group = X*Y*8/64 (must be even).  Every rank's packet is expanded, the
 * caller's own included. */
int arvx_occupancy_expand_striped(arvx_ctx *ctx, const void *dev_packets, int world,
                                  int64_t n_words64, int64_t cap_words64, int64_t words_per_group,
                                  void *dev_full_words, int *dev_overflow);
/* The rank's own packet straight from the state (no arvx_pack_occupancy in between): the
 * context's planes in local order -> dev_packet, the same packet arvx_pack_occupancy +
 * arvx_occupancy_compress produce, in two launches instead of four and one pass over the words
 * less.  dev_full_words (may be null): the whole grid's word plane -- the rank's own words are
 * stored at their place there as well (contiguous and striped slabs), so that the expansion of the
 * gathered packets can leave the caller's out (arvx_occupancy_expand's self_rank /
 * arvx_occupancy_expand_striped_others).  Needs X % 32 == 0 and X*Y % 64 == 0. */
int arvx_occupancy_pack_compress(arvx_ctx *ctx, void *dev_packet, int64_t cap_words64,
                                 void *dev_full_words);
/* arvx_occupancy_expand_striped that leaves rank self_rank's packet out (-1: none). */
int arvx_occupancy_expand_striped_others(arvx_ctx *ctx, const void *dev_packets, int world,
                                         int self_rank, int64_t n_words64, int64_t cap_words64,
                                         int64_t words_per_group, void *dev_full_words,
                                         int *dev_overflow);

/* ---- hot path --------------------------------------------------------- */

/* Dense carve over all views: reference carve(), src/VoxelCarving.cpp:60-72. */
int arvx_carve(arvx_ctx *ctx, unsigned flags);
/* Views first <= i < first+count only; count = 1 is the reference's
 * single-view carve (src/VoxelCarving.cpp:23-58), used for its per-view
 * intermediate meshes (:65-68). */
int arvx_carve_views(arvx_ctx *ctx, int first, int count, unsigned flags);
/* Greedy carve: reference fastCarve(), src/VoxelCarving.cpp:74-167.
 * Needs the whole grid in one context (no slab).  The context keeps the work
 * buffer of the flood fill (about 1.25 bytes per voxel) for later calls. */
int arvx_fast_carve(arvx_ctx *ctx);
/* Colour vote on the occupied surface voxels (reference
 * reconstructClosestColor / reconstructAvgColor). Result stays on the device
 * until arvx_export_model / arvx_surface_download. */
int arvx_color(arvx_ctx *ctx, int mode);
/* Number of coloured voxels of the last arvx_color, then their flat indices
 * and RGB values (3 floats each, integral), ascending index order. */
int arvx_surface_count(arvx_ctx *ctx, int64_t *count);
int arvx_surface_download(arvx_ctx *ctx, int64_t *index, float *rgb);
/* Smallest sample depth of each coloured voxel (same order), as the reference
 * computes it: (float)cv::norm(cameras[i] - world), src/ColorReconstruction.h:59. */
int arvx_surface_depth_download(arvx_ctx *ctx, float *depth);
/* The per-voxel colour lists behind the vote -- what the reference's voxel_pass appends with
 * Model::addColor (src/ColorReconstruction.h:44-60, src/Model.h:142-149) and getColors returns:
 * for each of n voxels (flat index over the context's own planes, any order) V samples in view
 * order, out[k * V + i] = view i's sample of voxel k: valid = 1 and (r, g, b) = the pixel the
 * voxel projects to in image i with depth = (float)cv::norm(cameras[i] - world), or valid = 0
 * when it projects outside the image.  Needs the cameras (with campos) and the images of the
 * colour pass (arvx_set_views, arvx_set_images); the occupancy is not looked at -- which voxels
 * a pass visits (occupied, not inner) is the caller's knowledge (arvx_surface_download). */
typedef struct arvx_color_sample {
    uint8_t r, g, b, valid;
    float depth;
} arvx_color_sample;
/* views: the samples per voxel `out` has room for -- must be the context's number of views
 * (ARVX_ERR_INVALID otherwise: a caller that sized its buffer for another set of views is told so
 * instead of being written past its end). */
int arvx_color_samples(arvx_ctx *ctx, int64_t n, const int64_t *index, int views,
                       arvx_color_sample *out);

/* Replace the device-side sparse colours by a caller-supplied list (n voxels,
 * ascending flat index, 3 floats RGB each, w = 1): lets a host Model that was
 * coloured elsewhere go through arvx_closure / arvx_export_model. */
int arvx_colors_upload(arvx_ctx *ctx, int64_t n, const int64_t *index, const float *rgb);

/* Morphological closure: reference applyClosure(model, kernel_size),
 * src/Postprocessing3d.cpp:4-100 (called at src/main.cpp:297-299 after the
 * colour pass and handleUnseen).  As in the reference the erosion half never
 * removes anything (it tests w < 0), so this is one dilation with a
 * kernel_size^3 box whose new voxels get the mean RGBA of their occupied
 * neighbours.  apply_unseen != 0: the model is taken as it is after
 * handleUnseen().  State bytes may carry bit2 (voxel painted UNSEEN_COLOR by
 * a host Model; kept beside the records as a bit plane until the next carve, plane upload or
 * reset).  Whole-grid contexts and slabs with at least r + 1 halo planes
 * (arvx_ctx_create_slab_halo): a slab fills -- and reports -- the voxels of its own planes.
 * The filled voxels become occupied;
 * arvx_export_model(ctx, ., same apply_unseen) then returns the closed model. */
int arvx_closure(arvx_ctx *ctx, int kernel_size, int apply_unseen);
int arvx_closure_count(arvx_ctx *ctx, int64_t *count);
/* Filled voxels, ascending flat index, 4 floats RGBA each. */
int arvx_closure_download(arvx_ctx *ctx, int64_t *index, float *rgba);
/* The same with 32-bit indices (a grid has at most INT_MAX voxels, Model::flatten). */
int arvx_closure_download32(arvx_ctx *ctx, int32_t *index, float *rgba);

/* Marching-cubes hand-off.  The reference's marchingCubes() visits every cell
 * (x,y,z) of [-1,X) x [-1,Y) x [-1,Z), x outermost and z innermost
 * (src/MarchingCubes.cpp:12-18); a cell emits triangles only when its cube index
 * -- bit i set when corner i (order of src/MarchingCubes.h:537-552) is not part
 * of the model, :479-484 -- is neither 0 nor 255 (:486-488).  arvx_mc_cells
 * finds those cells on the device from the current occupancy and returns how
 * many there are; arvx_mc_cells_download copies them out, 4 ints per cell
 * (x, y, z, cube index), in the reference's visiting order, so that calling
 * ProcessVoxel on this list alone builds the same mesh.  Slab contexts list the
 * cells whose upper plane they own (global z; the last slab also those above the
 * grid); striped contexts are refused.  Valid for 0 < threshold <= 1, which
 * covers the 0.5 every reference call site passes. */
int arvx_mc_cells(arvx_ctx *ctx, int64_t *count);
int arvx_mc_cells_download(arvx_ctx *ctx, int32_t *cells);

/* The triangles marchingCubes() builds from the current model (before WriteMesh scales and
 * writes them), for the models the device holds: every w is 0 or 1, threshold in (0, 1].
 * Triangle t has the three fresh vertices 3t, 3t+1, 3t+2 (src/MarchingCubes.h:561-568) --
 * `verts` gets 9 floats per triangle, voxel units -- and the face colour face_rgb[3t..3t+2]
 * (rounded mean with the reference's `i + 1` quirk, :506).  Voxel colours: UNSEEN_COLOR where
 * handleUnseen painted (apply_unseen != 0: every never-seen voxel; bit2 of uploaded bytes),
 * else the closure's colour, else the colour pass's, else MODEL_COLOR.  Whole-grid contexts and
 * slabs with enough halo planes (arvx_ctx_create_slab_halo): a slab builds the triangles of the
 * cells arvx_mc_cells lists for it (global z).  Runs arvx_mc_cells itself; the order is the
 * reference's. */
int arvx_mc_mesh(arvx_ctx *ctx, int apply_unseen, int64_t *triangles);
int arvx_mc_mesh_download(arvx_ctx *ctx, float *verts, uint32_t *face_rgb);
/* The same triangles with whole face records, 6 uints per triangle: the vertex numbers 3t,
 * 3t+1, 3t+2, then r, g, b -- the reference's Triangle (src/MarchingCubes.h:19-31), so that a
 * host mesh takes both arrays without a conversion loop. */
int arvx_mc_mesh_download_faces(arvx_ctx *ctx, float *verts, uint32_t *faces);

/* Model::voxels as the reference would hold it after carve [+ colour]
 * [+ handleUnseen]: n*4 floats (RGBA), n = slab voxels. */
int arvx_export_model(arvx_ctx *ctx, float *rgba, int apply_unseen);

int arvx_get_stats(arvx_ctx *ctx, arvx_stats *out);

/* Self-test hook: evaluates the kernel's shared-reciprocal division and the IEEE
 * `/` on n host operand triples; out gets 4 floats per triple:
 * (a0/b fast, a1/b fast, a0/b IEEE, a1/b IEEE). */
int arvx_selftest_divide(arvx_ctx *ctx, int64_t n, const float *a0, const float *a1,
                         const float *b, float *out);

/* Self-test hooks for a host that has OpenCV (include/arvx/opencv_dropin.hpp runs them at first
 * use): the raw values the kernels compute for n voxels xyz (3 ints each, global coordinates).
 *   arvx_selftest_project: rows_uv = 3 n floats a0 a1 a2 -- the fp32 rows of M * toWord(x, y, z)
 *     with the context's grouping, what `intr * pose * world` holds (src/VoxelCarving.cpp:19) --
 *     followed by 2 n floats u = a0 / a2, v = a1 / a2 (:20).
 *   arvx_selftest_depth: (float)cv::norm(cameras[i] - toWord(x, y, z)) for campos = the first
 *     three components of cameras[i] (src/ColorReconstruction.h:59). */
int arvx_selftest_project(arvx_ctx *ctx, int64_t n, const float M[12], float voxel_size,
                          const int32_t *xyz, float *rows_uv);
int arvx_selftest_depth(arvx_ctx *ctx, int64_t n, const float campos[3], float voxel_size,
                        const int32_t *xyz, float *depth);

/* Self-test hook: what arvx_set_views[_device] derived for view `view` -- its background bit
 * plane (bit i of word i / 32 = pixel i is background; (W*H + 31) / 32 words) and the
 * summed-area table the rectangle tests read: (H + 1) rows of *ld two-byte entries, entry
 * [Y][X], X <= W, = foreground pixels in rows < Y and columns < X, modulo 2^16.  Either buffer may
 * be null; *ld is always set (size the table buffer after a first call with both null). */
int arvx_selftest_view_tables(arvx_ctx *ctx, int view, uint32_t *bg_bits, uint16_t *table, int *ld);

/* Self-test hook: the kernels' one-instruction pixel rounding (v_cvt_rpi_i32_f32) against
 * std::round on every float in (-0.5, 2^24], i.e. every quotient that can fall inside an
 * image; *mismatches must come back 0. */
int arvx_selftest_round(arvx_ctx *ctx, int64_t *mismatches);

#ifdef __cplusplus
}
#endif
#endif
