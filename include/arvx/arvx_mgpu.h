/*
 * arvx_mgpu.h -- the dense carve over the GPUs of one node from ONE process (libarvx_mgpu.so:
 * libarvx.so + RCCL).  north_star: "the grid shards naturally along the Z slab axis across the
 * 8 GPUs of one node with a single RCCL all-reduce of the boolean occupancy at the end".
 *
 * Every voxel's result depends only on its own coordinates and the read-only views (reference
 * src/VoxelCarving.cpp:39-55), so device r simply carves its planes: with the planes cut into
 * groups of 8, device r owns groups r, r + n, r + 2n, ... (arvx_ctx_create_striped: surface
 * voxels cluster in z, stripes balance the load).  No data-path exchange is needed to carve;
 * ONE collective at the end leaves the bit-packed occupancy of the WHOLE grid (voxel i -> bit
 * i % 32 of word i / 32, arvx_pack_occupancy) on every device:
 *   ARVX_MERGE_ALLREDUCE    ncclAllReduce(SUM) of int32 words over planes that are zero outside
 *                           a device's own groups (RCCL has no bitwise OR; exactly one device
 *                           holds non-zero words at any position)
 *   ARVX_MERGE_COMPRESSED   ncclAllGather of compressed packets (arvx_occupancy_compress /
 *                           _expand_striped: about 1/7 of the bytes); a packet that overflows
 *                           makes the call fall back to the all-reduce and reports it
 * The communicator comes from ncclCommInitAll, one stream per device, the collective inside a
 * ncclGroupStart / ncclGroupEnd.  n = 1 works without any link.
 *
 * X * Y must be a multiple of 64 and Z a multiple of 8 * n.
 */
#ifndef ARVX_MGPU_H
#define ARVX_MGPU_H

#include "arvx/arvx.h"

#ifdef __cplusplus
extern "C" {
#endif

#define ARVX_MERGE_ALLREDUCE 0
#define ARVX_MERGE_COMPRESSED 1
#define ARVX_ERR_RCCL 5

typedef struct arvx_mgpu arvx_mgpu;

int arvx_mgpu_create(arvx_mgpu **out, const int *devices, int n, int X, int Y, int Z,
                     float voxel_size);
int arvx_mgpu_destroy(arvx_mgpu *m);
int arvx_mgpu_devices(const arvx_mgpu *m, int *n);
/* The views, replicated to every device (arguments of arvx_set_views). */
int arvx_mgpu_set_views(arvx_mgpu *m, int V, const float *M, const float *campos,
                        const uint8_t *const *masks, int W, int H, int C, size_t stride);
/* Fresh model on every device, or the state as two bit planes of the WHOLE grid (layout of
 * arvx_state_upload_planes): every device takes its planes. */
int arvx_mgpu_state_reset(arvx_mgpu *m);
int arvx_mgpu_state_upload_planes(arvx_mgpu *m, const uint32_t *occ, const uint32_t *seen);
/* reference carve() over all views on all devices + the merge of the occupancy.  On return
 * every device holds the merged plane; *fell_back (may be NULL) tells whether a compressed
 * merge had to be redone as an all-reduce. */
int arvx_mgpu_carve(arvx_mgpu *m, unsigned flags, int merge, int *fell_back);
/* The merged packed occupancy: device address on device `rank` / a copy on the host
 * (ceil(X*Y*Z / 32) words). */
int arvx_mgpu_occupancy_device_ptr(arvx_mgpu *m, int rank, void **words, size_t *nwords);
int arvx_mgpu_occupancy_download(arvx_mgpu *m, uint32_t *words);
/* Occupied and seen planes of the whole grid on the host (layout of
 * arvx_state_download_planes), collected from the devices that own them. */
int arvx_mgpu_state_download_planes(arvx_mgpu *m, uint32_t *occ, uint32_t *seen);
/* The single-GPU context of device `rank` (its striped planes), e.g. for timing. */
int arvx_mgpu_context(arvx_mgpu *m, int rank, arvx_ctx **ctx);
/* Wall-clock split of the last arvx_mgpu_carve, milliseconds: carve kernels (slowest device),
 * packing + collective + expansion. */
int arvx_mgpu_last_times(const arvx_mgpu *m, float *carve_ms, float *merge_ms);

#ifdef __cplusplus
}
#endif
#endif
