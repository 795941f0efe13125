// marching_cubes.hpp -- hand-off from the voxel model to the reference's
// marching cubes (src/MarchingCubes.cpp:8-31, src/MarchingCubes.h:414-578).
//
// The triangulation itself (edge/triangle tables, vertex interpolation, OFF
// writer) stays the reference's.  What it spends its time on is the walk: every
// one of the (X+1)(Y+1)(Z+1) cells is visited and reads eight voxels through
// Model::get, although only cells cut by the surface emit anything
// (Polygonise returns at once when edgeTable[cubeIndex] == 0, :486-488).  The GPU
// finds exactly those cells, in the order the reference's loops reach them
// (x outermost, z innermost, each from -1), so
//
//     for (const arvx::McCell &c : arvx::marchingCubesCells(*model))
//         ProcessVoxel(model, c.x, c.y, c.z, &mesh, threshold);
//
// appends the same vertices and faces in the same order as the triple loop of
// src/MarchingCubes.cpp:12-18.  Valid for 0 < threshold <= 1 (the reference only
// ever passes 0.5 and w is 0 or 1).
#ifndef ARVX_MARCHING_CUBES_HPP
#define ARVX_MARCHING_CUBES_HPP

#include "arvx/voxel_carving.hpp"

namespace arvx {

struct McCell {
    int x, y, z;    // base corner of the cell, each in [-1, size)
    int cubeIndex;  // Polygonise's cubeIdx (src/MarchingCubes.h:479-484), never 0 or 255
};

inline std::vector<McCell> marchingCubesCells(const Model &model, int device = 0) {
    static_assert(sizeof(McCell) == 4 * sizeof(int32_t), "McCell is the C-ABI's 4-int record");
    arvx_ctx *ctx = nullptr;
    detail::check(arvx_ctx_create(&ctx, device, model.getX(), model.getY(), model.getZ(),
                                  model.getSize()),
                  "arvx_ctx_create");
    struct Guard {
        arvx_ctx *c;
        ~Guard() { arvx_ctx_destroy(c); }
    } guard{ctx};
    detail::check(arvx_state_upload(ctx, model.state_data()), "arvx_state_upload");
    int64_t n = 0;
    detail::check(arvx_mc_cells(ctx, &n), "arvx_mc_cells");
    std::vector<McCell> cells((size_t)n);
    if (n) detail::check(arvx_mc_cells_download(ctx, (int32_t *)cells.data()),
                         "arvx_mc_cells_download");
    return cells;
}

}  // namespace arvx
#endif
