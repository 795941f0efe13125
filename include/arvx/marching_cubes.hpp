// marching_cubes.hpp -- the reference's marchingCubes(Model*, scale, translation, threshold,
// outFileName) (src/MarchingCubes.h:596, src/MarchingCubes.cpp:8-31) over the GPU library:
// same name, arguments, defaults, return value, log lines and -- byte for byte -- the same
// OFF file (pinned against the reference's own Data/box_dataset/generated_models/1.off,
// tests/test_mc_off.py).
//
// The reference visits all (X+1)(Y+1)(Z+1) cells and reads eight voxels per cell through
// Model::get, although only cells cut by the surface emit anything (Polygonise returns at once
// when edgeTable[cubeIdx] == 0, src/MarchingCubes.h:486-488).  Here the GPU finds exactly those
// cells (arvx_mc_cells: cube index and position, in the order the reference's loops reach
// them -- x outermost, z innermost, each from -1) and the host triangulates that list with the
// reference's rules:
//   * a cut edge's vertex snaps to the corner that belongs to the model when the other one has
//     w == 0, and takes that corner's colour (VertexInterp, :428-441); otherwise position and
//     colour are interpolated, except that MODEL_COLOR / UNSEEN_COLOR are never blended in
//     (:443-467);
//   * every triangle gets three fresh vertices (ProcessVoxel, :561-568), and its third corner's
//     COLOUR is the second corner's (`i + 1` instead of `i + 2`, :506 -- kept);
//   * face colour = round((c0 + c1 + c2) / 3) per channel in fp32 (MeanColorFloats, :414-416);
//   * OFF text as SimpleMesh::WriteMesh prints it (:60-87): default ostream float format,
//     vertex * scaleFactor + translation evaluated in fp32.
#ifndef ARVX_MARCHING_CUBES_HPP
#define ARVX_MARCHING_CUBES_HPP

#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "arvx/host_pool.hpp"
#include "arvx/mc_tables.hpp"
#include "arvx/voxel_carving.hpp"

namespace arvx {

struct Triangle {  // src/MarchingCubes.h:19-31
    unsigned int idx0, idx1, idx2;
    unsigned int r, g, b;
};

// (the two arrays live in recycled page-locked memory, include/arvx/host_pool.hpp: a mesh of
// 1.3 M triangles is 77 MB that the device writes in the arrays' own layout)
class SimpleMesh {  // src/MarchingCubes.h:33-92
   public:
    unsigned int AddVertex(const Vec3f &vertex) {
        m_vertices.push_back(vertex);
        return (unsigned int)m_vertices.size() - 1;
    }
    unsigned int AddFace(unsigned int idx0, unsigned int idx1, unsigned int idx2, unsigned int r = 0,
                         unsigned int g = 0, unsigned int b = 0) {
        m_triangles.push_back(Triangle{idx0, idx1, idx2, r, g, b});
        return (unsigned int)m_triangles.size() - 1;
    }
    HostVector<Vec3f> &GetVertices() { return m_vertices; }
    HostVector<Triangle> &GetTriangles() { return m_triangles; }

    // The reference ends every line with std::endl (one flush per line); '\n' and one flush at
    // the end give the same bytes.
    bool WriteMesh(const std::string &filename, float scaleFactor = 1.f,
                   Vec3f translation = Vec3f(0, 0, 0)) {
        std::ofstream outFile(filename);
        if (!outFile.is_open()) return false;
        outFile << "OFF" << '\n';
        outFile << m_vertices.size() << " " << m_triangles.size() << " 0" << '\n';
        for (const Vec3f &p : m_vertices) {
            // fp32 multiply, then fp32 add (two statements: no contraction into an fma)
            volatile float px = p.x() * scaleFactor, py = p.y() * scaleFactor,
                           pz = p.z() * scaleFactor;
            const float ox = px + translation.x(), oy = py + translation.y(),
                        oz = pz + translation.z();
            outFile << ox << " " << oy << " " << oz << '\n';
        }
        for (const Triangle &t : m_triangles)
            outFile << "3 " << t.idx0 << " " << t.idx1 << " " << t.idx2 << " " << t.r << " " << t.g
                    << " " << t.b << '\n';
        outFile.flush();
        const bool ok = outFile.good();
        outFile.close();
        return ok;
    }

   private:
    HostVector<Vec3f> m_vertices;
    HostVector<Triangle> m_triangles;
};

struct McCell {
    int x, y, z;    // base corner of the cell, each in [-1, size)
    int cubeIndex;  // Polygonise's cubeIdx (src/MarchingCubes.h:479-484), never 0 or 255
};

// The cells marchingCubes() triangulates, in its visiting order, found on the GPU.
// A corner counts as outside when its w < threshold (:481); the reference's own pipeline only
// ever holds w in {0, 1} and passes 0.5.
inline std::vector<McCell> marchingCubesCells(Model &model, float threshold = 0.5f) {
    static_assert(sizeof(McCell) == 4 * sizeof(int32_t), "McCell is the C-ABI's 4-int record");
    // w >= 0 everywhere the grid is, and 0 outside it: with threshold <= 0 no corner is outside
    if (!(threshold > 0.f)) return {};
    bool on_device = false;
    const std::vector<uint8_t> inside = model.inside_state(threshold, on_device);
    arvx_ctx *ctx = nullptr;
    struct Guard {
        arvx_ctx *c;
        ~Guard() {
            if (c) arvx_ctx_destroy(c);
        }
    } guard{nullptr};
    if (on_device) {  // "inside" is the occupancy the model's context already holds
        ctx = model.device_for_reading();
    } else {  // fractional w: a thresholded plane of its own, in a context of its own
        detail::check(arvx_ctx_create(&ctx, 0, model.getX(), model.getY(), model.getZ(),
                                      model.getSize()),
                      "arvx_ctx_create");
        guard.c = ctx;
        detail::check(arvx_state_upload(ctx, inside.data()), "arvx_state_upload");
    }
    int64_t n = 0;
    detail::check(arvx_mc_cells(ctx, &n), "arvx_mc_cells");
    std::vector<McCell> cells((size_t)n);
    if (n) detail::check(arvx_mc_cells_download(ctx, (int32_t *)cells.data()),
                         "arvx_mc_cells_download");
    return cells;
}

namespace mc {

struct Interp {
    Vec3f coord, color;
};

inline bool is_default_color(const Vec3f &c) {  // MODEL_COLOR / UNSEEN_COLOR, src/Model.h:90-91
    return c == Vec3f(50, 168, 141) || c == Vec3f(204, 0, 0);
}

// src/MarchingCubes.h:428-468
inline Interp vertex_interp(float threshold, const Vec3f &point0, const Vec4f &val0,
                            const Vec3f &point1, const Vec4f &val1) {
    Interp ret;
    if (val0.w() == 0.0f && val1.w() != 0.0f) {  // corner 0 is not part of the model
        ret.color = Vec3f(val1.x(), val1.y(), val1.z());
        ret.coord = point1;
        return ret;
    }
    if (val0.w() != 0.0f && val1.w() == 0.0f) {
        ret.color = Vec3f(val0.x(), val0.y(), val0.z());
        ret.coord = point0;
        return ret;
    }
    const float f = (val0.w() == val1.w()) ? 0.5f : (threshold - val0.w()) / (val1.w() - val0.w());
    const float g = 1 - f;
    const Vec3f col0(val0.x(), val0.y(), val0.z()), col1(val1.x(), val1.y(), val1.z());
    for (int k = 0; k < 3; ++k) {
        volatile float a = g * point0[k], b = f * point1[k];  // (1-f)*p0 + f*p1, unfused
        ret.coord[k] = a + b;
    }
    if (is_default_color(col0)) {
        ret.color = col1;
    } else if (is_default_color(col1)) {
        ret.color = col0;
    } else {
        for (int k = 0; k < 3; ++k) {
            volatile float a = g * col0[k], b = f * col1[k];
            ret.color[k] = a + b;
        }
    }
    return ret;
}

// MeanColorFloats, src/MarchingCubes.h:414-416
inline unsigned int mean_color(float c1, float c2, float c3) {
    volatile float s = c1 + c2;
    s = s + c3;
    return (unsigned int)std::round(s / 3);
}

}  // namespace mc

// One cell: Polygonise + ProcessVoxel of the reference (src/MarchingCubes.h:478-511, 532-578).
// Returns whether the cell emitted a triangle.
inline bool ProcessVoxel(Model *model, int x, int y, int z, SimpleMesh *mesh, float threshold) {
    static const int corner[8][3] = {{1, 0, 0}, {0, 0, 0}, {0, 1, 0}, {1, 1, 0},
                                     {1, 0, 1}, {0, 0, 1}, {0, 1, 1}, {1, 1, 1}};  // :537-552
    Vec4f val[8];
    Vec3f p[8];
    int cubeIdx = 0;
    for (int i = 0; i < 8; ++i) {
        const int cx = x + corner[i][0], cy = y + corner[i][1], cz = z + corner[i][2];
        val[i] = model->get(cx, cy, cz);
        p[i] = Vec3f((float)cx, (float)cy, (float)cz);
        if (val[i].w() < threshold) cubeIdx |= 1 << i;
    }
    const int edges = mc::edge_mask(cubeIdx);
    if (edges == 0) return false;
    mc::Interp vert[12];
    for (int e = 0; e < 12; ++e)
        if (edges & (1 << e))
            vert[e] = mc::vertex_interp(threshold, p[e % 8], val[e % 8], p[mc::kSecondCorner[e]],
                                        val[mc::kSecondCorner[e]]);
    const char *tri = mc::kTriangles[cubeIdx];
    bool any = false;
    for (; tri[0]; tri += 3) {
        const mc::Interp &a = vert[mc::hex_digit(tri[0])], &b = vert[mc::hex_digit(tri[1])],
                         &c = vert[mc::hex_digit(tri[2])];
        const unsigned int h0 = mesh->AddVertex(a.coord);
        const unsigned int h1 = mesh->AddVertex(b.coord);
        const unsigned int h2 = mesh->AddVertex(c.coord);
        // the third corner's colour is the second's: src/MarchingCubes.h:506 reads [i + 1]
        const Vec3f &c0 = a.color, &c1 = b.color, &c2 = b.color;
        mesh->AddFace(h0, h1, h2, mc::mean_color(c0.x(), c1.x(), c2.x()),
                      mc::mean_color(c0.y(), c1.y(), c2.y()), mc::mean_color(c0.z(), c1.z(), c2.z()));
        any = true;
    }
    return any;
}

// The mesh marchingCubes() writes, without writing it.  When every w of the model is 0 or 1
// (all the reference's own pipeline produces) the triangles come from the device
// (arvx_mc_mesh: state, colour list and closure result are already there); a model with
// fractional w is triangulated on the host from the device's cell list.
inline SimpleMesh marchingCubesMesh(Model *model, float threshold = 0.5f) {
    SimpleMesh mesh;
    if (!(threshold > 0.f)) return mesh;
#ifdef ARVX_HOST_TRACE  // diagnostic builds: where the host time of this call goes
    using TraceClock = std::chrono::steady_clock;
    auto trace_t = TraceClock::now();
    auto trace = [&](const char *what) {
        const auto now = TraceClock::now();
        std::fprintf(stderr, "    [mesh] %-18s %.3f ms\n", what,
                     std::chrono::duration<double, std::milli>(now - trace_t).count());
        trace_t = now;
    };
#define ARVX_TRACE(what) trace(what)
#else
#define ARVX_TRACE(what) (void)0
#endif
    bool on_device = false;
    (void)model->inside_state(threshold, on_device);
    ARVX_TRACE("inside_state");
    if (!on_device) {
        for (const McCell &c : marchingCubesCells(*model, threshold))
            ProcessVoxel(model, c.x, c.y, c.z, &mesh, threshold);
        return mesh;
    }
    arvx_ctx *ctx = model->device_for_reading();
    // painted voxels: derived on the device when the paint is exactly "not seen", else carried
    // by the byte plane (bit2)
    const bool painted = model->painted();
    if (painted && !model->paint_is_unseen()) {
        const std::vector<uint8_t> st = model->byte_state();
        detail::check(arvx_state_upload(ctx, st.data()), "arvx_state_upload");
        model->set_colors_on_device(false);
    }
    if (!model->colors_on_device()) {
        std::vector<int64_t> idx;
        std::vector<float> rgb;
        (void)model->sorted_colors(idx, rgb);  // (w is 0 or 1 here)
        detail::check(arvx_colors_upload(ctx, (int64_t)idx.size(), idx.data(), rgb.data()),
                      "arvx_colors_upload");
        model->set_colors_on_device(true);
    }
    ARVX_TRACE("context + colours");
    int64_t n = 0;
    detail::check(arvx_mc_mesh(ctx, (painted && model->paint_is_unseen()) ? 1 : 0, &n),
                  "arvx_mc_mesh");
    ARVX_TRACE("arvx_mc_mesh");
    // the device writes both arrays in the mesh's own layout: a triangle's three corners as
    // nine floats, its face as (3t, 3t+1, 3t+2, r, g, b)
    static_assert(sizeof(Vec3f) == 3 * sizeof(float), "Vec3f is three packed floats");
    static_assert(sizeof(Triangle) == 6 * sizeof(uint32_t), "Triangle is six packed uints");
    HostVector<Vec3f> &mv = mesh.GetVertices();
    HostVector<Triangle> &mt = mesh.GetTriangles();
    mv.resize((size_t)n * 3);  // (recycled memory, not zeroed: host_pool.hpp)
    mt.resize((size_t)n);
    ARVX_TRACE("resize");
    if (n) detail::check(arvx_mc_mesh_download_faces(ctx, mv[0].data(), &mt[0].idx0),
                         "arvx_mc_mesh_download_faces");
    ARVX_TRACE("download");
#undef ARVX_TRACE
    return mesh;
}

inline bool marchingCubes(Model *model, float scale = 1.0f, Vec3f translation = Vec3f(0, 0, 0),
                          float threshold = 0.5f, std::string outFileName = "out/mesh.off") {
    std::cout << "LOG - MC: starting to process Voxels." << std::endl;
    detail::timing(kStageMarchingCubes, true);
    SimpleMesh mesh = marchingCubesMesh(model, threshold);
    detail::timing(kStageMarchingCubes, false);
    std::cout << "LOG - MC: voxel processing completed.\n Writing mesh..." << std::endl;
    volatile float factor = scale * model->getSize();
    if (!mesh.WriteMesh(outFileName, factor, translation)) {
        std::cout << "ERR - MC: unable to write output file!" << std::endl;
        return false;
    }
    std::cout << "LOG - MC: Mesh written, marchingCubes completed." << std::endl;
    return true;
}

namespace detail {
// carve(..., intermediateMeshes = true) without a caller-supplied hook writes what the
// reference writes after every view (src/VoxelCarving.cpp:65-68): the model so far, moved
// along x by i * (X + 2) voxels.  (The directory is the caller's to create, as in the
// reference: src/main.cpp:47.)
inline bool install_default_intermediate_hook() {
    defaultIntermediateHook() = [](int i, Model &model) {
        marchingCubes(&model, 1.0f, Vec3f(i * (model.getX() + 2) * model.getSize(), 0, 0), 0.5f,
                      (std::string)("out/intermediate/image_" + std::to_string(i) + "_mesh.off"));
    };
    return true;
}
static const bool default_intermediate_hook_installed = install_default_intermediate_hook();
}  // namespace detail

}  // namespace arvx
#endif
