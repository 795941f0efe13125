// test_host.cpp -- exercises the C++ host layer (include/arvx/model.hpp,
// voxel_carving.hpp).  `test_host model` needs no GPU; `test_host carve <scene>
// <out> <mode>` runs the reference-shaped entry points on a scene dumped by the
// Python test and writes the resulting Model::voxels for comparison with the
// oracle.
#include <cassert>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "arvx/calibration.hpp"
#include "arvx/marching_cubes.hpp"
#include "arvx/multi_gpu.hpp"
#include "arvx/postprocessing.hpp"
#include "arvx/voxel_carving.hpp"

using arvx::Model;
using arvx::Vec3i;
using arvx::Vec4f;

#define EXPECT(c)                                                       \
    do {                                                                \
        if (!(c)) {                                                     \
            std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, #c); \
            return 1;                                                   \
        }                                                               \
    } while (0)

static int test_model() {
    Model m(4, 3, 2, 0.5f);
    EXPECT(m.getX() == 4 && m.getY() == 3 && m.getZ() == 2 && m.getSize() == 0.5f);
    EXPECT(m.get(0, 0, 0) == Vec4f(50, 168, 141, 1));   // MODEL_COLOR
    EXPECT(m.get(-1, 0, 0) == Vec4f(0, 0, 0, 0));        // outside -> zero
    EXPECT(m.get(4, 0, 0) == Vec4f(0, 0, 0, 0));
    EXPECT(!m.isInner(1, 1, 0));                         // z-1 is outside the grid
    Model big(3, 3, 3, 1.f);
    EXPECT(big.isInner(1, 1, 1));
    EXPECT(big.pristine() && big.get(1, 1, 1).w() == 1 && big.pristine());  // reads keep it
    big.set(1, 1, 2, Vec4f(0, 0, 0, 0));
    EXPECT(!big.pristine());
    Model seen1(2, 2, 2, 1.f);
    seen1.see(0, 0, 0);
    EXPECT(!seen1.pristine());
    EXPECT(!big.isInner(1, 1, 1));
    EXPECT(big.get(1, 1, 2).w() == 0);
    // toWord swaps x/y and negates z
    Vec4f w = m.toWord(3, 2, 1);
    EXPECT(w(0) == 1.0f && w(1) == 1.5f && w(2) == -0.5f && w(3) == 1.0f);
    EXPECT(m.toWord(Vec3i(3, 2, 1)) == w);
    // colours, lists, seen
    m.set(1, 1, 1, Vec4f(10, 20, 30, 1));
    EXPECT(m.get(1, 1, 1) == Vec4f(10, 20, 30, 1));
    m.addColor(1, 1, 1, Vec4f(1, 2, 3, 1), 0.25f);
    m.addColor(1, 1, 1, Vec4f(4, 5, 6, 1), 0.5f);
    EXPECT(m.getColors(1, 1, 1).size() == 2 && m.getColors(1, 1, 1)[1].depth == 0.5f);
    EXPECT(m.getColors(0, 0, 0).empty());
    EXPECT(!m.visited(Vec3i(2, 2, 1)));
    m.visit(Vec3i(2, 2, 1));
    m.see(1, 1, 1);
    EXPECT(m.visited(Vec3i(2, 2, 1)) && m.visited(Vec3i(1, 1, 1)));
    m.set(0, 0, 0, Vec4f(0, 0, 0, 0));
    m.handleUnseen();
    EXPECT(m.get(0, 0, 0) == Vec4f(204, 0, 0, 1));      // unseen, even if carved
    EXPECT(m.get(1, 1, 1) == Vec4f(10, 20, 30, 1));     // seen keeps its colour
    EXPECT(m.get(2, 2, 1) == Vec4f(50, 168, 141, 1));
    EXPECT(m.get(3, 2, 1) == Vec4f(204, 0, 0, 1));
    std::string s = Model(1, 1, 1, 1.f).to_string();
    EXPECT(s == "z = 0:\n(50, 168, 141, 1)\n\n");
    std::puts("model ok");
    return 0;
}

// packet file (tests/test_cpp_host.py): int64 n, H, total ; total uint64 packet words ; n uint64 plane
// words.  Model::packet_bit for every bit and Model::expand_packet against the plane -- no GPU.
static int test_packet_decode(const char *path) {
    std::ifstream f(path, std::ios::binary);
    int64_t hd[3];
    f.read((char *)hd, sizeof hd);
    const int64_t n = hd[0], H = hd[1], total = hd[2];
    std::vector<uint64_t> pk((size_t)total), plane((size_t)n);
    f.read((char *)pk.data(), (std::streamsize)(pk.size() * 8));
    f.read((char *)plane.data(), (std::streamsize)(plane.size() * 8));
    if (!f) { std::fprintf(stderr, "short packet file\n"); return 2; }
    for (int64_t i = 0; i < 64 * n; ++i)
        if (Model::packet_bit(pk.data(), n, H, (size_t)i) != (bool)((plane[(size_t)(i >> 6)] >> (i & 63)) & 1u)) {
            std::fprintf(stderr, "packet_bit differs at bit %lld\n", (long long)i);
            return 1;
        }
    std::vector<uint32_t> out((size_t)(2 * n), 0x5a5a5a5au);
    Model::expand_packet(pk.data(), n, H, out.data());
    if (std::memcmp(out.data(), plane.data(), (size_t)n * 8)) { std::fprintf(stderr, "expand_packet differs\n"); return 1; }
    std::puts("packet decode ok");
    return 0;
}

// scene file: int32 X,Y,Z,V,W,H,C ; float s ; float K[9] ; V*(float pose[12]) ;
//             V*H*W*C mask bytes ; V*H*W*3 image bytes ; X*Y*Z initial state bytes
static int run_carve(const char *scene, const char *out, const char *mode) {
    std::ifstream f(scene, std::ios::binary);
    int32_t hd[7];
    f.read((char *)hd, sizeof hd);
    const int X = hd[0], Y = hd[1], Z = hd[2], V = hd[3], W = hd[4], H = hd[5], C = hd[6];
    float s;
    f.read((char *)&s, 4);
    arvx::Intrinsics intr;
    f.read((char *)intr.K, 36);
    std::vector<arvx::View> views(V);
    for (auto &v : views) f.read((char *)v.pose, 48);
    std::vector<uint8_t> masks((size_t)V * H * W * C), images((size_t)V * H * W * 3);
    f.read((char *)masks.data(), masks.size());
    f.read((char *)images.data(), images.size());
    std::vector<uint8_t> st0((size_t)X * Y * Z);
    f.read((char *)st0.data(), st0.size());
    if (!f) { std::fprintf(stderr, "short scene file\n"); return 2; }
    for (int i = 0; i < V; ++i) {
        views[i].mask = {masks.data() + (size_t)i * H * W * C, W, H, C, (size_t)W * C};
        views[i].image = {images.data() + (size_t)i * H * W * 3, W, H, 3, (size_t)W * 3};
    }
    Model model(X, Y, Z, s);
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                const uint8_t b = st0[(size_t)x + (size_t)X * (y + (size_t)Y * z)];
                if (!(b & 1)) model.set(x, y, z, Vec4f(0, 0, 0, 0));
                if (b & 2) model.see(x, y, z);
            }
    int hooks = 0;
    std::vector<arvx::McCell> cells;
    try {
        if (!std::strcmp(mode, "carve")) arvx::carve(intr, model, views);
        else if (!std::strcmp(mode, "carve_steps"))
            arvx::carve(intr, model, views, true, [&](int, Model &) { ++hooks; });
        else if (!std::strcmp(mode, "carve_devices"))  // one process, a list of devices + RCCL
            arvx::carve(intr, model, views, std::vector<int>{0}, ARVX_MERGE_COMPRESSED);
        else if (!std::strcmp(mode, "fast")) arvx::fastCarve(intr, model, views);
        else if (!std::strcmp(mode, "closest")) {
            arvx::carve(intr, model, views);
            arvx::reconstructClosestColor(intr, model, views);
            // Model::getColors (src/Model.h:147-149): the samples behind the vote.  A voxel has a
            // list exactly when the pass coloured it, and the closest sample (strict <, the
            // first view wins, src/ColorReconstruction.cpp:33-40) is the colour it got.
            size_t with_list = 0;
            for (int z = 0; z < Z; ++z)
                for (int y = 0; y < Y; ++y)
                    for (int x = 0; x < X; ++x) {
                        const std::vector<arvx::DCLR> cl = model.getColors(x, y, z);
                        const bool visited = model.get(x, y, z)(3) != 0 && !model.isInner(x, y, z);
                        if (cl.empty()) continue;
                        ++with_list;
                        if (!visited || cl.size() > (size_t)V) return 8;
                        size_t best = 0;
                        for (size_t k = 1; k < cl.size(); ++k)
                            if (cl[k].depth < cl[best].depth) best = k;
                        const Vec4f got = model.get(x, y, z);
                        if (std::memcmp(got.data(), cl[best].color.data(), 16)) return 9;
                    }
            if (!with_list) return 10;
            Model copy(model);  // (a copy has no device context: only what addColor stored)
            copy.addColor(1, 2, 3, Vec4f(1, 2, 3, 1), 0.5f);
            if (copy.getColors(1, 2, 3).size() != 1 || !copy.getColors(2, 2, 3).empty()) return 11;
        } else if (!std::strcmp(mode, "recarve_colored")) {
            // a coloured, painted model goes through the carve again: state comes back
            // as bit planes; colours and paint stay where voxels survive
            arvx::carve(intr, model, views);
            arvx::reconstructClosestColor(intr, model, views);
            model.handleUnseen();
            arvx::carve(intr, model, views);
        } else if (!std::strcmp(mode, "average_unseen")) {
            arvx::carve(intr, model, views);
            arvx::reconstructAvgColor(intr, model, views);
            model.handleUnseen();
        } else if (!std::strcmp(mode, "closure")) {  // src/main.cpp:262-299
            arvx::carve(intr, model, views);
            arvx::reconstructAvgColor(intr, model, views);
            model.handleUnseen();
            if (arvx::applyClosure(&model, 3) != 0) return 5;
            if (arvx::applyClosure(&model, 4) != -1) return 6;  // even size: skipped
        } else if (!std::strcmp(mode, "closure_mc")) {  // ... src/main.cpp:303
            arvx::carve(intr, model, views);
            arvx::reconstructAvgColor(intr, model, views);
            model.handleUnseen();
            if (arvx::applyClosure(&model, 3) != 0) return 5;
            cells = arvx::marchingCubesCells(model);
        } else if (!std::strcmp(mode, "debug_mesh")) {  // Model::WriteModel, src/Model.cpp:49-107
            arvx::carve(intr, model, views);
            arvx::reconstructClosestColor(intr, model, views);
            model.handleUnseen();
            if (!model.WriteModel(std::string(out) + ".off")) return 12;
            if (model.WriteModel("/nonexistent-dir/x.off")) return 13;
        } else if (!std::strcmp(mode, "packets")) {
            // The carved model reaches the host as two compressed packets (Model::sync_bits) where
            // the grid allows it; the accessors must answer from them exactly what they answer
            // from the planes, which exist only after somebody asks for them as planes.
            arvx::carve(intr, model, views);
            const size_t N = (size_t)X * Y * Z;
            std::vector<uint8_t> a(N), b(N);
            auto sweep = [&](std::vector<uint8_t> &dst) {
                for (int z = 0; z < Z; ++z)
                    for (int y = 0; y < Y; ++y)
                        for (int x = 0; x < X; ++x)
                            dst[(size_t)x + (size_t)X * (y + (size_t)Y * z)] =
                                (uint8_t)((model.get(x, y, z)(3) != 0) | (model.isInner(x, y, z) << 1) |
                                          (model.visited(Vec3i(x, y, z)) << 2));
            };
            sweep(a);
            if (!model.planes_pending() || model.packet_bytes() == 0) return 14;  // (answered from packets)
            const uint32_t *po = model.occ_plane(), *ps = model.seen_plane();  // now as planes
            if (model.planes_pending()) return 15;
            sweep(b);
            if (a != b) return 16;
            std::vector<uint32_t> qo(model.plane_words()), qs(model.plane_words());
            if (arvx_state_download_planes(model.device_if_any(), qo.data(), qs.data()) != ARVX_OK) return 17;
            if (std::memcmp(qo.data(), po, qo.size() * 4) || std::memcmp(qs.data(), ps, qs.size() * 4)) return 18;
            // a host-side write goes to the planes, the next carve brings packets again
            model.set(1, 1, 1, Vec4f(0, 0, 0, 0));
            arvx::carve(intr, model, views);
            if (model.get(1, 1, 1)(3) != 0 || !model.planes_pending()) return 19;
            model.handleUnseen();  // (on the device; the paint plane is derived at the next sync)
        } else if (!std::strcmp(mode, "threads")) {
            // several jobs in flight from several host threads: every thread owns its models
            // (contexts share nothing; calls on ONE context are not thread safe, calls on
            // different contexts are), runs the pipeline a few times and keeps its last model
            constexpr int kThreads = 3, kRounds = 3;
            std::vector<std::unique_ptr<Model>> last(kThreads);
            std::vector<int> rc(kThreads, 0);
            std::vector<std::thread> pool;
            for (int t = 0; t < kThreads; ++t)
                pool.emplace_back([&, t] {
                    try {
                        for (int r = 0; r < kRounds; ++r) {
                            auto m = std::make_unique<Model>(model);  // an independent copy
                            arvx::carve(intr, *m, views);
                            arvx::reconstructAvgColor(intr, *m, views);
                            m->handleUnseen();
                            if (arvx::applyClosure(m.get(), 3) != 0) rc[t] = 5;
                            last[t] = std::move(m);
                        }
                    } catch (const arvx::Error &e) {
                        std::fprintf(stderr, "thread %d: arvx::Error %d: %s\n", t, e.code, e.what());
                        rc[t] = 3;
                    }
                });
            for (auto &th : pool) th.join();
            for (int t = 0; t < kThreads; ++t)
                if (rc[t]) return rc[t];
            arvx::carve(intr, model, views);
            arvx::reconstructAvgColor(intr, model, views);
            model.handleUnseen();
            if (arvx::applyClosure(&model, 3) != 0) return 5;
            for (int t = 0; t < kThreads; ++t)
                for (int z = 0; z < Z; ++z)
                    for (int y = 0; y < Y; ++y)
                        for (int x = 0; x < X; ++x) {
                            const Vec4f a = model.get(x, y, z), b = last[t]->get(x, y, z);
                            if (std::memcmp(a.data(), b.data(), 16) ||
                                model.visited(Vec3i(x, y, z)) != last[t]->visited(Vec3i(x, y, z))) {
                                std::fprintf(stderr, "thread %d differs at %d %d %d\n", t, x, y, z);
                                return 7;
                            }
                        }
        } else { std::fprintf(stderr, "unknown mode %s\n", mode); return 2; }
    } catch (const arvx::Error &e) {
        std::fprintf(stderr, "arvx::Error %d: %s\n", e.code, e.what());
        return 3;
    }
    if (!std::strcmp(mode, "carve_steps") && hooks != V) return 4;
    std::ofstream o(out, std::ios::binary);
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                Vec4f v = model.get(x, y, z);
                o.write((const char *)v.data(), 16);
            }
    // then the seen bits, one byte each
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                char b = model.visited(Vec3i(x, y, z)) ? 1 : 0;
                o.write(&b, 1);
            }
    // then the marching-cubes cells (closure_mc): int32 count, 4 ints each
    const int32_t nc = (int32_t)cells.size();
    o.write((const char *)&nc, 4);
    o.write((const char *)cells.data(), (std::streamsize)(cells.size() * sizeof(arvx::McCell)));
    return 0;
}

// model file: int32 X,Y,Z ; float voxel size ; X*Y*Z * 4 floats (Model::voxels, x fastest).
// Writes what the reference's marchingCubes(&model, scale, t, threshold, out) writes.
static int run_mc(const char *model_path, const char *out_off, float scale, float tx, float ty,
                  float tz, float threshold) {
    std::ifstream f(model_path, std::ios::binary);
    int32_t hd[3];
    float size;
    f.read((char *)hd, sizeof hd);
    f.read((char *)&size, 4);
    const int X = hd[0], Y = hd[1], Z = hd[2];
    std::vector<float> rgba((size_t)X * Y * Z * 4);
    f.read((char *)rgba.data(), (std::streamsize)(rgba.size() * 4));
    if (!f) { std::fprintf(stderr, "short model file\n"); return 2; }
    Model model(X, Y, Z, size);
    for (int z = 0; z < Z; ++z)
        for (int y = 0; y < Y; ++y)
            for (int x = 0; x < X; ++x) {
                const float *v = &rgba[4 * ((size_t)x + (size_t)X * (y + (size_t)Y * z))];
                model.set(x, y, z, Vec4f(v[0], v[1], v[2], v[3]));
            }
    try {
        if (!arvx::marchingCubes(&model, scale, arvx::Vec3f(tx, ty, tz), threshold, out_off))
            return 7;
        // an unwritable path: false, as in the reference (src/MarchingCubes.cpp:23-26)
        if (arvx::marchingCubes(&model, scale, arvx::Vec3f(tx, ty, tz), threshold,
                                "/nonexistent_dir/mesh.off"))
            return 8;
    } catch (const arvx::Error &e) {
        std::fprintf(stderr, "arvx::Error %d: %s\n", e.code, e.what());
        return 3;
    }
    return 0;
}

static int test_calibration(const char *path) {
    double K[9];
    std::vector<double> dist;
    EXPECT(arvx::readCameraParameters(path, K, dist));
    EXPECT(K[0] == 4.9650601017248454e+02 && K[2] == 3.1217886794504733e+02 && K[1] == 0.);
    EXPECT(K[4] == 4.9678498089444867e+02 && K[5] == 2.5089544695238374e+02 && K[8] == 1.);
    EXPECT(dist.size() == 5 && dist[1] == -3.2859270046436123e-01);
    EXPECT(!arvx::readCameraParameters("/nonexistent.yml", K, dist));
    std::puts("calibration ok");
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !std::strcmp(argv[1], "model")) return test_model();
    if (argc >= 3 && !std::strcmp(argv[1], "packet_decode")) return test_packet_decode(argv[2]);
    if (argc == 3 && !std::strcmp(argv[1], "calibration")) return test_calibration(argv[2]);
    if (argc == 5 && !std::strcmp(argv[1], "carve")) return run_carve(argv[2], argv[3], argv[4]);
    if (argc == 9 && !std::strcmp(argv[1], "mc"))
        return run_mc(argv[2], argv[3], std::stof(argv[4]), std::stof(argv[5]), std::stof(argv[6]),
                      std::stof(argv[7]), std::stof(argv[8]));
    std::fprintf(stderr, "usage: test_host model | test_host carve <scene> <out> <mode>\n");
    return 2;
}
