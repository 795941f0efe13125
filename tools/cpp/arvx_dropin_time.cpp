// arvx_dropin_time.cpp -- wall-clock of the drop-in C++ entry points (include/arvx/*.hpp) on one
// model size: what a caller of the reference's API pays per stage of src/main.cpp:262-303,
// host buffers in, results on the host where the reference has them.
//
//   arvx_dropin_time <scene file> <X> <Y> <Z> <voxel size> [rounds]
//   (scene file: tests/cpp/test_host.cpp; the grid in it is ignored)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <vector>

#include "arvx/marching_cubes.hpp"
#include "arvx/postprocessing.hpp"
#include "arvx/voxel_carving.hpp"

using Clock = std::chrono::steady_clock;
static double ms(Clock::time_point a, Clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
}

int main(int argc, char **argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: arvx_dropin_time <scene> X Y Z size [rounds]\n");
        return 2;
    }
    const int X = atoi(argv[2]), Y = atoi(argv[3]), Z = atoi(argv[4]);
    const float size = (float)atof(argv[5]);
    const int rounds = argc > 6 ? atoi(argv[6]) : 3;
    std::ifstream f(argv[1], std::ios::binary);
    int32_t hd[7];
    f.read((char *)hd, sizeof hd);
    const int V = hd[3], W = hd[4], H = hd[5], C = hd[6];
    float s_unused;
    f.read((char *)&s_unused, 4);
    arvx::Intrinsics intr;
    f.read((char *)intr.K, 36);
    std::vector<arvx::View> views(V);
    for (auto &v : views) f.read((char *)v.pose, 48);
    std::vector<uint8_t> masks((size_t)V * H * W * C), images((size_t)V * H * W * 3);
    f.read((char *)masks.data(), masks.size());
    f.read((char *)images.data(), images.size());
    if (!f) {
        std::fprintf(stderr, "short scene file\n");
        return 2;
    }
    for (int i = 0; i < V; ++i) {
        views[i].mask = {masks.data() + (size_t)i * H * W * C, W, H, C, (size_t)W * C};
        views[i].image = {images.data() + (size_t)i * H * W * 3, W, H, 3, (size_t)W * 3};
    }
    std::cout.setstate(std::ios::failbit);  // the LOG lines of the entry points
    for (int r = 0; r < rounds; ++r) {
        {  // the carve alone, with its result on the host (2 bits per voxel)
            auto t0 = Clock::now();
            arvx::Model model(X, Y, Z, size);
            auto t1 = Clock::now();
            arvx::carve(intr, model, views);
            auto t2 = Clock::now();
            // the model on the host: as packets where the grid allows (what the accessors read:
            // Model::sync_bits), then -- on demand only -- rebuilt as two bit planes (sync_host)
            model.sync_bits();
            auto t3 = Clock::now();
            const size_t pkb = model.packet_bytes();
            const bool inner = model.isInner(X / 2, Y / 2, Z / 2);
            model.sync_host();
            auto t4 = Clock::now();
            std::fprintf(stderr, "round %d  %dx%dx%d x %d views (C=%d): Model() %.2f | carve %.3f | "
                                 "+ state on host %.3f ms (%s, %.2f MB) | + planes rebuilt on the host %.3f ms%s\n",
                         r, X, Y, Z, V, C, ms(t0, t1), ms(t1, t2), ms(t2, t3), pkb ? "packets" : "planes",
                         pkb ? pkb / 1e6 : 2.0 * model.plane_words() * 4 / 1e6, ms(t3, t4), inner ? "" : " ");
        }
        // the pipeline of src/main.cpp:262-303: nothing comes back to the host before the
        // closure's colours and the mesh are needed there
        auto t0 = Clock::now();
        arvx::Model model(X, Y, Z, size);
        auto t1 = Clock::now();
        arvx::carve(intr, model, views);
        auto t2 = Clock::now();
        arvx::reconstructAvgColor(intr, model, views);
        auto t4 = Clock::now();
        model.handleUnseen();
        auto t5 = Clock::now();
        arvx::applyClosure(&model, 3);
        auto t6 = Clock::now();
        arvx::SimpleMesh mesh = arvx::marchingCubesMesh(&model, 0.5f);
        auto t7 = Clock::now();
        std::fprintf(stderr,
                     "         pipeline: Model() %.2f | carve %.3f | colour %.3f (%zu coloured) | "
                     "handleUnseen %.3f | closure %.3f | marching cubes mesh %.3f (%zu triangles) | "
                     "total %.2f ms\n",
                     ms(t0, t1), ms(t1, t2), ms(t2, t4), model.colored_voxels(), ms(t4, t5),
                     ms(t5, t6), ms(t6, t7), mesh.GetTriangles().size(), ms(t0, t7));
    }
    return 0;
}
