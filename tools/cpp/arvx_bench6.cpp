// arvx_bench6.cpp -- the reference's `-c=6` benchmark (src/main.cpp:306-440) on
// top of the GPU library: the same eight runs (model sizes, carve version, and --
// as in the reference, whatever the label says -- average colouring for all of
// them), the same stage sequence carve -> colour -> handleUnseen -> applyClosure,
// and the same table layout as Benchmark::to_string() (src/Benchmark.h:131-150).
// and marchingCubes to out/bench/mesh_*.off, timed by the library's own stage brackets
// through arvx::Benchmark (include/arvx/benchmark.hpp), the reference's Benchmark singleton.
//
// Times are wall clock around each C++ entry point: they include the host->device copies of
// masks / images and whatever a stage brings back to the host, not just kernel time.
//
//   arvx_bench6 <scene file>      (format: tests/cpp/test_host.cpp)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>

#include "bench6.hpp"

int main(int argc, char **argv) {
    if (argc != 2) {
        std::fprintf(stderr, "usage: arvx_bench6 <scene file>\n");
        return 2;
    }
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
    std::cout << "LOG - Benchmark: images read." << std::endl;
    std::cout << "LOG - Benchmark: masks read." << std::endl;

    const std::vector<bench6::Run> runs = bench6::run(intr, views, 1.0f, arvx::Vec3f(0, 0, 0));
    for (const auto &r : runs) std::cout << "occupied " << r.occupied << std::endl;
    return 0;
}
