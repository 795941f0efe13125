// arvx_bench6.cpp -- the reference's `-c=6` benchmark (src/main.cpp:306-440) on
// top of the GPU library: the same eight runs (model sizes, carve version, and --
// as in the reference, whatever the label says -- average colouring for all of
// them), the same stage sequence carve -> colour -> handleUnseen -> applyClosure,
// and the same table layout as Benchmark::to_string() (src/Benchmark.h:131-150).
// Marching cubes is not part of this library: its column is n/a.
//
// Times are wall clock around each C++ entry point, i.e. they INCLUDE context
// creation, host->device copies of masks/images/state and the copy back -- the
// honest end-to-end cost of the drop-in functions, not kernel time.
//
//   arvx_bench6 <scene file>      (format: tests/cpp/test_host.cpp)
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <vector>

#include "arvx/postprocessing.hpp"
#include "arvx/voxel_carving.hpp"

using Clock = std::chrono::steady_clock;

struct Run {
    std::string name;
    int x, y, z;
    float size;
    int version;
    double carving = 0, coloring = 0, post = 0, overall = 0;
    size_t occupied = 0;
};

static double ms(Clock::time_point a, Clock::time_point b) {
    return std::chrono::duration<double, std::milli>(b - a).count();
}

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

    // src/main.cpp:353-432
    std::vector<Run> runs = {
        {"Small, V1, avg. coloring\t", 10, 10, 5, 0.028f, 1},
        {"Medium, V1, avg. coloring\t", 50, 50, 25, 0.0056f, 1},
        {"Medium, V1, closest coloring\t", 50, 50, 25, 0.0056f, 1},
        {"Large, V1, avg. coloring\t", 100, 100, 50, 0.0028f, 1},
        {"Small, V2, avg. coloring\t", 10, 10, 5, 0.028f, 2},
        {"Medium, V2, avg. coloring\t", 50, 50, 25, 0.0056f, 2},
        {"Medium, V2, closest coloring\t", 50, 50, 25, 0.0056f, 2},
        {"Large, V2, avg. coloring\t", 100, 100, 50, 0.0028f, 2},
    };
    // one untimed warm-up (HIP runtime and code-object load), like the reference's
    // "dummy" run that its table skips (src/Benchmark.h:143)
    {
        arvx::Model warm(10, 10, 5, 0.028f);
        arvx::carve(intr, warm, views);
    }
    for (auto &r : runs) {
        auto t0 = Clock::now();
        arvx::Model model(r.x, r.y, r.z, r.size);
        auto a = Clock::now();
        if (r.version == 1) arvx::carve(intr, model, views);
        else arvx::fastCarve(intr, model, views);
        auto b = Clock::now();
        arvx::reconstructAvgColor(intr, model, views);  // all eight runs, as in the reference
        auto c = Clock::now();
        model.handleUnseen();
        auto d = Clock::now();
        arvx::applyClosure(&model, 3);
        auto e = Clock::now();
        r.carving = ms(a, b);
        r.coloring = ms(b, c);
        r.post = ms(d, e);
        r.overall = ms(t0, e);
        for (int z = 0; z < r.z; ++z)
            for (int y = 0; y < r.y; ++y)
                for (int x = 0; x < r.x; ++x) r.occupied += model.get(x, y, z).w() != 0;
    }
    std::ostringstream ss;
    ss << std::endl << "Benchmark (all times in milliseconds)" << std::endl;
    ss << "Name\t\t\t\t" << "|  Model size (x,y,z, voxel size)\t" << "|  Carving time\t"
       << "|  Coloring time\t" << "|  Postprocessing time\t" << "|  Marching cubes time\t"
       << "|  Overall time" << std::endl;
    for (int i = 0; i < 177; i++) ss << "-";
    ss << std::endl;
    for (const auto &r : runs)
        ss << r.name << "|  " << r.x << "x" << r.y << "x" << r.z << ", " << r.size << "\t\t\t|  "
           << r.carving << "\t|  " << r.coloring << "\t\t|  " << r.post << "\t\t|  " << "n/a"
           << "\t\t|  " << r.overall << std::endl;
    std::cout << ss.str();
    for (const auto &r : runs) std::cout << "occupied " << r.occupied << std::endl;
    return 0;
}
