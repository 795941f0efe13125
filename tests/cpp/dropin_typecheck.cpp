// Type check of the drop-in headers against the mock declarations in tests/cpp/mock_opencv
// (and, with -Itests/cpp/mock_eigen, the mock Eigen): compiled with -fsyntax-only; never
// linked, never run.
//
// Besides the five free functions this file uses Model the way the reference's OWN replaced
// sources use it, so that code a maintainer keeps still compiles: the calls below follow
// src/VoxelCarving.cpp:45-55,93-160 (set, see, visit / visited / toWord with cv::Vec3i),
// src/ColorReconstruction.cpp:26-43,52-67 (getColors, DCLR, Vector4f arithmetic),
// src/Postprocessing3d.cpp:13-60 (get / set), src/MarchingCubes.cpp:12-29 (SimpleMesh).
#include <string>
#include <vector>

#include "opencv2/core.hpp"
cv::Mat estimatePoseFromImage(cv::Mat, cv::Mat, cv::Mat, bool);  // src/PoseEstimation.h:18

#include "Model.h"
#include "VoxelCarving.h"
#include "ColorReconstruction.h"
#include "Postprocessing3d.h"
#include "MarchingCubes.h"
#include "Benchmark.h"

void use(cv::Mat &K, cv::Mat &dist, std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks) {
    Model model(10, 10, 5, 0.028f);
    carve(K, dist, model, images, masks);
    carve(K, dist, model, images, masks, true);
    fastCarve(K, dist, model, images, masks);
    reconstructClosestColor(K, dist, model, images, masks);
    reconstructAvgColor(K, dist, model, images, masks);

    // Model as the replaced sources use it
    model.set(1, 2, 3, Vector4f(0, 0, 0, 0));
    model.see(1, 2, 3);
    cv::Vec3i voxel(0, 0, 0);
    model.visit(voxel);
    if (model.visited(cv::Vec3i(voxel(0) + 1, voxel(1), voxel(2)))) model.set(voxel, MODEL_COLOR);
    Vector4f word = model.toWord(voxel);
    Vector4f word2 = model.toWord(1, 2, 3);
    (void)word;
    (void)word2;
    if (model.get(1, 2, 3).w() != 0 && !model.isInner(1, 2, 3)) {
        model.addColor(1, 2, 3, Vector4f(255, 0, 0, 1), 0.5f);
        std::vector<DCLR> colors = model.getColors(1, 2, 3);
        DCLR min = colors.front();
        for (DCLR &c : colors)
            if (c.depth < min.depth) min = c;
        model.set(1, 2, 3, min.color);
        if (model.get(1, 2, 3) == UNSEEN_COLOR || model.get(1, 2, 3)(3) == 1) model.handleUnseen();
    }
    int n = model.getX() * model.getY() * model.getZ();
    float s = model.getSize();
    (void)n;
    (void)s;
    std::string text = model.to_string();
    model.WriteModel();
    model.WriteModel("out/model.off");

    int rc = applyClosure(&model, 3);
    (void)rc;

    SimpleMesh mesh;
    for (int x = -1; x < model.getX(); x++) ProcessVoxel(&model, x, 0, 0, &mesh, 0.5f);
    unsigned int v0 = mesh.AddVertex(Vector3f(0, 0, 0));
    mesh.AddFace(v0, v0, v0);
    mesh.AddFace(v0, v0, v0, 1, 2, 3);
    bool ok = mesh.WriteMesh("out/mesh.off", 1.0f * model.getSize(), Vector3f(0, 0, 0));
    ok = ok && mesh.WriteMesh("out/mesh.off");
    ok = ok && marchingCubes(&model);
    ok = ok && marchingCubes(&model, 1.0f, Vector3f(1, 0, 0), 0.5f, "out/mesh.off");
    (void)ok;

    Benchmark::GetInstance().NextRun("run\t", Vector4f(10, 10, 5, 0.028f));
    Benchmark::GetInstance().LogOverall(true);
    Benchmark::GetInstance().LogCarving(true);
    Benchmark::GetInstance().LogColoring(true);
    Benchmark::GetInstance().LogPostProcessing(true);
    Benchmark::GetInstance().LogMarchingCubes(true);
    std::string table = Benchmark::GetInstance().to_string();
}
