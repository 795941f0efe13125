// voxel_carving.hpp -- the reference's carving / colouring entry points over the
// GPU library (header only, C++17, links libarvx.so).
//
//   carve(...)                    reference src/VoxelCarving.h:19,  .cpp:60-72
//   fastCarve(...)                reference src/VoxelCarving.h:31,  .cpp:74-167
//   reconstructClosestColor(...)  reference src/ColorReconstruction.h:131, .cpp:22-46
//   reconstructAvgColor(...)      reference src/ColorReconstruction.h:142, .cpp:48-70
//
// Same names, argument order and effect on `model`.  The reference's signatures
// take cv::Mat camera matrix / distortion / images / masks and do three things
// per view before its voxel loop: ChArUco pose estimation, pose inversion and
// cv::undistort (src/VoxelCarving.cpp:25-36).  Those are third-party OpenCV
// calls and stay with the caller; a `View` carries their results.  With OpenCV
// present, include/arvx/opencv_dropin.hpp adds overloads with the reference's
// exact cv::Mat signatures that do that pre-processing and forward here.
#ifndef ARVX_VOXEL_CARVING_HPP
#define ARVX_VOXEL_CARVING_HPP

#include <functional>
#include <iostream>
#include <string>
#include <vector>

#include "arvx/arvx.h"
#include "arvx/model.hpp"

namespace arvx {

struct Image {  // undistorted u8 image: what cv::undistort returned in the reference
    const uint8_t *data = nullptr;
    int width = 0, height = 0, channels = 0;
    size_t stride = 0;  // bytes per row
};

struct View {
    float pose[12];  // world->camera, top 3x4 of estimatePoseFromImage(...).inv()
    Image image;     // BGR, used by the colour pass only
    Image mask;      // background where all channel bytes are 0
    // optional: the product intr * pose computed by the caller (e.g. by cv::gemm,
    // to be bit-identical with the reference's OpenCV build); otherwise it is
    // composed by arvx_compose_projection (fp32, unfused, left to right)
    bool has_M = false;
    float M[12] = {0};
};

struct Intrinsics {
    float K[9];  // cameraMatrix converted to CV_32F (src/VoxelCarving.cpp:29-30)
};

// called after each view when intermediateMeshes is set (the reference writes
// out/intermediate/image_<i>_mesh.off there, src/VoxelCarving.cpp:65-68); without a hook
// carve() does what the reference does (include/arvx/marching_cubes.hpp is needed for that:
// it installs the default)
using IntermediateHook = std::function<void(int view, Model &model)>;

// Timing sink: the reference brackets its stages with Benchmark::GetInstance().LogCarving(true/
// false) etc. (src/VoxelCarving.cpp:62,70,76,165; src/ColorReconstruction.cpp:24,44,50,68;
// src/Postprocessing3d.cpp; src/MarchingCubes.cpp:10,19).  The same brackets are reported here
// to a callback instead of a singleton: sink(stage, start).  include/arvx/benchmark.hpp is a
// sink that prints the reference's table.
enum Stage { kStageCarving = 0, kStageColoring, kStagePostProcessing, kStageMarchingCubes };
using TimingSink = std::function<void(Stage stage, bool start)>;
inline TimingSink &timingSink() {
    static TimingSink sink;
    return sink;
}
inline void setTimingSink(TimingSink sink) { timingSink() = std::move(sink); }

namespace detail {

// brackets nest: only the outermost pair of a stage reaches the sink (the drop-in signatures
// bracket their OpenCV pre-processing together with the carve, as the reference's carve() does)
inline void timing(Stage stage, bool start) {
    static int depth[4] = {0, 0, 0, 0};
    if (start ? depth[stage]++ != 0 : --depth[stage] != 0) return;
    if (timingSink()) timingSink()(stage, start);
}
struct StageBracket {
    Stage stage;
    explicit StageBracket(Stage s) : stage(s) { timing(s, true); }
    ~StageBracket() { timing(stage, false); }
};
inline IntermediateHook &defaultIntermediateHook() {
    static IntermediateHook hook;
    return hook;
}

// the views of one call on the model's device context: matrices, camera positions, masks
// (and colour images for the colour pass)
inline arvx_ctx *bind_views(const Intrinsics &intr, Model &model, const std::vector<View> &views,
                            bool with_images) {
    auto fail = [](const char *msg) {
        std::cerr << "LOG(ERR) - GPU: " << msg << std::endl;
        throw Error(ARVX_ERR_INVALID, msg);
    };
    if (views.empty()) fail("no views");
    arvx_ctx *ctx = model.device();
    const int V = (int)views.size();
    std::vector<float> M((size_t)V * 12), cam((size_t)V * 3);
    std::vector<const uint8_t *> masks(V), imgs(V);
    for (int i = 0; i < V; ++i) {
        if (views[i].has_M)
            for (int k = 0; k < 12; ++k) M[12 * (size_t)i + k] = views[i].M[k];
        else
            check(arvx_compose_projection(intr.K, views[i].pose, &M[12 * (size_t)i]),
                  "arvx_compose_projection");
        // cameras[i] = (pose(0,3), pose(1,3), pose(2,3), 1), src/ColorReconstruction.h:21
        cam[3 * (size_t)i] = views[i].pose[3];
        cam[3 * (size_t)i + 1] = views[i].pose[7];
        cam[3 * (size_t)i + 2] = views[i].pose[11];
        masks[i] = views[i].mask.data;
        if (views[i].mask.width != views[0].mask.width ||
            views[i].mask.height != views[0].mask.height ||
            views[i].mask.channels != views[0].mask.channels ||
            views[i].mask.stride != views[0].mask.stride)
            fail("all masks must share one size and layout");
        if (with_images) {
            const Image &im = views[i].image;
            if (!im.data || im.channels != 3 || im.width != views[0].mask.width ||
                im.height != views[0].mask.height || im.stride != views[0].image.stride)
                fail("colour images must be BGR u8 with the masks' size");
            imgs[i] = im.data;
        }
    }
    const Image &m0 = views[0].mask;
    // (the colour pass never looks at the masks: cameras only)
    check(arvx_set_views(ctx, V, M.data(), cam.data(), with_images ? nullptr : masks.data(),
                         m0.width, m0.height, m0.channels, m0.stride),
          "arvx_set_views");
    if (with_images)
        check(arvx_set_images(ctx, imgs.data(), views[0].image.stride), "arvx_set_images");
    return ctx;
}

// colour vote on the device, result into the model (model.set(x, y, z, (R, G, B, 1)) for every
// voxel that received a sample, src/ColorReconstruction.cpp:41/65)
inline void color_pass(const Intrinsics &intr, Model &model, const std::vector<View> &views,
                       int mode) {
    arvx_ctx *ctx = bind_views(intr, model, views, true);
    const bool had_colors = model.colored_voxels() != 0;
    check(arvx_color(ctx, mode), "arvx_color");
    int64_t n = 0;
    check(arvx_surface_count(ctx, &n), "arvx_surface_count");
    std::vector<int64_t> idx((size_t)n);
    std::vector<float> rgb((size_t)n * 3);
    if (n) check(arvx_surface_download(ctx, idx.data(), rgb.data()), "arvx_surface_download");
    model.set_sorted(idx, rgb.data(), 3, false);
    model.set_colors_on_device(!had_colors);
    model.set_sampled(std::move(idx), (int)views.size());  // (Model::getColors asks the device)
}

}  // namespace detail

// reference carve(): every view carves the model; `model` may already be carved.  The result
// stays on the device until somebody reads the model on the host.
inline void carve(const Intrinsics &intr, Model &model, const std::vector<View> &views,
                  bool intermediateMeshes = false, const IntermediateHook &hook = nullptr) {
    std::cout << "LOG - VC: starting carving process (version 1)." << std::endl;
    detail::timing(kStageCarving, true);
    arvx_ctx *ctx = detail::bind_views(intr, model, views, false);
    const IntermediateHook &each = hook ? hook : detail::defaultIntermediateHook();
    if (!intermediateMeshes) {
        detail::check(arvx_carve(ctx, 0), "arvx_carve");
        model.device_changed();
    } else {
        for (int i = 0; i < (int)views.size(); ++i) {  // one view at a time, :63-69
            detail::check(arvx_carve_views(ctx, i, 1, 0), "arvx_carve_views");
            model.device_changed();
            std::cout << "LOG - VC: completed carving of a single image." << std::endl;
            if (each) {
                std::cout << "LOG - VC: generating intermediate mesh for image " << i << std::endl;
                each(i, model);
            }
        }
    }
    detail::check(arvx_ctx_synchronize(ctx), "arvx_ctx_synchronize");
    detail::timing(kStageCarving, false);
    std::cout << "LOG - VC: carving complete." << std::endl;
}

// reference fastCarve(): greedy flood from voxel (0,0,0).
inline void fastCarve(const Intrinsics &intr, Model &model, const std::vector<View> &views) {
    std::cout << "LOG - VC: starting carving process (version 2)." << std::endl;
    detail::timing(kStageCarving, true);
    arvx_ctx *ctx = detail::bind_views(intr, model, views, false);
    detail::check(arvx_fast_carve(ctx), "arvx_fast_carve");
    model.device_changed();
    detail::timing(kStageCarving, false);
    std::cout << "LOG - VC: carving complete." << std::endl;
}

inline void reconstructClosestColor(const Intrinsics &intr, Model &model,
                                    const std::vector<View> &views) {
    std::cout << "LOG - CR: starting color reconstruction (closest color)." << std::endl;
    detail::timing(kStageColoring, true);
    detail::color_pass(intr, model, views, ARVX_COLOR_CLOSEST);
    detail::timing(kStageColoring, false);
    std::cout << "LOG - CR: color reconstruction finished." << std::endl;
}

inline void reconstructAvgColor(const Intrinsics &intr, Model &model,
                                const std::vector<View> &views) {
    std::cout << "LOG - CR: starting color reconstruction (average color)." << std::endl;
    detail::timing(kStageColoring, true);
    detail::color_pass(intr, model, views, ARVX_COLOR_AVERAGE);
    detail::timing(kStageColoring, false);
    std::cout << "LOG - CR: color reconstruction finished." << std::endl;
}

}  // namespace arvx
#endif
