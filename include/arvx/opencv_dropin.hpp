// opencv_dropin.hpp -- the reference's exact cv::Mat signatures, for a build of
// the reference's main.cpp against this library.
//
//   void carve(cv::Mat&, cv::Mat&, Model&, std::vector<cv::Mat>&, std::vector<cv::Mat>&, bool)
//   void fastCarve(cv::Mat&, cv::Mat&, Model&, std::vector<cv::Mat>&, std::vector<cv::Mat>&)
//   void reconstructClosestColor(...same five...) / reconstructAvgColor(...)
//   (reference src/VoxelCarving.h:19,31; src/ColorReconstruction.h:131,142)
//
// Only compiled where OpenCV (core, calib3d) is installed; this image has none: the repo's
// tests TYPE-CHECK this header against mock declarations of the cv::Mat members it uses
// (tests/cpp/mock_opencv, tests/test_cpp_host.py) and RUN its self-pinning against mock
// cv::gemm / cv::norm / cv::undistort implementations (tests/cpp/test_selfpin.cpp).
//
// Self-pinning.  Three pieces of the path's arithmetic live inside OpenCV and cannot be checked
// where OpenCV is absent: how cv::gemm groups the four products of a row of M * world, cv::norm
// of a Vec4f, and cv::undistort's remap.  On its first use this header asks OpenCV itself
// (dropin::self_pin): one cv::gemm call on a known-answer voxel selects the library's grouping
// (arvx_set_projection_assoc), a few more voxels and cv::norm values are compared bit for bit
// with what the kernels compute (arvx_selftest_project / _depth), and cv::undistort of a small
// ramp is compared with arvx_undistort.  The findings are logged ("LOG - PIN: ...") and kept
// in dropin::self_pin(); a mismatch is a warning, not an error: pre-processing is done by OpenCV
// itself here, so only the grouping changes what the kernels compute.  The per-view
// pre-processing is done with OpenCV itself, exactly as the reference does it
// (src/VoxelCarving.cpp:25-36): pose = estimatePoseFromImage(...).inv(),
// cv::undistort on mask and image, intr = cameraMatrix as CV_32F; and
// M = intr * pose is taken from cv::gemm so that it is bit-identical to the
// reference's.  `estimatePoseFromImage` is the reference's own function
// (src/PoseEstimation.h:18-76, ChArUco, third party): the including
// translation unit must provide it, e.g. by including the reference's header.
#ifndef ARVX_OPENCV_DROPIN_HPP
#define ARVX_OPENCV_DROPIN_HPP

#if __has_include(<opencv2/core.hpp>) && __has_include(<opencv2/calib3d.hpp>)
#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>

#include <cmath>
#include <cstring>
#include <iostream>
#include <vector>

#include "arvx/voxel_carving.hpp"

cv::Mat estimatePoseFromImage(cv::Mat cameraMatrix, cv::Mat distCoeffs, cv::Mat image,
                              bool visualize);  // reference src/PoseEstimation.h:18

namespace arvx {
namespace dropin {

// what OpenCV said when asked (see the header comment)
struct PinReport {
    bool ran = false;
    int assoc = -1;               // ARVX_ASSOC_* cv::gemm showed on the known-answer voxel; -1: neither
    bool projection_ok = false;   // rows and quotients of probe voxels: kernels == cv::gemm, bit for bit
    bool norm_ok = false;         // (float)cv::norm(cameras[i] - world): kernels == OpenCV
    bool undistort_ok = false;    // arvx_undistort == cv::undistort on a distorted ramp
    int undistort_diff = 0;       // bytes that differ
};

inline PinReport run_self_pin(std::ostream &log) {
    PinReport rep;
    rep.ran = true;
    // (1) the known-answer voxel of tests/scenes.py::assoc_kat: voxel (1, 1, 1), s = 1, so
    // world = (1, 1, -1, 1); row 0 of M holds 1, 2^-24, 2^-54, 2^-53 -- the four products are
    // 1, 2^-24, -2^-54 * -1, 2^-53, and their fp64 sum rounds to 1 + 2^-23 in fp32 when grouped
    // p0 + ((p1 + p2) + p3) and to 1 when grouped ((p0 + p1) + p2) + p3.
    {
        const float third = (float)(1.0 / 3.5);
        float Mk[12] = {1.0f, std::ldexp(1.f, -24), -std::ldexp(1.f, -54), std::ldexp(1.f, -53),
                        0, 0, 0, third, 0, 0, 0, third};
        float wk[4] = {1.f, 1.f, -1.f, 1.f};
        cv::Mat M(3, 4, CV_32F, Mk), w(4, 1, CV_32F, wk);
        cv::Mat proj = M * w;  // the call of worldToCamera, src/VoxelCarving.cpp:19
        const float a0 = proj.at<float>(0, 0);
        if (a0 == 1.0f) rep.assoc = ARVX_ASSOC_LEFT;
        else if (a0 == 1.0f + std::ldexp(1.f, -23)) rep.assoc = ARVX_ASSOC_RIGHT;
        if (rep.assoc >= 0) {
            detail::check(arvx_set_projection_assoc(rep.assoc), "arvx_set_projection_assoc");
            log << "LOG - PIN: cv::gemm sums the rows of M * world as "
                << (rep.assoc == ARVX_ASSOC_LEFT ? "((p0 + p1) + p2) + p3 (ARVX_ASSOC_LEFT)"
                                                 : "p0 + ((p1 + p2) + p3) (ARVX_ASSOC_RIGHT)")
                << "; the kernels follow." << std::endl;
        } else {
            log << "LOG(WARN) - PIN: cv::gemm gave " << a0 << " on the known-answer voxel: neither "
                << "grouping the library knows; keeping its default ("
                << arvx_projection_assoc() << ")." << std::endl;
        }
    }
    // (2) probe voxels through cv::gemm / cv::norm and through the kernels
    arvx_ctx *ctx = nullptr;
    detail::check(arvx_ctx_create(&ctx, 0, 8, 8, 8, 0.001f), "arvx_ctx_create");
    struct Guard {
        arvx_ctx *c;
        ~Guard() { arvx_ctx_destroy(c); }
    } guard{ctx};
    {
        // a camera of the data set's kind: K(496.5, 496.8, 312.2, 250.9) * [R | t], rounded to fp32
        float Mp[12] = {420.37548828125f, -331.6723937988281f, -152.04873657226562f, 310.4010009765625f,
                        -67.51531219482422f, 98.3125991821289f, -538.1117553710938f, 355.2073974609375f,
                        0.4330126941204071f, 0.7071067690849304f, -0.5590170025825500f, 1.0240000486373901f};
        const float s = 0.001f;
        const int n = 64;
        std::vector<int32_t> xyz(3 * n);
        unsigned lcg = 12345u;
        for (int &v : xyz) {
            lcg = lcg * 1664525u + 1013904223u;
            v = (int)((lcg >> 8) % 1024u);
        }
        std::vector<float> dev(5 * n), ddepth(n);
        detail::check(arvx_selftest_project(ctx, n, Mp, s, xyz.data(), dev.data()),
                      "arvx_selftest_project");
        float campos[3] = {Mp[3] * 0.001f, Mp[7] * 0.001f, Mp[11]};
        detail::check(arvx_selftest_depth(ctx, n, campos, s, xyz.data(), ddepth.data()),
                      "arvx_selftest_depth");
        cv::Mat M(3, 4, CV_32F, Mp);
        int bad_rows = 0, bad_depth = 0;
        for (int i = 0; i < n; ++i) {
            const int x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
            float wv[4] = {y * s, x * s, -1 * z * s, 1.f};  // Model::toWord, src/Model.h:134-140
            cv::Mat w(4, 1, CV_32F, wv);
            cv::Mat proj = M * w;
            const float a0 = proj.at<float>(0, 0), a1 = proj.at<float>(1, 0), a2 = proj.at<float>(2, 0);
            const float got[5] = {dev[3 * i], dev[3 * i + 1], dev[3 * i + 2], dev[3 * n + 2 * i],
                                  dev[3 * n + 2 * i + 1]};
            const float want[5] = {a0, a1, a2, a0 / a2, a1 / a2};
            for (int k = 0; k < 5; ++k)
                if (std::memcmp(&got[k], &want[k], sizeof(float)) != 0) ++bad_rows;
            const cv::Vec4f cam(campos[0], campos[1], campos[2], 1.f), world(wv[0], wv[1], wv[2], wv[3]);
            const float d = (float)cv::norm(cam - world);  // src/ColorReconstruction.h:59
            if (std::memcmp(&d, &ddepth[i], sizeof(float)) != 0) ++bad_depth;
        }
        rep.projection_ok = bad_rows == 0;
        rep.norm_ok = bad_depth == 0;
        if (bad_rows)
            log << "LOG(WARN) - PIN: " << bad_rows << " of " << 5 * n
                << " projected values differ between cv::gemm and the kernels." << std::endl;
        else
            log << "LOG - PIN: rows and quotients of " << n << " probe voxels: kernels == cv::gemm."
                << std::endl;
        if (bad_depth)
            log << "LOG(WARN) - PIN: " << bad_depth << " of " << n
                << " depths differ between cv::norm and the kernels." << std::endl;
        else
            log << "LOG - PIN: depths of " << n << " probe voxels: kernels == cv::norm." << std::endl;
    }
    // (3) cv::undistort of a ramp against the device's remap
    {
        const int W = 96, H = 64;
        std::vector<uint8_t> src((size_t)W * H), dev((size_t)W * H);
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) src[(size_t)y * W + x] = (uint8_t)((x * 5 + y * 3 + (x * y) / 7) & 255);
        double Kd[9] = {74.5, 0, 46.8, 0, 74.6, 33.4, 0, 0, 1};
        double dist[5] = {0.12, -0.27, 0.0015, -0.0021, 0.11};
        cv::Mat K(3, 3, CV_64F, Kd), D(1, 5, CV_64F, dist), in(H, W, CV_8UC1, src.data()), out;
        cv::undistort(in, out, K, D);
        const uint8_t *sp = src.data();
        uint8_t *dp = dev.data();
        detail::check(arvx_undistort(ctx, 1, &sp, W, H, 1, (size_t)W, Kd, dist, 5, &dp), "arvx_undistort");
        int diff = 0;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                if (out.data[(size_t)y * (size_t)out.step + x] != dev[(size_t)y * W + x]) ++diff;
        rep.undistort_diff = diff;
        rep.undistort_ok = diff == 0;
        if (diff)
            log << "LOG(WARN) - PIN: arvx_undistort differs from cv::undistort in " << diff << " of "
                << W * H << " pixels; keep undistorting with OpenCV (this header does)." << std::endl;
        else
            log << "LOG - PIN: arvx_undistort == cv::undistort on a distorted " << W << "x" << H
                << " ramp." << std::endl;
    }
    return rep;
}

// runs once per process, before the first carve / colour pass of this header
inline const PinReport &self_pin() {
    static const PinReport rep = run_self_pin(std::cerr);
    return rep;
}

struct Prepared {
    Intrinsics intr;
    std::vector<cv::Mat> undist_imgs, undist_masks;  // keep the pixels alive
    std::vector<View> views;
};

inline Image as_image(const cv::Mat &m) {
    Image im;
    im.data = m.data;
    im.width = m.cols;
    im.height = m.rows;
    im.channels = m.channels();
    im.stride = m.step;
    return im;
}

inline void prepare(cv::Mat &cameraMatrix, cv::Mat &distCoeffs, std::vector<cv::Mat> &images,
                    std::vector<cv::Mat> &masks, bool with_images, Prepared &out) {
    (void)self_pin();  // first use: ask OpenCV how it computes (header comment)
    cv::Mat intr = cameraMatrix.clone();
    intr.convertTo(intr, CV_32F);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out.intr.K[3 * r + c] = intr.at<float>(r, c);
    const size_t V = images.size();
    out.undist_imgs.resize(V);
    out.undist_masks.resize(V);
    out.views.resize(V);
    for (size_t i = 0; i < V; ++i) {
        cv::Mat pose = estimatePoseFromImage(cameraMatrix, distCoeffs, images[i], false);
        pose = pose.inv();
        cv::Mat top = pose(cv::Rect(0, 0, 4, 3));
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) out.views[i].pose[4 * r + c] = top.at<float>(r, c);
        cv::Mat M = intr * top;  // cv::gemm, as in worldToCamera (src/VoxelCarving.cpp:19)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 4; ++c) out.views[i].M[4 * r + c] = M.at<float>(r, c);
        out.views[i].has_M = true;
        cv::undistort(masks[i], out.undist_masks[i], cameraMatrix, distCoeffs);
        out.views[i].mask = as_image(out.undist_masks[i]);
        if (with_images) {
            cv::undistort(images[i], out.undist_imgs[i], cameraMatrix, distCoeffs);
            out.views[i].image = as_image(out.undist_imgs[i]);
        }
    }
}

}  // namespace dropin
}  // namespace arvx

// ---- the reference's signatures, at global scope like the reference ----
using Model = arvx::Model;

namespace arvx {
namespace dropin {
// arvx_set_projection_assoc (self_pin) only reaches contexts created afterwards: a model whose
// context exists already -- uploaded, closed or meshed before its first carve here -- gets what
// cv::gemm showed as well, so that one process never carves with two groupings.
inline void apply_pin(Model &model) {
    const PinReport &rep = self_pin();
    if (rep.assoc >= 0 && model.device_if_any())
        detail::check(arvx_ctx_set_projection_assoc(model.device_if_any(), rep.assoc),
                      "arvx_ctx_set_projection_assoc");
}
}  // namespace dropin
}  // namespace arvx

inline void carve(cv::Mat &cameraMatrix, cv::Mat &distCoeffs, Model &model,
                  std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks,
                  bool intermediateMeshes = false) {
    // (pose estimation and cv::undistort are inside the reference's bracket too: src/VoxelCarving.cpp:62)
    arvx::detail::StageBracket timed(arvx::kStageCarving);
    arvx::dropin::Prepared p;
    arvx::dropin::prepare(cameraMatrix, distCoeffs, images, masks, false, p);
    arvx::dropin::apply_pin(model);
    arvx::carve(p.intr, model, p.views, intermediateMeshes);
}

inline void fastCarve(cv::Mat &cameraMatrix, cv::Mat &distCoeffs, Model &model,
                      std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks) {
    arvx::detail::StageBracket timed(arvx::kStageCarving);
    arvx::dropin::Prepared p;
    arvx::dropin::prepare(cameraMatrix, distCoeffs, images, masks, false, p);
    arvx::dropin::apply_pin(model);
    arvx::fastCarve(p.intr, model, p.views);
}

inline void reconstructClosestColor(cv::Mat &cameraMatrix, cv::Mat &distCoeffs, Model &model,
                                    std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks) {
    arvx::detail::StageBracket timed(arvx::kStageColoring);
    arvx::dropin::Prepared p;
    arvx::dropin::prepare(cameraMatrix, distCoeffs, images, masks, true, p);
    arvx::dropin::apply_pin(model);
    arvx::reconstructClosestColor(p.intr, model, p.views);
}

inline void reconstructAvgColor(cv::Mat &cameraMatrix, cv::Mat &distCoeffs, Model &model,
                                std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks) {
    arvx::detail::StageBracket timed(arvx::kStageColoring);
    arvx::dropin::Prepared p;
    arvx::dropin::prepare(cameraMatrix, distCoeffs, images, masks, true, p);
    arvx::dropin::apply_pin(model);
    arvx::reconstructAvgColor(p.intr, model, p.views);
}

#else
#error "arvx/opencv_dropin.hpp needs OpenCV (core, calib3d); use arvx/voxel_carving.hpp without it"
#endif
#endif
