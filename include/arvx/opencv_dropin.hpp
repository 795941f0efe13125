// opencv_dropin.hpp -- the reference's exact cv::Mat signatures, for a build of
// the reference's main.cpp against this library.
//
//   void carve(cv::Mat&, cv::Mat&, Model&, std::vector<cv::Mat>&, std::vector<cv::Mat>&, bool)
//   void fastCarve(cv::Mat&, cv::Mat&, Model&, std::vector<cv::Mat>&, std::vector<cv::Mat>&)
//   void reconstructClosestColor(...same five...) / reconstructAvgColor(...)
//   (reference src/VoxelCarving.h:19,31; src/ColorReconstruction.h:131,142)
//
// Only compiled where OpenCV (core, calib3d) is installed; this image has none: the repo's
// tests only TYPE-CHECK this header against mock declarations of the cv::Mat members it uses
// (tests/cpp/mock_opencv, tests/test_cpp_host.py); it has never run.  The per-view
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

#include "arvx/voxel_carving.hpp"

cv::Mat estimatePoseFromImage(cv::Mat cameraMatrix, cv::Mat distCoeffs, cv::Mat image,
                              bool visualize);  // reference src/PoseEstimation.h:18

namespace arvx {
namespace dropin {

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

inline void carve(cv::Mat &cameraMatrix, cv::Mat &distCoeffs, Model &model,
                  std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks,
                  bool intermediateMeshes = false) {
    arvx::dropin::Prepared p;
    arvx::dropin::prepare(cameraMatrix, distCoeffs, images, masks, false, p);
    arvx::carve(p.intr, model, p.views, intermediateMeshes);
}

inline void fastCarve(cv::Mat &cameraMatrix, cv::Mat &distCoeffs, Model &model,
                      std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks) {
    arvx::dropin::Prepared p;
    arvx::dropin::prepare(cameraMatrix, distCoeffs, images, masks, false, p);
    arvx::fastCarve(p.intr, model, p.views);
}

inline void reconstructClosestColor(cv::Mat &cameraMatrix, cv::Mat &distCoeffs, Model &model,
                                    std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks) {
    arvx::dropin::Prepared p;
    arvx::dropin::prepare(cameraMatrix, distCoeffs, images, masks, true, p);
    arvx::reconstructClosestColor(p.intr, model, p.views);
}

inline void reconstructAvgColor(cv::Mat &cameraMatrix, cv::Mat &distCoeffs, Model &model,
                                std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks) {
    arvx::dropin::Prepared p;
    arvx::dropin::prepare(cameraMatrix, distCoeffs, images, masks, true, p);
    arvx::reconstructAvgColor(p.intr, model, p.views);
}

#else
#error "arvx/opencv_dropin.hpp needs OpenCV (core, calib3d); use arvx/voxel_carving.hpp without it"
#endif
#endif
