// NOT OpenCV: definitions behind tests/cpp/mock_opencv/opencv2/*.hpp for tests/cpp/test_selfpin.cpp.
// The arithmetic here is whatever the TEST wants OpenCV to look like:
//   cv::mock::gemm_assoc      how operator* sums the four products of a row (0: p0+((p1+p2)+p3),
//                             1: ((p0+p1)+p2)+p3, 2: a result scaled by 1 + 2^-20: neither on the
//                             known-answer voxel, 3: plain fp32 accumulation: looks like 1 on the
//                             known-answer voxel, differs on ordinary ones)
//   cv::mock::undistort_mode  0: the library's own remap (arvx_undistort: "OpenCV agrees"),
//                             1: a copy of the input ("OpenCV disagrees")
//   cv::mock::norm_fp32       cv::norm accumulates in fp32 instead of fp64 ("disagrees")
#ifndef ARVX_TESTS_MOCK_OPENCV_IMPL_HPP
#define ARVX_TESTS_MOCK_OPENCV_IMPL_HPP
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "arvx/arvx.h"
#include "opencv2/calib3d.hpp"
#include "opencv2/core.hpp"

namespace cv {
namespace mock {
inline int gemm_assoc = 1;
inline int undistort_mode = 0;
inline bool norm_fp32 = false;
inline size_t elem_size(int type) { return type == CV_8U ? 1 : (type == CV_32F ? 4 : 8); }
}  // namespace mock

inline Mat::Mat(int rows_, int cols_, int type, void *data_)
    : data((unsigned char *)data_), rows(rows_), cols(cols_), type_(type) {
    step.v = (size_t)cols_ * mock::elem_size(type);
}
inline Mat::Mat(int rows_, int cols_, int type) : rows(rows_), cols(cols_), type_(type) {
    step.v = (size_t)cols_ * mock::elem_size(type);
    own = std::make_shared<std::vector<unsigned char>>((size_t)rows_ * step.v);
    data = own->data();
}
inline int Mat::channels() const { return 1; }

// 3x4 * 4x1 only: fp32 inputs, fp64 products, the grouping the test selected
inline Mat operator*(const Mat &a, const Mat &b) {
    if (a.type_ != CV_32F || b.type_ != CV_32F || a.cols != 4 || b.rows != 4 || b.cols != 1)
        throw std::logic_error("mock cv::gemm: 3x4 * 4x1 CV_32F only");
    Mat d(a.rows, 1, CV_32F);
    Mat &am = const_cast<Mat &>(a), &bm = const_cast<Mat &>(b);
    for (int r = 0; r < a.rows; ++r) {
        double p[4];
        for (int k = 0; k < 4; ++k) p[k] = (double)am.at<float>(r, k) * (double)bm.at<float>(k, 0);
        float out;
        if (mock::gemm_assoc == 0) out = (float)(p[0] + ((p[1] + p[2]) + p[3]));
        else if (mock::gemm_assoc == 1) out = (float)(((p[0] + p[1]) + p[2]) + p[3]);
        else if (mock::gemm_assoc == 2) out = (float)((((p[0] + p[1]) + p[2]) + p[3]) * (1.0 + std::ldexp(1.0, -20)));
        else {
            volatile float acc = 0.f;
            for (int k = 0; k < 4; ++k) acc = acc + (float)p[k];
            out = acc;
        }
        d.at<float>(r, 0) = out;
    }
    return d;
}

inline Vec4f operator-(const Vec4f &a, const Vec4f &b) {
    return Vec4f(a[0] - b[0], a[1] - b[1], a[2] - b[2], a[3] - b[3]);
}
inline double norm(const Vec4f &v) {
    if (mock::norm_fp32) {
        volatile float s = 0.f;
        for (int k = 0; k < 4; ++k) s = s + v[k] * v[k];
        return std::sqrt((double)s);
    }
    double s = 0;
    const double v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
    s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
    return std::sqrt(s);
}

inline void undistort(const Mat &src, Mat &dst, const Mat &cameraMatrix, const Mat &distCoeffs) {
    dst = Mat(src.rows, src.cols, CV_8U);
    if (mock::undistort_mode == 1) {
        for (int y = 0; y < src.rows; ++y)
            std::memcpy(dst.data + (size_t)y * dst.step, src.data + (size_t)y * src.step, (size_t)src.cols);
        return;
    }
    arvx_ctx *ctx = nullptr;
    if (arvx_ctx_create(&ctx, 0, 8, 8, 8, 0.001f) != ARVX_OK) throw std::runtime_error("no device");
    const uint8_t *sp = src.data;
    uint8_t *dp = dst.data;
    const int rc = arvx_undistort(ctx, 1, &sp, src.cols, src.rows, 1, (size_t)src.step,
                                  (const double *)cameraMatrix.data, (const double *)distCoeffs.data,
                                  distCoeffs.cols, &dp);
    arvx_ctx_destroy(ctx);
    if (rc != ARVX_OK) throw std::runtime_error(arvx_last_error());
}

}  // namespace cv
#endif
