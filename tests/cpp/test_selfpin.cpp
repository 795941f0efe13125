// Runs the self-pinning of include/arvx/opencv_dropin.hpp (dropin::run_self_pin) against the
// stand-ins of tests/cpp/mock_opencv: a "cv::gemm" for each row-sum grouping, an unknown one, a
// cv::norm and a cv::undistort that agree or disagree with the library.  What is checked is
// the LOGIC of the self-pinning -- that it reads OpenCV's answer correctly, switches the
// library, and notices disagreement -- not OpenCV's arithmetic (there is no OpenCV here).
// Needs a GPU (arvx_selftest_project / _depth / arvx_undistort).  Prints "selfpin ok".
#include <cstdio>
#include <iostream>
#include <sstream>

#include "mock_impl.hpp"
//
#include "arvx/opencv_dropin.hpp"

cv::Mat estimatePoseFromImage(cv::Mat, cv::Mat, cv::Mat, bool) { return cv::Mat(); }

#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::fprintf(stderr, "FAILED %s:%d: %s\n%s", __FILE__, __LINE__, #cond, log.str().c_str()); \
            return 1;                                                       \
        }                                                                   \
    } while (0)

int main() {
    using arvx::dropin::PinReport;
    for (int assoc = 0; assoc < 2; ++assoc) {  // OpenCV sums one way or the other
        std::ostringstream log;
        cv::mock::gemm_assoc = assoc;
        cv::mock::undistort_mode = 0;
        cv::mock::norm_fp32 = false;
        arvx_set_projection_assoc(1 - assoc);  // the library starts with the OTHER one
        const PinReport r = arvx::dropin::run_self_pin(log);
        EXPECT(r.ran && r.assoc == assoc);
        EXPECT(arvx_projection_assoc() == assoc);  // switched
        EXPECT(r.projection_ok && r.norm_ok && r.undistort_ok && r.undistort_diff == 0);
        EXPECT(log.str().find("LOG(WARN)") == std::string::npos);
        arvx_ctx *c = nullptr;  // new contexts follow
        EXPECT(arvx_ctx_create(&c, 0, 4, 4, 4, 1.f) == ARVX_OK);
        int got = -1;
        EXPECT(arvx_ctx_projection_assoc(c, &got) == ARVX_OK && got == assoc);
        arvx_ctx_destroy(c);
    }
    {  // an OpenCV whose gemm is neither: warn, keep the default, report the mismatch
        std::ostringstream log;
        cv::mock::gemm_assoc = 2;
        arvx_set_projection_assoc(ARVX_ASSOC_LEFT);
        const PinReport r = arvx::dropin::run_self_pin(log);
        EXPECT(r.assoc == -1 && arvx_projection_assoc() == ARVX_ASSOC_LEFT);
        EXPECT(!r.projection_ok);
        EXPECT(log.str().find("LOG(WARN) - PIN: cv::gemm gave") != std::string::npos);
    }
    {  // one that answers the known-answer voxel like LEFT but is not LEFT: the probes notice
        std::ostringstream log;
        cv::mock::gemm_assoc = 3;
        const PinReport r = arvx::dropin::run_self_pin(log);
        EXPECT(r.assoc == ARVX_ASSOC_LEFT && !r.projection_ok);
        EXPECT(log.str().find("projected values differ") != std::string::npos);
    }
    {  // cv::norm and cv::undistort that disagree are noticed
        std::ostringstream log;
        cv::mock::gemm_assoc = 1;
        cv::mock::norm_fp32 = true;
        cv::mock::undistort_mode = 1;
        const PinReport r = arvx::dropin::run_self_pin(log);
        EXPECT(r.assoc == 1 && r.projection_ok);
        EXPECT(!r.norm_ok && !r.undistort_ok && r.undistort_diff > 0);
        EXPECT(log.str().find("depths differ") != std::string::npos);
        EXPECT(log.str().find("arvx_undistort differs") != std::string::npos);
    }
    std::printf("selfpin ok\n");
    return 0;
}
