// NOT OpenCV: see core.hpp in this directory.
#ifndef ARVX_TESTS_MOCK_OPENCV_CALIB3D_HPP
#define ARVX_TESTS_MOCK_OPENCV_CALIB3D_HPP
#include "opencv2/core.hpp"
namespace cv {
void undistort(const Mat &src, Mat &dst, const Mat &cameraMatrix, const Mat &distCoeffs);
void Rodrigues(InputArray src, OutputArray dst, OutputArray jacobian = noArray());
void drawFrameAxes(InputOutputArray image, InputArray cameraMatrix, InputArray distCoeffs, InputArray rvec,
                   InputArray tvec, float length, int thickness = 3);
enum {
    CALIB_USE_INTRINSIC_GUESS = 0x00001,
    CALIB_FIX_ASPECT_RATIO = 0x00002,
    CALIB_FIX_PRINCIPAL_POINT = 0x00004,
    CALIB_ZERO_TANGENT_DIST = 0x00008
};
}  // namespace cv
#endif
