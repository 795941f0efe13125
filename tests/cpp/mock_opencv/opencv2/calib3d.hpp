// NOT OpenCV: see core.hpp in this directory.
#ifndef ARVX_TESTS_MOCK_OPENCV_CALIB3D_HPP
#define ARVX_TESTS_MOCK_OPENCV_CALIB3D_HPP
#include "opencv2/core.hpp"
namespace cv {
void undistort(const Mat &src, Mat &dst, const Mat &cameraMatrix, const Mat &distCoeffs);
}
#endif
