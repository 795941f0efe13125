// NOT OpenCV: see core.hpp in this directory.  Declarations only.
#ifndef ARVX_TESTS_MOCK_OPENCV_IMGPROC_HPP
#define ARVX_TESTS_MOCK_OPENCV_IMGPROC_HPP
#include "opencv2/core.hpp"
namespace cv {
enum ColorConversionCodes { COLOR_BGR2RGB = 4, COLOR_BGR2HSV = 40 };
enum HersheyFonts { FONT_HERSHEY_SIMPLEX = 0 };
void cvtColor(InputArray src, OutputArray dst, int code, int dstCn = 0);
void resize(InputArray src, OutputArray dst, Size dsize, double fx = 0, double fy = 0, int interpolation = 1);
void putText(InputOutputArray img, const String &text, Point org, int fontFace, double fontScale, Scalar color,
             int thickness = 1, int lineType = 8, bool bottomLeftOrigin = false);
}  // namespace cv
#endif
