// NOT OpenCV: see core.hpp in this directory.  Declarations only (highgui, imgcodecs, videoio).
#ifndef ARVX_TESTS_MOCK_OPENCV_HIGHGUI_HPP
#define ARVX_TESTS_MOCK_OPENCV_HIGHGUI_HPP
#include "opencv2/core.hpp"
namespace cv {
void imshow(const String &winname, InputArray mat);
int waitKey(int delay = 0);
void destroyAllWindows();
Mat imread(const String &filename, int flags = 1);
bool imwrite(const String &filename, InputArray img);
class VideoCapture {
   public:
    VideoCapture();
    bool open(int index, int apiPreference = 0);
    bool grab();
    bool retrieve(Mat &image, int flag = 0);
};
}  // namespace cv
#endif
