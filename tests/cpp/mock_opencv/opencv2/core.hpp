// NOT OpenCV.  The few declarations of cv::Mat that include/arvx/opencv_dropin.hpp uses, so
// that the header can be type-checked in an image without OpenCV
// (tests/test_cpp_host.py::test_opencv_dropin_header_type_checks).  Declarations only: nothing
// here computes anything, and no result of the repo depends on it.
#ifndef ARVX_TESTS_MOCK_OPENCV_CORE_HPP
#define ARVX_TESTS_MOCK_OPENCV_CORE_HPP
#include <cstddef>

#define CV_32F 5

namespace cv {

struct Rect {
    int x, y, width, height;
    Rect(int x_, int y_, int w_, int h_) : x(x_), y(y_), width(w_), height(h_) {}
};

struct MatStep {
    size_t v = 0;
    operator size_t() const { return v; }
};

class Mat {
   public:
    unsigned char *data = nullptr;
    int rows = 0, cols = 0;
    MatStep step;
    Mat clone() const;
    void convertTo(Mat &dst, int rtype) const;
    Mat inv() const;
    Mat operator()(const Rect &roi) const;
    int channels() const;
    template <class T>
    T &at(int r, int c);
};

Mat operator*(const Mat &a, const Mat &b);

}  // namespace cv
#endif
