// NOT OpenCV.  The few declarations of cv::Mat / cv::Vec4f that include/arvx/opencv_dropin.hpp
// uses, so that the header can be type-checked in an image without OpenCV
// (tests/test_cpp_host.py::test_opencv_dropin_header_type_checks), and so that its
// self-pinning can be RUN against stand-ins whose arithmetic the test chooses
// (tests/cpp/test_selfpin.cpp + mock_impl.hpp: a cv::gemm per row-sum grouping, a cv::undistort
// that agrees or disagrees with the library).  No result of the repo depends on this file.
#ifndef ARVX_TESTS_MOCK_OPENCV_CORE_HPP
#define ARVX_TESTS_MOCK_OPENCV_CORE_HPP
#include <cstddef>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 0

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
    int type_ = 0;
    std::shared_ptr<std::vector<unsigned char>> own;  // storage of matrices the mock allocates
    Mat() = default;
    Mat(int rows_, int cols_, int type, void *data_);  // header over caller memory
    Mat(int rows_, int cols_, int type);                // allocates
    Mat clone() const;
    void convertTo(Mat &dst, int rtype) const;
    Mat inv() const;
    Mat operator()(const Rect &roi) const;
    int channels() const;
    template <class T>
    T &at(int r, int c) {
        return *reinterpret_cast<T *>(data + (size_t)r * (size_t)step + (size_t)c * sizeof(T));
    }
};

Mat operator*(const Mat &a, const Mat &b);

struct Vec4f {
    float val[4];
    Vec4f(float a, float b, float c, float d) : val{a, b, c, d} {}
    float operator[](int i) const { return val[i]; }
};
Vec4f operator-(const Vec4f &a, const Vec4f &b);
double norm(const Vec4f &v);

}  // namespace cv
#endif
