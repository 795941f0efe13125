// NOT OpenCV.  The few declarations of cv::Mat / cv::Vec4f that include/arvx/opencv_dropin.hpp
// uses, so that the header can be type-checked in an image without OpenCV
// (tests/test_cpp_host.py::test_opencv_dropin_header_type_checks), and so that its
// self-pinning can be RUN against stand-ins whose arithmetic the test chooses
// (tests/cpp/test_selfpin.cpp + mock_impl.hpp: a cv::gemm per row-sum grouping, a cv::undistort
// that agrees or disagrees with the library).  tests/test_dropin_main.py additionally
// type-checks the reference's own main.cpp (with the headers it keeps: Calibration.h,
// PoseEstimation.h, Segmentation.h) against include/arvx/dropin/: for that the directory also
// DECLARES -- nothing more -- the rest of the OpenCV surface those files name (highgui,
// imgproc, aruco, FileStorage, CommandLineParser), with catch-all array proxies instead of
// OpenCV's InputArray machinery.  No result of the repo depends on this file.
#ifndef ARVX_TESTS_MOCK_OPENCV_CORE_HPP
#define ARVX_TESTS_MOCK_OPENCV_CORE_HPP
#include <algorithm>  // (the reference's Calibration.h gets std::find through OpenCV's headers)
#include <cstddef>
#include <memory>
#include <ostream>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 0
#define CV_8UC3 16

typedef unsigned char uchar;

namespace cv {

typedef std::string String;

template <class T>
struct Ptr : std::shared_ptr<T> {
    using std::shared_ptr<T>::shared_ptr;
    template <class Y>
    Ptr<Y> staticCast() const;
};

template <class T, int N>
struct Vec {
    T val[N];
    Vec() : val{} {}
    Vec(T a, T b, T c) : val{a, b, c} { static_assert(N == 3, "three elements"); }
    T &operator()(int i) { return val[i]; }
    const T &operator()(int i) const { return val[i]; }
    T &operator[](int i) { return val[i]; }
    const T &operator[](int i) const { return val[i]; }
};
typedef Vec<int, 3> Vec3i;
typedef Vec<double, 3> Vec3d;
typedef Vec<uchar, 3> Vec3b;

struct Point2f {
    float x = 0, y = 0;
};
struct Point {
    int x = 0, y = 0;
    Point() = default;
    Point(int x_, int y_) : x(x_), y(y_) {}
};
struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
};
struct Scalar {
    double val[4];
    Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {}
};
struct TermCriteria {
    enum Type { COUNT = 1, MAX_ITER = COUNT, EPS = 2 };
    TermCriteria(int type, int maxCount, double epsilon);
};

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
    Mat(Size size, int type);
    static Mat eye(int rows, int cols, int type);
    Mat t() const;
    Mat row(int y) const;
    size_t total() const;
    Size size() const;
    int type() const;
    bool empty() const;
    void copyTo(Mat &dst) const;
    void copyTo(Mat &dst, const Mat &mask) const;
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
Mat operator*(const Mat &a, const Vec3d &b);
Mat operator-(const Mat &a);
Mat operator~(const Mat &a);
std::ostream &operator<<(std::ostream &os, const Mat &m);

// catch-all proxies where OpenCV has InputArray / OutputArray / InputOutputArray
struct AnyArray {
    AnyArray() {}
    template <class T>
    AnyArray(const T &) {}
};
typedef const AnyArray &InputArray;
typedef const AnyArray &InputArrayOfArrays;
typedef const AnyArray &OutputArray;
typedef const AnyArray &OutputArrayOfArrays;
typedef const AnyArray &InputOutputArray;
typedef const AnyArray &InputOutputArrayOfArrays;
const AnyArray &noArray();

void glob(String pattern, std::vector<String> &result, bool recursive = false);
double kmeans(InputArray data, int K, InputOutputArray bestLabels, TermCriteria criteria, int attempts,
              int flags, OutputArray centers = noArray());
enum KmeansFlags { KMEANS_RANDOM_CENTERS = 0, KMEANS_PP_CENTERS = 2 };
void inRange(InputArray src, InputArray lowerb, InputArray upperb, OutputArray dst);

class FileNode {};
void operator>>(const FileNode &n, Mat &m);
class FileStorage {
   public:
    enum Mode { READ = 0, WRITE = 1 };
    FileStorage(const String &filename, int flags);
    bool isOpened() const;
    FileNode operator[](const char *nodename) const;
};
template <class T>
FileStorage &operator<<(FileStorage &fs, const T &value);

class CommandLineParser {
   public:
    CommandLineParser(int argc, const char *const argv[], const String &keys);
    void about(const String &message);
    void printMessage() const;
    bool has(const String &name) const;
    template <class T>
    T get(const String &name, bool space_delete = true) const;
};

struct Vec4f {
    float val[4];
    Vec4f(float a, float b, float c, float d) : val{a, b, c, d} {}
    float operator[](int i) const { return val[i]; }
};
Vec4f operator-(const Vec4f &a, const Vec4f &b);
double norm(const Vec4f &v);

}  // namespace cv
#endif
