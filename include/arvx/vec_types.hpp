// vec_types.hpp -- the small vector types of the reference's interface.
//
// The reference's Model / MarchingCubes / Benchmark signatures speak Eigen::Vector4f,
// Eigen::Vector3f (src/Model.h:5,68; src/MarchingCubes.h:6,12; src/Benchmark.h:3,71) and
// cv::Vec3i (src/Model.h:115,138,154,158).  Where those libraries are installed the C++ layer
// uses THEIR types (arvx::Vec4f IS Eigen::Vector4f, arvx::Vec3i IS cv::Vec3i), so the
// reference's callers compile unchanged; where they are not (this image has neither) the
// stand-ins below offer the members the reference's callers use: the (a, b, c[, d])
// constructor, (i), [i], x() y() z() w(), ==, !=, data().  -DARVX_NO_EIGEN / -DARVX_NO_OPENCV
// force the stand-ins.
//
// Code in this layer only relies on what both forms provide; in particular a
// default-constructed Eigen vector is NOT zeroed, so nothing here reads one.
#ifndef ARVX_VEC_TYPES_HPP
#define ARVX_VEC_TYPES_HPP

#if !defined(ARVX_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Dense>)
#define ARVX_HAVE_EIGEN 1
#endif
#endif
#if !defined(ARVX_NO_OPENCV) && defined(__has_include)
#if __has_include(<opencv2/core.hpp>)
#define ARVX_HAVE_OPENCV 1
#endif
#endif

#ifdef ARVX_HAVE_EIGEN
#include <Eigen/Dense>
#endif
#ifdef ARVX_HAVE_OPENCV
#include <opencv2/core.hpp>
#endif

namespace arvx {

#ifdef ARVX_HAVE_EIGEN
using Vec4f = Eigen::Vector4f;
using Vec3f = Eigen::Vector3f;
#else
struct Vec4f {
    float v[4];
    Vec4f() : v{0, 0, 0, 0} {}
    Vec4f(float a, float b, float c, float d) : v{a, b, c, d} {}
    float &operator()(int i) { return v[i]; }
    float operator()(int i) const { return v[i]; }
    float &operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
    float w() const { return v[3]; }
    float *data() { return v; }
    const float *data() const { return v; }
    bool operator==(const Vec4f &o) const {
        return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3];
    }
    bool operator!=(const Vec4f &o) const { return !(*this == o); }
};

struct Vec3f {
    float v[3];
    Vec3f() : v{0, 0, 0} {}
    Vec3f(float a, float b, float c) : v{a, b, c} {}
    float &operator()(int i) { return v[i]; }
    float operator()(int i) const { return v[i]; }
    float &operator[](int i) { return v[i]; }
    float operator[](int i) const { return v[i]; }
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
    float *data() { return v; }
    const float *data() const { return v; }
    bool operator==(const Vec3f &o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2]; }
    bool operator!=(const Vec3f &o) const { return !(*this == o); }
};
#endif

#ifdef ARVX_HAVE_OPENCV
using Vec3i = cv::Vec3i;
#else
struct Vec3i {
    int v[3];
    Vec3i() : v{0, 0, 0} {}
    Vec3i(int a, int b, int c) : v{a, b, c} {}
    int &operator()(int i) { return v[i]; }
    int operator()(int i) const { return v[i]; }
    int &operator[](int i) { return v[i]; }
    int operator[](int i) const { return v[i]; }
};
#endif

// the device hands colours and vertices over as packed floats, straight into arrays of these
static_assert(sizeof(Vec4f) == 4 * sizeof(float), "Vec4f is four packed floats");
static_assert(sizeof(Vec3f) == 3 * sizeof(float), "Vec3f is three packed floats");

}  // namespace arvx
#endif
