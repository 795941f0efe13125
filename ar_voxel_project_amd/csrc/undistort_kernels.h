// undistort_kernels.h -- cv::undistort(src, dst, cameraMatrix, distCoeffs) on the device: the
// per-view pre-processing step the reference runs on every mask and image before its voxel
// loops (src/VoxelCarving.cpp:35-36,86-90; src/ColorReconstruction.h:22-27).
//
// Restated from the published OpenCV 4.x sources (modules/calib3d/src/undistort.dispatch.cpp,
// undistort.simd.hpp; modules/imgproc/src/imgwarp.cpp), PARITY UNPINNED -- no OpenCV exists in
// this image to compare against:
//   * cv::undistort = initUndistortRectifyMap(A, dist, I, A, size, CV_16SC2) + remap(...,
//     INTER_LINEAR, BORDER_CONSTANT 0).
//   * the map, in double: x = (u - cx) / fx, y = (v - cy) / fy (as u * ir0 + ir2 with
//     ir = inv(A)); r2 = x^2 + y^2; kr = (1 + ((k3 r2 + k2) r2 + k1) r2) / (1 + ((k6 r2 + k5)
//     r2 + k4) r2); xd = x kr + p1 2xy + p2 (r2 + 2 x^2); yd = y kr + p1 (r2 + 2 y^2) + p2 2xy;
//     u' = fx xd + cx, v' = fy yd + cy.  (OpenCV accumulates x along a row by repeated
//     addition of ir0, and its AVX2 build evaluates several pixels at once: the last bit of
//     u' can differ from this closed form, which moves a pixel only at an exact rounding tie
//     of the next step.)
//   * fixed point: iu = cvRound(u' * 32) (round half to even), integer part iu >> 5 stored as
//     short, fraction iu & 31; same for v'.
//   * bilinear weights from OpenCV's table: (32 - a)(32 - b) * 32, a (32 - b) * 32,
//     (32 - a) b * 32, a b * 32 of 32768 (products of multiples of 1/32 are exact in float, so
//     the table's rounding and its sum correction never act); result
//     (sum of weight * tap + 16384) >> 15; taps outside the source read the border value 0.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace arvx {

struct UndistortParams {
    double fx, fy, cx, cy;  // cameraMatrix
    double ir0, ir2, ir4, ir5;  // inv(cameraMatrix): 1/fx, -cx/fx, 1/fy, -cy/fy
    double k1, k2, p1, p2, k3, k4, k5, k6;
    int W, H, C;
};

__device__ __forceinline__ int cv_round(double v) { return __double2int_rn(v); }

__global__ __launch_bounds__(256) void undistort_kernel(const uint8_t *__restrict__ src,
                                                        uint8_t *__restrict__ dst,
                                                        const UndistortParams p) {
    const int u = blockIdx.x * 64 + (threadIdx.x & 63);
    const int v = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (u >= p.W || v >= p.H) return;
    const size_t img = (size_t)p.W * p.H * p.C;
    const uint8_t *s = src + img * blockIdx.z;
    uint8_t *d = dst + img * blockIdx.z + ((size_t)v * p.W + u) * p.C;
    const double x = (double)u * p.ir0 + p.ir2, y = (double)v * p.ir4 + p.ir5;
    const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
    const double kr = (1 + ((p.k3 * r2 + p.k2) * r2 + p.k1) * r2) /
                      (1 + ((p.k6 * r2 + p.k5) * r2 + p.k4) * r2);
    const double xd = x * kr + p.p1 * _2xy + p.p2 * (r2 + 2 * x2);
    const double yd = y * kr + p.p1 * (r2 + 2 * y2) + p.p2 * _2xy;
    const double uu = p.fx * xd + p.cx, vv = p.fy * yd + p.cy;
    const int iu = cv_round(uu * 32.0), iv = cv_round(vv * 32.0);
    int sx = iu >> 5, sy = iv >> 5;
    sx = max(-32768, min(32767, sx));  // saturate_cast<short>
    sy = max(-32768, min(32767, sy));
    const int a = iu & 31, b = iv & 31;
    const int w00 = (32 - a) * (32 - b) * 32, w01 = a * (32 - b) * 32, w10 = (32 - a) * b * 32,
              w11 = a * b * 32;
    const bool x0 = sx >= 0 && sx < p.W, x1 = sx + 1 >= 0 && sx + 1 < p.W;
    const bool y0 = sy >= 0 && sy < p.H, y1 = sy + 1 >= 0 && sy + 1 < p.H;
    for (int c = 0; c < p.C; ++c) {
        const int t00 = (x0 && y0) ? s[((size_t)sy * p.W + sx) * p.C + c] : 0;
        const int t01 = (x1 && y0) ? s[((size_t)sy * p.W + sx + 1) * p.C + c] : 0;
        const int t10 = (x0 && y1) ? s[((size_t)(sy + 1) * p.W + sx) * p.C + c] : 0;
        const int t11 = (x1 && y1) ? s[((size_t)(sy + 1) * p.W + sx + 1) * p.C + c] : 0;
        const int acc = w00 * t00 + w01 * t01 + w10 * t10 + w11 * t11;
        d[c] = (uint8_t)((acc + (1 << 14)) >> 15);
    }
}

}  // namespace arvx
