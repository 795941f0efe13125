// Type check of include/arvx/opencv_dropin.hpp against the mock declarations in
// tests/cpp/mock_opencv (compiled with -fsyntax-only; never linked, never run).
#include <vector>

#include "arvx/opencv_dropin.hpp"

void use(cv::Mat &K, cv::Mat &dist, std::vector<cv::Mat> &images, std::vector<cv::Mat> &masks) {
    Model model(10, 10, 5, 0.028f);
    carve(K, dist, model, images, masks);
    carve(K, dist, model, images, masks, true);
    fastCarve(K, dist, model, images, masks);
    reconstructClosestColor(K, dist, model, images, masks);
    reconstructAvgColor(K, dist, model, images, masks);
}
