// calibration.hpp -- reader for the camera calibration file the reference loads with
// loadCalibrationFile() (src/PoseEstimation.h:11-16 -> readCameraParameters,
// src/aruco_samples_utility.hpp:9-16): an OpenCV FileStorage YAML with the keys
// `camera_matrix` (3x3, dt: d) and `distortion_coefficients` (1xN, dt: d), e.g.
// Data/box_dataset/cameracalibration.yml.  No OpenCV needed: the two `data: [...]`
// lists are parsed directly.
#ifndef ARVX_CALIBRATION_HPP
#define ARVX_CALIBRATION_HPP

#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace arvx {

namespace detail {
inline bool yaml_matrix(const std::string &text, const std::string &key, int &rows, int &cols,
                        std::vector<double> &data) {
    const size_t k = text.find(key + ":");
    if (k == std::string::npos) return false;
    auto field = [&](const char *name, size_t from) -> std::string {
        const size_t p = text.find(name, from);
        if (p == std::string::npos) return "";
        const size_t e = text.find('\n', p);
        return text.substr(p + std::char_traits<char>::length(name), e - p);
    };
    rows = std::atoi(field("rows:", k).c_str());
    cols = std::atoi(field("cols:", k).c_str());
    const size_t d = text.find("data:", k);
    if (d == std::string::npos) return false;
    const size_t lb = text.find('[', d), rb = text.find(']', d);
    if (lb == std::string::npos || rb == std::string::npos || rb < lb) return false;
    std::string body = text.substr(lb + 1, rb - lb - 1);
    for (char &c : body)
        if (c == ',' || c == '\n' || c == '\r') c = ' ';
    std::istringstream ss(body);
    data.clear();
    std::string tok;
    while (ss >> tok) data.push_back(std::strtod(tok.c_str(), nullptr));
    return rows > 0 && cols > 0 && (int)data.size() == rows * cols;
}
}  // namespace detail

// cameraMatrix (row-major 3x3, double as stored) and distCoeffs; false if the file or a
// key is missing -- the reference prints "Invalid camera file" in that case.
inline bool readCameraParameters(const std::string &path, double K[9], std::vector<double> &dist) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::stringstream buf;
    buf << f.rdbuf();
    const std::string text = buf.str();
    int r = 0, c = 0;
    std::vector<double> k;
    if (!detail::yaml_matrix(text, "camera_matrix", r, c, k) || r != 3 || c != 3) return false;
    for (int i = 0; i < 9; ++i) K[i] = k[i];
    if (!detail::yaml_matrix(text, "distortion_coefficients", r, c, dist)) return false;
    return true;
}

}  // namespace arvx
#endif
