// arvx_cli.cpp -- the reference's command line (src/main.cpp:16-39) for the parts of it that
// this library covers: `-c=5` (voxel carving: carve | fastCarve -> colour -> handleUnseen ->
// [closure] -> marching cubes -> OFF file, src/main.cpp:184-305) and `-c=6` (the eight-run
// benchmark table, :306-440), with the reference's flag names, defaults, checks, messages
// and stage order.
//
// What the reference does with OpenCV before its voxel loops -- JPEG decoding, ChArUco pose
// estimation, cv::undistort -- is not in this image, so the inputs are what those
// steps produce:
//   -images=<dir> -masks=<dir>   binary PPM (P6) / PGM (P5) files, already undistorted, taken
//                                in sorted file-name order like cv::glob
//   -poses=<file>                one world->camera matrix per view, 12 floats per line: the
//                                top three rows of estimatePoseFromImage(...).inv()
//                                (src/VoxelCarving.cpp:25-26)
//   -scene=<file>                or all of it in one binary file (tests/cpp/test_host.cpp)
//   -calibration=<yml>           the reference's calibration file; its camera matrix is used
//                                (src/PoseEstimation.h:11-16), the distortion coefficients only
//                                with -undistort=true: then the inputs are RAW images and
//                                arvx_undistort remaps them on the device (see arvx.h)
// -c=1..4 (board creation, camera calibration, live pose estimation, segmentation) are
// OpenCV/aruco programs around the path and are not provided.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "arvx/calibration.hpp"
#include "bench6.hpp"

namespace fs = std::filesystem;

namespace {

const char *about = "AR Voxel Carving Project\n";

// cv::CommandLineParser-like: -name=value or --name=value; a bare -name means true
struct Args {
    std::map<std::string, std::string> kv;
    Args(int argc, char **argv, const std::map<std::string, std::string> &defaults) : kv(defaults) {
        for (int i = 1; i < argc; ++i) {
            std::string a = argv[i];
            while (!a.empty() && a[0] == '-') a.erase(0, 1);
            const size_t eq = a.find('=');
            if (eq == std::string::npos) kv[a] = "true";
            else kv[a.substr(0, eq)] = a.substr(eq + 1);
        }
    }
    std::string str(const std::string &k) const {
        auto it = kv.find(k);
        return it == kv.end() ? "" : it->second;
    }
    int i(const std::string &k) const { return std::atoi(str(k).c_str()); }
    float f(const std::string &k) const { return (float)std::atof(str(k).c_str()); }
    bool b(const std::string &k) const {
        const std::string v = str(k);
        return v == "true" || v == "1" || v == "TRUE" || v == "True";
    }
};

struct Picture {
    int w = 0, h = 0, c = 0;
    std::vector<uint8_t> px;
};

// binary PGM (P5, 1 channel) / PPM (P6, RGB -> stored BGR like cv::imread)
bool read_pnm(const std::string &path, Picture &out, bool as_color) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    if (magic != "P5" && magic != "P6") return false;
    auto next_int = [&]() {
        int v = 0;
        for (;;) {
            int ch = f.peek();
            if (ch == '#') {
                std::string line;
                std::getline(f, line);
            } else if (isspace(ch)) {
                f.get();
            } else {
                break;
            }
        }
        f >> v;
        return v;
    };
    const int w = next_int(), h = next_int(), maxv = next_int();
    f.get();  // the single whitespace before the raster
    if (w < 1 || h < 1 || maxv != 255) return false;
    const int cin = magic == "P6" ? 3 : 1;
    std::vector<uint8_t> raw((size_t)w * h * cin);
    f.read((char *)raw.data(), (std::streamsize)raw.size());
    if (!f) return false;
    out.w = w;
    out.h = h;
    out.c = as_color ? 3 : cin;  // cv::imread(.., 1): always 3 channels, BGR
    out.px.resize((size_t)w * h * out.c);
    for (size_t p = 0; p < (size_t)w * h; ++p) {
        if (cin == 3 && out.c == 3) {
            out.px[3 * p] = raw[3 * p + 2];
            out.px[3 * p + 1] = raw[3 * p + 1];
            out.px[3 * p + 2] = raw[3 * p];
        } else if (cin == 1 && out.c == 3) {
            out.px[3 * p] = out.px[3 * p + 1] = out.px[3 * p + 2] = raw[p];
        } else {
            out.px[p] = raw[p];
        }
    }
    return true;
}

std::vector<std::string> glob_sorted(const std::string &dir) {  // cv::glob: sorted names
    std::vector<std::string> names;
    if (!fs::is_directory(dir)) return names;
    for (const auto &e : fs::directory_iterator(dir))
        if (e.is_regular_file()) names.push_back(e.path().string());
    std::sort(names.begin(), names.end());
    return names;
}

struct Inputs {
    arvx::Intrinsics intr{};
    std::vector<double> dist;
    std::vector<arvx::View> views;
    std::vector<Picture> images, masks;
    std::vector<uint8_t> scene_masks, scene_images;
};

// -scene=<file>: int32 X,Y,Z,V,W,H,C ; float s ; float K[9] ; V x pose[12] ; masks ; images
bool load_scene(const std::string &path, Inputs &in) {
    std::ifstream f(path, std::ios::binary);
    int32_t hd[7];
    f.read((char *)hd, sizeof hd);
    const int V = hd[3], W = hd[4], H = hd[5], C = hd[6];
    float s_unused;
    f.read((char *)&s_unused, 4);
    f.read((char *)in.intr.K, 36);
    if (!f || V < 1) return false;
    in.views.resize(V);
    for (auto &v : in.views) f.read((char *)v.pose, 48);
    in.scene_masks.resize((size_t)V * H * W * C);
    in.scene_images.resize((size_t)V * H * W * 3);
    f.read((char *)in.scene_masks.data(), (std::streamsize)in.scene_masks.size());
    f.read((char *)in.scene_images.data(), (std::streamsize)in.scene_images.size());
    if (!f) return false;
    for (int i = 0; i < V; ++i) {
        in.views[i].mask = {in.scene_masks.data() + (size_t)i * H * W * C, W, H, C, (size_t)W * C};
        in.views[i].image = {in.scene_images.data() + (size_t)i * H * W * 3, W, H, 3, (size_t)W * 3};
    }
    return true;
}

bool load_poses(const std::string &path, std::vector<arvx::View> &views) {
    std::ifstream f(path);
    if (!f) return false;
    for (auto &v : views)
        for (int k = 0; k < 12; ++k)
            if (!(f >> v.pose[k])) return false;
    return true;
}

}  // namespace

int main(int argc, char *argv[]) {
    // src/main.cpp:45-47
    fs::create_directories("./out");
    fs::create_directories("./out/tmp");
    fs::create_directories("./out/intermediate");
    const std::map<std::string, std::string> defaults = {
        {"calibration", "out/cameracalibration.yml"}, {"carve", "1"}, {"x", "100"}, {"y", "100"},
        {"z", "100"}, {"size", "0.0028"}, {"color", "0"}, {"scale", "1.0"}, {"dx", "0.0"},
        {"dy", "0.0"}, {"dz", "0.0"}, {"model_debug", "false"}, {"postprocessing", "true"},
        {"intermediateMesh", "false"}, {"outFile", "./out/mesh.off"}, {"undistort", "false"}};
    Args parser(argc, argv, defaults);
    if (argc < 2) {
        std::cout << about
                  << "  -c=5 voxel carving, -c=6 benchmark; -images -masks -poses | -scene, "
                     "-calibration, -carve, -x -y -z -size, -color, -scale -dx -dy -dz,\n"
                     "  -postprocessing, -intermediateMesh, -outFile (flags of the reference's "
                     "src/main.cpp:16-39)\n";
        return 0;
    }
    const int choose = parser.i("c");
    if (choose != 5 && choose != 6) {
        std::cerr << "-c=" << choose
                  << ": board creation, camera calibration, pose estimation and segmentation are "
                     "OpenCV/aruco programs of the reference and not part of this library (use "
                     "-c=5 or -c=6)"
                  << std::endl;
        return 1;
    }
    const char *tag = choose == 5 ? "VC" : "Benchmark";
    if (choose == 5) {
        const int carveArg = parser.i("carve");
        if (carveArg < 1 || 2 < carveArg) {
            std::cerr << "Invalid carve argument.";
            return 1;
        }
    }
    Inputs in;
    if (!parser.str("scene").empty()) {
        if (!load_scene(parser.str("scene"), in)) {
            std::cerr << "Could not read the scene file (--scene)" << std::endl;
            return 1;
        }
        std::cout << "LOG - " << tag << ": images read." << std::endl;
        std::cout << "LOG - " << tag << ": masks read." << std::endl;
    } else {
        const std::string image_dir = parser.str("images");
        if (image_dir.empty()) {
            std::cerr << "You need to define a images path (--images)" << std::endl;
            return 1;
        }
        if (choose == 5 && parser.f("scale") == 0)
            std::cerr << "The program won't compute stuff for fun! Enter a scale != 0.0" << std::endl;
        for (const std::string &name : glob_sorted(image_dir)) {
            Picture p;
            if (read_pnm(name, p, true)) in.images.push_back(std::move(p));
        }
        std::cout << "LOG - " << tag << ": images read." << std::endl;
        const std::string masks_dir = parser.str("masks");
        if (masks_dir.empty()) {
            std::cerr << "You need to define a image masks path (--masks)" << std::endl;
            return 1;
        }
        for (const std::string &name : glob_sorted(masks_dir)) {
            Picture p;
            if (read_pnm(name, p, true)) in.masks.push_back(std::move(p));
        }
        std::cout << "LOG - " << tag << ": masks read." << std::endl;
        if (in.images.size() != in.masks.size()) {
            std::cerr << "Number of images doesn't match number of masks." << std::endl;
            return 1;
        }
        if (in.images.empty()) {
            std::cerr << "No readable PPM/PGM files in --images" << std::endl;
            return 1;
        }
        in.views.resize(in.images.size());
        for (size_t i = 0; i < in.views.size(); ++i) {
            const Picture &im = in.images[i], &mk = in.masks[i];
            in.views[i].image = {im.px.data(), im.w, im.h, im.c, (size_t)im.w * im.c};
            in.views[i].mask = {mk.px.data(), mk.w, mk.h, mk.c, (size_t)mk.w * mk.c};
        }
        if (parser.str("poses").empty() || !load_poses(parser.str("poses"), in.views)) {
            std::cerr << "You need to give the world->camera matrices of the views (--poses: 12 "
                         "floats per view); pose estimation itself is OpenCV-aruco"
                      << std::endl;
            return 1;
        }
    }
    int x = 0, y = 0, z = 0;
    float size = 0;
    if (choose == 5) {
        x = parser.i("x");
        y = parser.i("y");
        z = parser.i("z");
        if (x < 1 || y < 1 || z < 1) {
            std::cerr << "You need to define a valid number of voxels for the model. (--x/--y/--z)";
            return 1;
        }
        size = parser.f("size");
        if (size <= 0) {
            std::cerr << "You need to define a strictly positive voxel size. (--size)";
            return 1;
        }
    }
    if (parser.str("calibration").empty()) {
        std::cerr << "You need to define the path to the calibration file (--calibration)"
                  << std::endl;
        return 1;
    }
    double K[9];
    if (arvx::readCameraParameters(parser.str("calibration"), K, in.dist)) {
        for (int k = 0; k < 9; ++k) in.intr.K[k] = (float)K[k];  // convertTo(CV_32F), :29-30
    } else if (parser.str("scene").empty()) {
        std::cerr << "Invalid camera file" << std::endl;  // src/PoseEstimation.h:13
        return 1;
    }
    std::cout << "LOG - " << tag << ": read cameraMatrix and distCoefficients." << std::endl;
    if (parser.b("undistort")) {
        // raw inputs: what the reference does per view inside carve / the colour pass
        // (cv::undistort of mask and image, src/VoxelCarving.cpp:35-36), once, on the device
        arvx_ctx *u = nullptr;
        if (arvx_ctx_create(&u, 0, 8, 8, 8, 1.f) != ARVX_OK) return 3;
        const int V = (int)in.views.size();
        std::vector<const uint8_t *> src(V);
        std::vector<uint8_t *> dst(V);
        int rc = ARVX_OK;
        for (int pass = 0; pass < 2 && rc == ARVX_OK; ++pass) {
            for (int i = 0; i < V; ++i) {
                const arvx::Image &im = pass ? in.views[i].image : in.views[i].mask;
                src[i] = im.data;
                dst[i] = const_cast<uint8_t *>(im.data);
            }
            const arvx::Image &f = pass ? in.views[0].image : in.views[0].mask;
            if (!f.data) continue;
            rc = arvx_undistort(u, V, src.data(), f.width, f.height, f.channels, f.stride, K,
                                in.dist.data(), (int)in.dist.size(), dst.data());
        }
        arvx_ctx_destroy(u);
        if (rc != ARVX_OK) {
            std::cerr << "arvx_undistort: " << arvx_last_error() << std::endl;
            return 3;
        }
        std::cout << "LOG - " << tag << ": undistorted masks and images." << std::endl;
    }
    const arvx::Vec3f modelTranslation(parser.f("dx"), parser.f("dy"), parser.f("dz"));
    try {
        if (choose == 6) {
            bench6::run(in.intr, in.views, parser.f("scale"), modelTranslation);
            return 0;
        }
        // 100, 100, 100, 0.0028 ~ Caruco
        arvx::Model model(x, y, z, size);
        switch (parser.i("carve")) {
            case 1: arvx::carve(in.intr, model, in.views, parser.b("intermediateMesh")); break;
            case 2: arvx::fastCarve(in.intr, model, in.views); break;
        }
        const int color = parser.i("color");
        if (color < 0 || 3 < color) {
            std::cerr << "You need to select a predefined color reconstruction mode. (--color)";
            return 1;
        }
        switch (color) {
            case 0: break;
            case 1: arvx::reconstructClosestColor(in.intr, model, in.views); break;
            case 2: arvx::reconstructAvgColor(in.intr, model, in.views); break;
            default: std::cerr << "Ups, something went wrong!" << std::endl;
        }
        model.handleUnseen();
        if (parser.b("model_debug"))
            std::cerr << "-model_debug: the reference's debug cube mesh (Model::WriteModel) is not "
                         "part of this library"
                      << std::endl;
        if (parser.b("postprocessing")) arvx::applyClosure(&model, 3);
        arvx::marchingCubes(&model, parser.f("scale"), modelTranslation, 0.5f, parser.str("outFile"));
    } catch (const arvx::Error &e) {
        return 3;  // (already logged)
    }
    return 0;
}
