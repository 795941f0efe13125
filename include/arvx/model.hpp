// model.hpp -- host-side voxel model with the reference's Model API
// (reference src/Model.h:93-163, src/Model.cpp:9-47), header only.
//
// Same public methods, same results, different storage: the reference keeps
// 16 B of RGBA floats + a 24 B std::vector header + 1 bit per voxel (40 GiB at
// 1024^3).  Here a voxel is ONE byte -- bit0 occupied (w != 0), bit1 seen,
// bit2 "painted with UNSEEN_COLOR" -- which is exactly the state plane the
// GPU library works on (include/arvx/arvx.h); colours and colour lists are
// sparse, because only surface voxels ever get one.  get() reconstructs the
// Vector4f the reference would hold.
//
// Vec4f/Vec3i below stand in for Eigen::Vector4f / cv::Vec3i so that this header
// needs neither library; include/arvx/opencv_dropin.hpp maps them when the real
// ones are present.
#ifndef ARVX_MODEL_HPP
#define ARVX_MODEL_HPP

#include <algorithm>
#include <cstdint>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

namespace arvx {

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
    bool operator==(const Vec4f &o) const {
        return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3];
    }
    bool operator!=(const Vec4f &o) const { return !(*this == o); }
};

struct Vec3i {
    int v[3];
    Vec3i() : v{0, 0, 0} {}
    Vec3i(int a, int b, int c) : v{a, b, c} {}
    int &operator()(int i) { return v[i]; }
    int operator()(int i) const { return v[i]; }
};

struct DCLR {  // reference src/Model.h:70-73
    Vec4f color;
    float depth;
};

inline Vec4f model_color() { return Vec4f(50, 168, 141, 1); }  // MODEL_COLOR, src/Model.h:90
inline Vec4f unseen_color() { return Vec4f(204, 0, 0, 1); }    // UNSEEN_COLOR, src/Model.h:91

class Model {
   public:
    static constexpr uint8_t kOcc = 1, kSeen = 2, kUnseenPaint = 4;

    // reference src/Model.cpp:9-14: every voxel MODEL_COLOR, nothing seen
    Model(int x, int y, int z, float size)
        : size_x(x), size_y(y), size_z(z), voxel_size(size),
          state_((size_t)x * y * z, kOcc) {}

    void set(int x, int y, int z, const Vec4f &v) {  // src/Model.cpp:16-18
        const int i = flatten(x, y, z);
        pristine_ = false;
        uint8_t &s = state_[i];
        s = (uint8_t)((s & kSeen) | (v.w() != 0 ? kOcc : 0));
        if (v == model_color() || (v.w() == 0 && v.x() == 0 && v.y() == 0 && v.z() == 0))
            colors_.erase(i);  // the two values the state byte alone encodes
        else
            colors_[i] = v;
    }
    void set(Vec3i voxel, const Vec4f &value) { set(voxel(0), voxel(1), voxel(2), value); }

    int getX() const { return size_x; }
    int getY() const { return size_y; }
    int getZ() const { return size_z; }
    float getSize() const { return voxel_size; }

    Vec4f get(int x, int y, int z) const {  // src/Model.h:119-124: zero outside the grid
        if (x < 0 || x >= size_x || y < 0 || y >= size_y || z < 0 || z >= size_z)
            return Vec4f(0, 0, 0, 0);
        const int i = flatten(x, y, z);
        auto it = colors_.find(i);
        if (it != colors_.end()) return it->second;
        const uint8_t s = state_[i];
        if (s & kUnseenPaint) return unseen_color();
        return (s & kOcc) ? model_color() : Vec4f(0, 0, 0, 0);
    }

    bool isInner(int x, int y, int z) const {  // src/Model.h:126-132
        return occ(x - 1, y, z) && occ(x + 1, y, z) && occ(x, y - 1, z) && occ(x, y + 1, z) &&
               occ(x, y, z - 1) && occ(x, y, z + 1);
    }

    // src/Model.h:134-140 -- note the x/y swap and the negated z
    Vec4f toWord(int x, int y, int z) const {
        return Vec4f(y * voxel_size, x * voxel_size, -1 * z * voxel_size, 1);
    }
    Vec4f toWord(Vec3i v) const { return toWord(v(0), v(1), v(2)); }

    void addColor(int x, int y, int z, const Vec4f &color, float depth) {  // src/Model.h:142-145
        color_lists_[flatten(x, y, z)].push_back(DCLR{color, depth});
    }
    std::vector<DCLR> getColors(int x, int y, int z) const {  // src/Model.h:147-149
        auto it = color_lists_.find(flatten(x, y, z));
        return it == color_lists_.end() ? std::vector<DCLR>() : it->second;
    }

    void see(int x, int y, int z) {  // src/Model.h:151
        pristine_ = false;
        state_[flatten(x, y, z)] |= kSeen;
    }
    void visit(Vec3i v) { see(v(0), v(1), v(2)); }                        // :154-156
    bool visited(Vec3i v) const { return state_[flatten(v(0), v(1), v(2))] & kSeen; }  // :158-160

    void handleUnseen() {  // src/Model.cpp:36-47
        std::cout << "LOG - PP: marking unseen voxels from model." << std::endl;
        pristine_ = false;
        painted_ = true;
        for (size_t i = 0; i < state_.size(); ++i)
            if (!(state_[i] & kSeen)) {
                state_[i] = (uint8_t)(kOcc | kUnseenPaint);
                colors_.erase((int)i);
            }
    }

    std::string to_string() const {  // src/Model.cpp:20-34
        std::ostringstream ss;
        for (int z = 0; z < getZ(); z++) {
            ss << "z = " << z << ":\n";
            for (int y = 0; y < getY(); y++) {
                for (int x = 0; x < getX(); x++) {
                    Vec4f v = get(x, y, z);
                    ss << '(' << v.x() << ", " << v.y() << ", " << v.z() << ", " << v.w() << ')';
                }
                ss << '\n';
            }
            ss << '\n';
        }
        return ss.str();
    }

    // ---- access for the GPU path (not in the reference) ----
    size_t voxels() const { return state_.size(); }
    // still exactly as constructed (every voxel MODEL_COLOR, nothing seen): the GPU
    // side can start from arvx_state_reset instead of an N-byte upload
    bool pristine() const { return pristine_; }
    // no explicit colours and no UNSEEN paint: the state bytes are the whole model, so
    // a device plane can be downloaded straight into state_data()
    bool plain() const { return !painted_ && colors_.empty(); }
    uint8_t *state_data() {
        pristine_ = false;  // the caller may write through the pointer
        return state_.data();
    }
    const uint8_t *state_data() const { return state_.data(); }
    // state bytes as the C-ABI wants them (bit2 is host-only)
    std::vector<uint8_t> device_state() const {
        std::vector<uint8_t> s(state_);
        for (auto &b : s) b &= (uint8_t)(kOcc | kSeen);
        return s;
    }
    // one byte per voxel, 1 where marching cubes counts the voxel as inside (w >= threshold,
    // the complement of src/MarchingCubes.h:481): the occupancy the cell walk runs on
    std::vector<uint8_t> inside_state(float threshold) const {
        std::vector<uint8_t> s(state_.size());
        const uint8_t one = (1.0f >= threshold) ? kOcc : 0;  // MODEL / UNSEEN colour: w = 1
        for (size_t i = 0; i < s.size(); ++i) s[i] = (state_[i] & kOcc) ? one : 0;
        for (const auto &kv : colors_) s[(size_t)kv.first] = (kv.second.w() >= threshold) ? kOcc : 0;
        return s;
    }
    // take the carve result back: occupancy and seen from the device plane.  The
    // device plane started from this model's bits, so it already holds them; a
    // voxel that is still occupied keeps its UNSEEN paint bit, a carved one loses
    // it and its explicit colour (set(x,y,z,(0,0,0,0)), src/VoxelCarving.cpp:52).
    void absorb_state(const uint8_t *dev_state) {
        pristine_ = false;
        for (auto it = colors_.begin(); it != colors_.end();) {  // sparse: surface voxels only
            const size_t i = (size_t)it->first;
            if ((state_[i] & kOcc) && !(dev_state[i] & kOcc)) it = colors_.erase(it);
            else ++it;
        }
        uint8_t *st = state_.data();
        const size_t n = state_.size();
        for (size_t i = 0; i < n; ++i) {  // branch-free, vectorises
            const uint8_t now = (uint8_t)(dev_state[i] & (kOcc | kSeen));
            st[i] = (uint8_t)(now | (st[i] & kUnseenPaint & (uint8_t)((now & kOcc) << 2)));
        }
    }
    void set_flat(int i, const Vec4f &v) {
        const int x = i % size_x, y = (i / size_x) % size_y, z = i / (size_x * size_y);
        set(x, y, z, v);
    }
    size_t colored_voxels() const { return colors_.size(); }
    // explicit colours of occupied voxels, ascending flat index (for arvx_colors_upload)
    std::vector<std::pair<int, Vec4f>> sorted_colors() const {
        std::vector<std::pair<int, Vec4f>> v;
        v.reserve(colors_.size());
        for (const auto &kv : colors_)
            if (kv.second.w() != 0) v.push_back(kv);
        std::sort(v.begin(), v.end(),
                  [](const std::pair<int, Vec4f> &a, const std::pair<int, Vec4f> &b) {
                      return a.first < b.first;
                  });
        return v;
    }

   private:
    const int size_x, size_y, size_z;
    const float voxel_size;
    std::vector<uint8_t> state_;
    bool pristine_ = true;
    bool painted_ = false;  // handleUnseen() ran: some bytes may carry kUnseenPaint
    std::unordered_map<int, Vec4f> colors_;
    std::unordered_map<int, std::vector<DCLR>> color_lists_;

    int flatten(int x, int y, int z) const {  // src/Model.h:104-106
        return x + getX() * (y + getY() * z);
    }
    bool occ(int x, int y, int z) const {
        if (x < 0 || x >= size_x || y < 0 || y >= size_y || z < 0 || z >= size_z) return false;
        return state_[flatten(x, y, z)] & kOcc;
    }
};

}  // namespace arvx
#endif
