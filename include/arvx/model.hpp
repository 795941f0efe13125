// model.hpp -- host-side voxel model with the reference's Model API
// (reference src/Model.h:93-163, src/Model.cpp:9-47), header only, over libarvx.so.
//
// Same public methods, same results, different storage.  The reference keeps 16 B of RGBA
// floats + a 24 B std::vector header + 1 bit per voxel (40 GiB at 1024^3).  Here a voxel is
// TWO BITS -- occupied (w != 0) and seen -- in two bit planes whose rows are padded to 32-bit
// words: exactly the form the GPU library exchanges (arvx_state_upload_planes /
// _download_planes, include/arvx/arvx.h), so a carve result crosses PCIe as N / 4 bytes.  A third
// plane marks voxels painted UNSEEN_COLOR by handleUnseen(); colours are sparse (a sorted list,
// as the colour pass and the closure deliver them, plus a small map for single set() calls),
// because only surface voxels ever get one.  get() reconstructs the Vector4f the reference
// would hold.
//
// The model owns ONE device context, created on first use (carve, colour, closure, marching
// cubes) and kept until the model dies: state, masks, summed-area tables and the colour list
// stay on the GPU between the stages of src/main.cpp:262-303.  Host and device copies of the
// state are synchronised lazily: a stage leaves its result on the device, and the first host
// accessor that needs it (get, isInner, visited, ...) downloads the two bit planes.
//
// Vec4f / Vec3i are Eigen::Vector4f / cv::Vec3i where those libraries are installed and
// stand-ins with the same members otherwise (include/arvx/vec_types.hpp).
#ifndef ARVX_MODEL_HPP
#define ARVX_MODEL_HPP

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "arvx/arvx.h"
#include "arvx/host_pool.hpp"
#include "arvx/vec_types.hpp"

namespace arvx {

struct DCLR {  // reference src/Model.h:70-73
    Vec4f color;
    float depth;
};

inline Vec4f model_color() { return Vec4f(50, 168, 141, 1); }  // MODEL_COLOR, src/Model.h:90
inline Vec4f unseen_color() { return Vec4f(204, 0, 0, 1); }    // UNSEEN_COLOR, src/Model.h:91

class Error : public std::runtime_error {
   public:
    Error(int rc, const std::string &what) : std::runtime_error(what), code(rc) {}
    int code;
};

namespace detail {

inline void check(int rc, const char *what) {
    if (rc != ARVX_OK) {
        std::string msg = std::string(what) + ": " + arvx_last_error();
        std::cerr << "LOG(ERR) - GPU: " << msg << std::endl;
        throw Error(rc, msg);
    }
}

// the model's GPU context and what is known about its contents
struct DeviceLink {
    arvx_ctx *ctx = nullptr;
    std::vector<void *> registered;  // host ranges page-locked for the transfers
    ~DeviceLink() {
        for (void *p : registered) (void)arvx_host_unregister(p);
        if (ctx) arvx_ctx_destroy(ctx);
    }
};

}  // namespace detail

class Model {
   public:
    // reference src/Model.cpp:9-14: every voxel MODEL_COLOR, nothing seen
    Model(int x, int y, int z, float size)
        : size_x(x), size_y(y), size_z(z), voxel_size(size), wpr_((x + 31) / 32),
          occ_((size_t)wpr_ * y * z, 0xffffffffu), seen_((size_t)wpr_ * y * z, 0u) {
        if (x & 31) {  // the bits behind the end of every row stay zero
            const uint32_t last = (1u << (x & 31)) - 1u;
            for (size_t i = (size_t)wpr_ - 1; i < occ_.size(); i += (size_t)wpr_) occ_[i] = last;
        }
    }
    // a copy is an independent model: host data only, its own device context when it needs one
    Model(const Model &o)
        : size_x(o.size_x), size_y(o.size_y), size_z(o.size_z), voxel_size(o.voxel_size),
          wpr_(o.wpr_) {
        o.sync_host();
        occ_ = o.occ_;
        seen_ = o.seen_;
        paint_ = o.paint_;
        pristine_ = o.pristine_;
        paint_is_unseen_ = o.paint_is_unseen_;
        odd_w_ = o.odd_w_;
        cidx_ = o.cidx_;
        cval_ = o.cval_;
        fidx_ = o.fidx_;
        fval_ = o.fval_;
        overlay_ = o.overlay_;
        color_lists_ = o.color_lists_;
    }
    Model &operator=(const Model &) = delete;  // (the reference's has const members too)

    void set(int x, int y, int z, const Vec4f &v) {  // src/Model.cpp:16-18
        sync_host();
        if (v.w() != 0.f && v.w() != 1.f) odd_w_ = true;
        const size_t w = word(x, y, z);
        const uint32_t b = 1u << (x & 31);
        if (v.w() != 0) occ_[w] |= b;
        else occ_[w] &= ~b;
        if (!paint_.empty()) paint_[w] &= ~b;
        const int i = flatten(x, y, z);
        const bool by_state = v == model_color() ||
                              (v.w() == 0 && v.x() == 0 && v.y() == 0 && v.z() == 0);
        // the two values the bits alone encode need no entry -- unless an explicit colour
        // of this voxel has to be hidden
        if (by_state && !std::binary_search(cidx_.begin(), cidx_.end(), i) &&
            !std::binary_search(fidx_.begin(), fidx_.end(), i))
            overlay_.erase(i);
        else
            overlay_[i] = v;
        host_changed();
    }
    void set(Vec3i voxel, const Vec4f &value) { set(voxel(0), voxel(1), voxel(2), value); }

    int getX() const { return size_x; }
    int getY() const { return size_y; }
    int getZ() const { return size_z; }
    float getSize() const { return voxel_size; }

    Vec4f get(int x, int y, int z) const {  // src/Model.h:119-124: zero outside the grid
        if (x < 0 || x >= size_x || y < 0 || y >= size_y || z < 0 || z >= size_z)
            return Vec4f(0, 0, 0, 0);
        sync_bits();
        const int i = flatten(x, y, z);
        if (!overlay_.empty()) {  // what set() stored, exactly
            auto it = overlay_.find(i);
            if (it != overlay_.end()) return it->second;
        }
        const size_t w = word(x, y, z);
        const uint32_t b = 1u << (x & 31);
        if (!occ_bit(x, y, z)) return Vec4f(0, 0, 0, 0);  // carved (src/VoxelCarving.cpp:52)
        if (!paint_.empty() && (paint_[w] & b)) return unseen_color();
        if (!fidx_.empty()) {  // filled by the closure (newer than the colour pass's list)
            auto it = std::lower_bound(fidx_.begin(), fidx_.end(), i);
            if (it != fidx_.end() && *it == i) return fval_[(size_t)(it - fidx_.begin())];
        }
        if (!cidx_.empty()) {
            auto it = std::lower_bound(cidx_.begin(), cidx_.end(), i);
            if (it != cidx_.end() && *it == i) return cval_[(size_t)(it - cidx_.begin())];
        }
        return model_color();
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
    // src/Model.h:147-149.  The lists of the LAST colour pass on the device are not kept per
    // voxel (the reference's vector of vectors costs 24 bytes per voxel before the first sample):
    // the pass leaves the list of the voxels it coloured, and a voxel's samples -- the DCLR
    // entries voxel_pass would have appended, in view order -- are recomputed by the device on
    // request (arvx_color_samples) while the context still holds that pass's cameras and images;
    // after another carve or a new set of views the device part is empty.  Samples added with
    // addColor on the host follow the device's.
    std::vector<DCLR> getColors(int x, int y, int z) const {
        std::vector<DCLR> out;
        const int64_t flat = flatten(x, y, z);
        if (link_ && link_->ctx && !sampled_.empty() &&
            std::binary_search(sampled_.begin(), sampled_.end(), flat)) {
            std::vector<arvx_color_sample> smp((size_t)sample_views_);
            const int rc = arvx_color_samples(link_->ctx, 1, &flat, sample_views_, smp.data());
            if (rc == ARVX_OK) {
                for (const arvx_color_sample &c : smp)
                    if (c.valid) out.push_back(DCLR{Vec4f((float)c.r, (float)c.g, (float)c.b, 1.f), c.depth});
            } else {
                // the context no longer holds that pass's views / images (or another number of
                // views): say so once, the device part of the list is gone
                std::fprintf(stderr, "arvx::Model::getColors: the device samples are gone (%s)\n",
                             arvx_last_error());
                sampled_.clear();
            }
        }
        auto it = color_lists_.find((int)flat);
        if (it != color_lists_.end()) out.insert(out.end(), it->second.begin(), it->second.end());
        return out;
    }
    // the colour pass: the voxels it coloured (ascending flat index) and its number of views
    void set_sampled(std::vector<int64_t> &&index, int views) {
        sampled_ = std::move(index);
        sample_views_ = views;
    }

    void see(int x, int y, int z) {  // src/Model.h:151
        sync_host();
        seen_[word(x, y, z)] |= 1u << (x & 31);
        host_changed();
    }
    void visit(Vec3i v) { see(v(0), v(1), v(2)); }  // :154-156
    bool visited(Vec3i v) const {                   // :158-160
        sync_bits();
        return seen_bit(v(0), v(1), v(2));
    }

    // src/Model.cpp:36-47: every voxel that no view saw becomes UNSEEN_COLOR (204, 0, 0, 1) --
    // occupied, whatever it was.  When the current state lives on the device the bit operation
    // runs there (arvx_handle_unseen) and the paint plane is derived at the next download.
    void handleUnseen() {
        std::cout << "LOG - PP: marking unseen voxels from model." << std::endl;
        pristine_ = false;
        if (host_stale_) {
            detail::check(arvx_handle_unseen(link_->ctx), "arvx_handle_unseen");
            paint_pending_ = true;
        } else {
            paint_unseen_host();
            device_stale_ = true;
        }
        paint_is_unseen_ = true;
    }

    // src/Model.cpp:49-107: the debug mesh -- one unit cube (8 vertices, 6 quads in the voxel's
    // colour) per voxel that is there and not inner, x outermost and z innermost, as OFF text.
    // Written in two passes over the surface voxels instead of through two vectors.
    bool WriteModel(const std::string &filename = "./out/model_mesh.off") const {
        std::cout << "LOG - Debug: generating debug mesh from model..." << std::endl;
        std::ofstream out(filename);
        if (!out.is_open()) {
            std::cerr << "LOG(ERR) - Debug: could not open file " << filename
                      << ". Aborting mesh generation!" << std::endl;
            return false;
        }
        struct Cube {
            int x, y, z;
            unsigned r, g, b;
        };
        std::vector<Cube> cubes;
        for (int x = 0; x < getX(); x++)
            for (int y = 0; y < getY(); y++)
                for (int z = 0; z < getZ(); z++) {
                    const Vec4f c = get(x, y, z);
                    if (c(3) == 0 || isInner(x, y, z)) continue;
                    cubes.push_back(Cube{x, y, z, (unsigned)c.x(), (unsigned)c.y(), (unsigned)c.z()});
                }
        out << "OFF" << std::endl;
        out << cubes.size() * 8 << " " << cubes.size() * 6 << " 0" << std::endl;
        // corner k of a cube: bit0 = +x ... in the reference's order id, r, u, h, ru, rh, uh, ruh
        static const int corner[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1},
                                         {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
        for (const Cube &c : cubes)
            for (const auto &d : corner)
                out << (float)(c.x + d[0]) << " " << (float)(c.y + d[1]) << " " << (float)(c.z + d[2])
                    << std::endl;
        // front, back, left, right, top, bottom
        static const int quad[6][4] = {{0, 2, 4, 1}, {3, 5, 7, 6}, {0, 2, 6, 3},
                                       {1, 4, 7, 5}, {2, 6, 7, 4}, {0, 3, 5, 1}};
        size_t base = 0;
        for (const Cube &c : cubes) {
            for (const auto &q : quad)
                out << "4 " << base + q[0] << " " << base + q[1] << " " << base + q[2] << " "
                    << base + q[3] << " " << c.r << " " << c.g << " " << c.b << std::endl;
            base += 8;
        }
        out.close();
        std::cout << "LOG - Debug: debug mesh written." << std::endl;
        return true;
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

    // ---- the GPU side (not in the reference; used by voxel_carving.hpp and friends) ----

    size_t voxels() const { return (size_t)size_x * size_y * size_z; }
    // still exactly as constructed (every voxel MODEL_COLOR, nothing seen)
    bool pristine() const { return pristine_; }
    // the paint plane is exactly "not seen" (handleUnseen ran and nothing was seen since): the
    // device can derive it itself (apply_unseen of arvx_closure / arvx_export_model)
    bool paint_is_unseen() const { return paint_is_unseen_; }
    bool painted() const { return paint_is_unseen_ || !paint_.empty(); }

    // The model's device context with the model's CURRENT state in it.
    arvx_ctx *device(int device_index = 0) {
        if (!link_) {
            link_ = std::make_shared<detail::DeviceLink>();
            detail::check(arvx_ctx_create(&link_->ctx, device_index, size_x, size_y, size_z,
                                          voxel_size),
                          "arvx_ctx_create");
            // page-lock the planes: the transfers then run at PCIe rate (best effort)
            for (void *p : {(void *)occ_.data(), (void *)seen_.data()})
                if (arvx_host_register(p, occ_.size() * sizeof(uint32_t)) == ARVX_OK)
                    link_->registered.push_back(p);
            device_stale_ = true;
        }
        if (closure_on_device_ && !mc_only_) {  // see closure_applied()
            sync_host();
            device_stale_ = true;
            closure_on_device_ = false;
        }
        if (device_stale_) colors_on_device_ = false;  // an upload drops the context's lists
        if (device_stale_) {
            if (pristine_) {
                detail::check(arvx_state_reset(link_->ctx), "arvx_state_reset");
            } else {
                detail::check(arvx_state_upload_planes(link_->ctx, occ_.data(), seen_.data()),
                              "arvx_state_upload_planes");
            }
            device_stale_ = false;
        }
        return link_->ctx;
    }
    // the device context for a stage that only READS the occupancy (marching-cubes cells)
    // the context if the model has one already (nothing is created or uploaded)
    arvx_ctx *device_if_any() const { return link_ ? link_->ctx : nullptr; }
    arvx_ctx *device_for_reading(int device_index = 0) {
        mc_only_ = true;
        arvx_ctx *c = device(device_index);
        mc_only_ = false;
        return c;
    }
    // a stage changed occupancy / seen on the device: the host planes are out of date
    void device_changed() {
        pristine_ = false;
        host_stale_ = true;
        paint_is_unseen_ = false;  // (carving sees voxels; closure adds some)
        colors_on_device_ = false;  // (arvx_carve* drops the context's colour list)
        sampled_.clear();           // (... and with it what getColors could ask the device for)
    }
    // the closure's result is in the context (filled voxels marked occupied): marching cubes can
    // run on it, but anything else that changes the state has to start from an upload again
    void closure_applied() { closure_on_device_ = true; }
    // the context's sparse colour list equals this model's explicit colours (set by the colour
    // pass, consumed by the closure without another upload)
    bool colors_on_device() const { return colors_on_device_; }
    void set_colors_on_device(bool v) { colors_on_device_ = v; }
    // a stage that only ADDS occupied voxels that are painted / coloured explicitly afterwards
    void device_changed_keep_paint() {
        pristine_ = false;
        host_stale_ = true;
    }

    // Colours of many voxels at once, ascending flat index (what arvx_surface_download and
    // arvx_closure_download deliver): model.set(x, y, z, (r, g, b, a)) for each, without the
    // per-voxel cost.  `occupy`: the voxels become occupied (closure).
    void set_sorted(const std::vector<int64_t> &index, const float *rgba, int channels,
                    bool occupy) {
        if (index.empty()) return;
        fold_closure_list();
        // The planes are only touched to occupy voxels or to clear paint.  Voxels a device stage
        // has just occupied (the closure's: host_stale_) arrive with the planes at the next
        // sync, and they were empty before, hence not painted: nothing to touch.
        const bool touch = (occupy && !host_stale_) || (!occupy && (!paint_.empty() || paint_pending_));
        if (touch) sync_host();
        HostVector<int> idx(index.size());
        HostVector<Vec4f> val(index.size());
        for (size_t k = 0; k < index.size(); ++k) {
            idx[k] = (int)index[k];
            const float *c = rgba + (size_t)channels * k;
            val[k] = Vec4f(c[0], c[1], c[2], channels == 4 ? c[3] : 1.f);
            if (channels == 4 && c[3] != 0.f && c[3] != 1.f) odd_w_ = true;
        }
        if (touch)
            for (size_t k = 0; k < index.size(); ++k) {
                const int x = idx[k] % size_x, y = (idx[k] / size_x) % size_y,
                          z = idx[k] / (size_x * size_y);
                const size_t w = word(x, y, z);
                const uint32_t b = 1u << (x & 31);
                if (occupy && val[k].w() != 0) occ_[w] |= b;
                if (!paint_.empty()) paint_[w] &= ~b;
            }
        if (!overlay_.empty())
            for (size_t k = 0; k < index.size(); ++k) overlay_.erase(idx[k]);
        if (cidx_.empty()) {
            cidx_.swap(idx);
            cval_.swap(val);
        } else {
            merge_into_colors(idx, val);
        }
        pristine_ = false;
    }

    // The closure's result (src/Postprocessing3d.cpp:52-60: model.set for every filled voxel),
    // ascending flat index, as the device delivers it: kept as a list of its own beside the
    // colour pass's -- get() looks there first --, so that a million-entry merge is not part of
    // applyClosure.  The voxels are occupied on the DEVICE already (the caller has said
    // device_changed_keep_paint()): the host planes get them with the next sync.
    // unit_w: the caller vouches that every w is 1 (the device's means of w = 1 neighbours are
    // count / count); otherwise the list is scanned for fractional w (see inside_state).
    void set_closure_list(HostVector<int> &&index, HostVector<Vec4f> &&rgba, bool unit_w) {
        fold_closure_list();  // (a second closure: the first one's list joins the colours)
        if (index.empty()) return;
        if (!unit_w)
            for (const Vec4f &v : rgba)
                if (v.w() != 0.f && v.w() != 1.f) odd_w_ = true;
        if (!host_stale_) {  // the host planes are current: the voxels become occupied here
            for (size_t k = 0; k < index.size(); ++k) {
                const int x = index[k] % size_x, y = (index[k] / size_x) % size_y,
                          z = index[k] / (size_x * size_y);
                if (rgba[k].w() != 0) occ_[word(x, y, z)] |= 1u << (x & 31);
                if (!paint_.empty()) paint_[word(x, y, z)] &= ~(1u << (x & 31));
            }
        }
        if (!overlay_.empty())
            for (size_t k = 0; k < index.size(); ++k) overlay_.erase(index[k]);
        fidx_ = std::move(index);
        fval_ = std::move(rgba);
        pristine_ = false;
    }

    // explicit colours of occupied, unpainted voxels, ascending flat index
    // (for arvx_colors_upload); false if one of them has w != 1 (the device list holds RGB only)
    bool sorted_colors(std::vector<int64_t> &index, std::vector<float> &rgb) const {
        sync_bits();
        std::vector<std::pair<int, Vec4f>> extra(overlay_.begin(), overlay_.end());
        std::sort(extra.begin(), extra.end(),
                  [](const std::pair<int, Vec4f> &a, const std::pair<int, Vec4f> &b) {
                      return a.first < b.first;
                  });
        index.clear();
        rgb.clear();
        bool plain = true;
        auto emit = [&](int i, const Vec4f &v) {
            const int x = i % size_x, y = (i / size_x) % size_y, z = i / (size_x * size_y);
            const size_t w = word(x, y, z);
            const uint32_t b = 1u << (x & 31);
            if (!occ_bit(x, y, z) || (!paint_.empty() && (paint_[w] & b))) return;
            if (v == model_color()) return;  // what the state alone says
            if (v.w() != 1.f) plain = false;
            index.push_back(i);
            rgb.push_back(v.x());
            rgb.push_back(v.y());
            rgb.push_back(v.z());
        };
        // three sorted sources; on equal indices single set() calls win over the closure's list,
        // and that over the colour pass's
        size_t a = 0, f = 0, b = 0;
        const int kEnd = 0x7fffffff;
        for (;;) {
            const int ia = a < cidx_.size() ? cidx_[a] : kEnd, jf = f < fidx_.size() ? fidx_[f] : kEnd,
                      ib = b < extra.size() ? extra[b].first : kEnd;
            const int i = std::min(ia, std::min(jf, ib));
            if (i == kEnd) break;
            if (ib == i) emit(i, extra[b].second);
            else if (jf == i) emit(i, fval_[f]);
            else emit(i, cval_[a]);
            if (ia == i) ++a;
            if (jf == i) ++f;
            if (ib == i) ++b;
        }
        return plain;
    }

    // one byte per voxel as the C-ABI's byte plane wants it: bit0 occupied, bit1 seen,
    // bit2 painted UNSEEN_COLOR (arvx_closure) -- the slow path for models whose paint is not
    // simply "not seen"
    std::vector<uint8_t> byte_state() const {
        sync_host();
        std::vector<uint8_t> s(voxels());
        for (int z = 0; z < size_z; ++z)
            for (int y = 0; y < size_y; ++y)
                for (int x = 0; x < size_x; ++x) {
                    const size_t w = word(x, y, z);
                    const int sh = x & 31;
                    s[(size_t)flatten(x, y, z)] =
                        (uint8_t)(((occ_[w] >> sh) & 1u) | (((seen_[w] >> sh) & 1u) << 1) |
                                  ((!paint_.empty() ? (paint_[w] >> sh) & 1u : 0u) << 2));
                }
        return s;
    }
    // one byte per voxel, 1 where marching cubes counts the voxel as inside (w >= threshold,
    // the complement of src/MarchingCubes.h:481); `same_as_occupancy` tells the caller that the
    // occupancy on the device already is that plane (every w is 0 or 1, 0 < threshold <= 1)
    std::vector<uint8_t> inside_state(float threshold, bool &same_as_occupancy) const {
        same_as_occupancy = threshold > 0.f && threshold <= 1.f;
        // no fractional w was ever stored: nothing to look at (and nothing to bring back from
        // the device -- this is the path of every model the reference's own pipeline builds)
        if (same_as_occupancy && !odd_w_) return {};
        sync_host();
        auto odd = [&](const Vec4f &v) { return v.w() != 0.f && v.w() != 1.f; };
        for (const auto &kv : overlay_) same_as_occupancy = same_as_occupancy && !odd(kv.second);
        for (const Vec4f &v : cval_) same_as_occupancy = same_as_occupancy && !odd(v);
        for (const Vec4f &v : fval_) same_as_occupancy = same_as_occupancy && !odd(v);
        if (same_as_occupancy) return {};
        std::vector<uint8_t> s(voxels());
        for (int z = 0; z < size_z; ++z)
            for (int y = 0; y < size_y; ++y)
                for (int x = 0; x < size_x; ++x)
                    s[(size_t)flatten(x, y, z)] = get(x, y, z).w() >= threshold ? 1 : 0;
        return s;
    }
    size_t colored_voxels() const { return cidx_.size() + fidx_.size() + overlay_.size(); }

    // the two host planes themselves (ceil(X / 32) * Y * Z words each), for stages that fill
    // them from elsewhere (include/arvx/multi_gpu.hpp); call planes_replaced() afterwards
    size_t plane_words() const { return occ_.size(); }
    const uint32_t *occ_plane() const { sync_host(); return occ_.data(); }
    const uint32_t *seen_plane() const { sync_host(); return seen_.data(); }
    uint32_t *occ_plane_for_writing() { sync_host(); return occ_.data(); }
    uint32_t *seen_plane_for_writing() { return seen_.data(); }
    void planes_replaced() { host_changed(); }

    // Bring the host's knowledge of the state up to date with the device (no-op when it is).
    // Where the grid allows it (X % 32 == 0, X * Y % 64 == 0) the state crosses PCIe as two
    // compressed PACKETS (arvx_state_download_packets: bitmaps of the all-one and the mixed 64-bit
    // words + the mixed words; a carved 1024^3 model is 24 MB instead of the planes' 268 MB) and the
    // accessors answer from them -- get / isInner / visited cost one bitmap look-up and at most one
    // more word; the planes are rebuilt from the packets only when somebody needs them as planes
    // (sync_host: a host-side write, an upload, a copy).
    void sync_bits() const {
        if (!host_stale_) return;
        host_stale_ = false;
        int64_t n64 = 0, H = 0;
        if (size_x % 32 == 0 && ((size_t)size_x * size_y) % 64 == 0 &&
            arvx_state_packet_geometry(link_->ctx, &n64, &H) == ARVX_OK) {
            pk_n_ = n64;
            pk_H_ = H;
            // room for what the last hand-off needed (first: a shell's share of the words)
            if (pk_occ_.size() < (size_t)H + 1024) pk_occ_.resize((size_t)(H + std::max<int64_t>(1024, n64 / 16)));
            if (pk_seen_.size() < (size_t)H + 1024) pk_seen_.resize((size_t)(H + std::max<int64_t>(1024, n64 / 64)));
            for (int attempt = 0;; ++attempt) {
                int64_t on = 0, sn = 0;
                detail::check(arvx_state_download_packets(link_->ctx, pk_occ_.data(), (int64_t)pk_occ_.size() - H,
                                                          pk_seen_.data(), (int64_t)pk_seen_.size() - H,
                                                          &on, &sn),
                              "arvx_state_download_packets");
                if ((on <= (int64_t)pk_occ_.size() - H && sn <= (int64_t)pk_seen_.size() - H) || attempt) break;
                // (the device kept the packets: the second call only copies)
                if (on > (int64_t)pk_occ_.size() - H) pk_occ_.resize((size_t)(H + on + on / 8));
                if (sn > (int64_t)pk_seen_.size() - H) pk_seen_.resize((size_t)(H + sn + sn / 8));
            }
            planes_stale_ = true;
        } else {
            detail::check(arvx_state_download_planes(link_->ctx, occ_.data(), seen_.data()),
                          "arvx_state_download_planes");
            planes_stale_ = false;
        }
        Model *self = const_cast<Model *>(this);
        if (paint_pending_) {  // (the paint plane is derived from the seen PLANE)
            paint_pending_ = false;
            expand_planes();
            self->paint_unseen_host();
        }
        // what set() stored for a voxel that the device has carved since is gone
        // (src/VoxelCarving.cpp:52 overwrites the voxel with zeros)
        for (auto it = self->overlay_.begin(); it != self->overlay_.end();) {
            const int i = it->first;
            const int x = i % size_x, y = (i / size_x) % size_y, z = i / (size_x * size_y);
            if (it->second.w() != 0 && !occ_bit(x, y, z)) it = self->overlay_.erase(it);
            else ++it;
        }
    }
    // ... and the two bit PLANES current as well (everything that reads or writes them as arrays)
    void sync_host() const {
        sync_bits();
        expand_planes();
    }
    // A state packet (include/arvx/arvx.h, arvx_state_download_packets) read on the host; n: 64-bit
    // words of the plane, H: header words.  packet_bit: bit i of the plane -- its word is all-one, or
    // mixed (then it is the popcount-th mixed word of its group of 64 words), or all-zero.
    // expand_packet: the whole plane as 2 n 32-bit words.  (Pure functions of the packet: also what
    // tests/test_cpp_host.py::test_packet_decoding_on_the_host checks without a GPU.)
    static bool packet_bit(const uint64_t *pk, int64_t n, int64_t H, size_t i) {
        const int64_t nb = (n + 63) / 64;
        const size_t W = i >> 6, g = W >> 6;
        const unsigned b = (unsigned)(W & 63);
        if ((pk[1 + g] >> b) & 1u) return true;
        const uint64_t m = pk[1 + nb + g];
        if (!((m >> b) & 1u)) return false;
        const uint32_t *goff = reinterpret_cast<const uint32_t *>(pk + 1 + 2 * nb);
        const uint64_t w = pk[H + goff[g] + (uint32_t)__builtin_popcountll(m & ((1ull << b) - 1ull))];
        return (w >> (i & 63)) & 1u;
    }
    static void expand_packet(const uint64_t *pk, int64_t n, int64_t H, uint32_t *out) {
        const int64_t nb = (n + 63) / 64;
        const uint32_t *goff = reinterpret_cast<const uint32_t *>(pk + 1 + 2 * nb);
        for (int64_t g = 0; g < nb; ++g) {
            const uint64_t ones = pk[1 + g], mixed = pk[1 + nb + g];
            const uint64_t *mw = pk + H + goff[g];
            const int64_t w0 = g * 64, w1 = std::min<int64_t>(w0 + 64, n);
            if (!mixed && (ones == 0 || (ones == ~0ull && w1 - w0 == 64))) {  // a uniform group
                std::fill(out + 2 * w0, out + 2 * w1, ones ? 0xffffffffu : 0u);
                continue;
            }
            for (int64_t w = w0; w < w1; ++w) {
                const unsigned b = (unsigned)(w - w0);
                uint64_t v = ((ones >> b) & 1u) ? ~0ull : 0ull;
                if ((mixed >> b) & 1u) v = *mw++;
                out[2 * w] = (uint32_t)v;
                out[2 * w + 1] = (uint32_t)(v >> 32);
            }
        }
    }
    // does the host hold the state as packets only right now?  (tests, tools/dropin_times)
    bool planes_pending() const { return planes_stale_; }
    size_t packet_bytes() const { return planes_stale_ ? 8 * (size_t)(2 * pk_H_ + (int64_t)pk_occ_[0] + (int64_t)pk_seen_[0]) : 0; }

   private:
    const int size_x, size_y, size_z;
    const float voxel_size;
    const int wpr_;  // 32-bit words per voxel row
    mutable std::vector<uint32_t> occ_, seen_;  // bit planes, rows padded to words
    // the same two planes as the device's compressed packets (include/arvx/arvx.h,
    // arvx_state_download_packets), in recycled page-locked memory; planes_stale_: the packets are
    // what is current, occ_ / seen_ are rebuilt from them on demand (expand_planes)
    mutable HostVector<uint64_t> pk_occ_, pk_seen_;
    mutable int64_t pk_n_ = 0, pk_H_ = 0;
    mutable bool planes_stale_ = false;
    mutable std::vector<uint32_t> paint_;       // painted UNSEEN_COLOR (empty: none)
    bool pristine_ = true;
    bool paint_is_unseen_ = false;
    bool odd_w_ = false;  // some colour with w outside {0, 1} was stored (sticky)
    mutable bool paint_pending_ = false;  // handleUnseen ran on the device: derive paint_ at sync
    mutable bool host_stale_ = false;     // the device holds a newer occupancy / seen
    bool device_stale_ = true;            // the host planes changed since the device saw them
    bool colors_on_device_ = false;
    bool closure_on_device_ = false, mc_only_ = false;
    HostVector<int> cidx_;                // explicit colours, ascending flat index ...
    HostVector<Vec4f> cval_;
    HostVector<int> fidx_;                // ... the closure's filled voxels, likewise ...
    HostVector<Vec4f> fval_;
    std::unordered_map<int, Vec4f> overlay_;  // ... and what single set() calls stored since
    std::unordered_map<int, std::vector<DCLR>> color_lists_;
    mutable std::vector<int64_t> sampled_;  // voxels the last device colour pass coloured (getColors)
    int sample_views_ = 0;
    std::shared_ptr<detail::DeviceLink> link_;

    int flatten(int x, int y, int z) const {  // src/Model.h:104-106
        return x + getX() * (y + getY() * z);
    }
    size_t word(int x, int y, int z) const {
        return ((size_t)z * size_y + y) * wpr_ + (size_t)(x >> 5);
    }
    bool occ(int x, int y, int z) const {
        if (x < 0 || x >= size_x || y < 0 || y >= size_y || z < 0 || z >= size_z) return false;
        sync_bits();
        return occ_bit(x, y, z);
    }
    // (in-grid voxel; after sync_bits)
    bool occ_bit(int x, int y, int z) const {
        if (planes_stale_) return packet_bit(pk_occ_.data(), pk_n_, pk_H_, (size_t)flatten(x, y, z));
        return (occ_[word(x, y, z)] >> (x & 31)) & 1u;
    }
    bool seen_bit(int x, int y, int z) const {
        if (planes_stale_) return packet_bit(pk_seen_.data(), pk_n_, pk_H_, (size_t)flatten(x, y, z));
        return (seen_[word(x, y, z)] >> (x & 31)) & 1u;
    }
    // packets -> planes (X % 32 == 0: the planes' 32-bit words are the packets' 64-bit words in halves)
    void expand_planes() const {
        if (!planes_stale_) return;
        planes_stale_ = false;
        expand_packet(pk_occ_.data(), pk_n_, pk_H_, occ_.data());
        expand_packet(pk_seen_.data(), pk_n_, pk_H_, seen_.data());
    }
    // merge a sorted list into the colour list, the new values win
    void merge_into_colors(const HostVector<int> &idx, const HostVector<Vec4f> &val) {
        HostVector<int> mi;
        HostVector<Vec4f> mv;
        mi.reserve(cidx_.size() + idx.size());
        mv.reserve(cidx_.size() + idx.size());
        size_t a = 0, b = 0;
        while (a < cidx_.size() || b < idx.size()) {
            if (b == idx.size() || (a < cidx_.size() && cidx_[a] < idx[b])) {
                mi.push_back(cidx_[a]);
                mv.push_back(cval_[a++]);
            } else {
                if (a < cidx_.size() && cidx_[a] == idx[b]) ++a;
                mi.push_back(idx[b]);
                mv.push_back(val[b++]);
            }
        }
        cidx_.swap(mi);
        cval_.swap(mv);
    }
    void fold_closure_list() {
        if (fidx_.empty()) return;
        if (cidx_.empty()) {
            cidx_.swap(fidx_);
            cval_.swap(fval_);
        } else {
            merge_into_colors(fidx_, fval_);
        }
        HostVector<int>().swap(fidx_);
        HostVector<Vec4f>().swap(fval_);
    }
    void host_changed() {
        pristine_ = false;
        device_stale_ = true;
        closure_on_device_ = false;
        paint_is_unseen_ = false;
        colors_on_device_ = false;
    }
    // occ |= ~seen, paint |= ~seen (inside the grid); explicit colours of those voxels go
    void paint_unseen_host() {
        if (paint_.empty()) paint_.assign(occ_.size(), 0u);
        const uint32_t last = (size_x & 31) ? ((1u << (size_x & 31)) - 1u) : 0xffffffffu;
        for (size_t i = 0; i < occ_.size(); ++i) {
            const uint32_t valid = ((int)(i % wpr_) == wpr_ - 1) ? last : 0xffffffffu;
            const uint32_t u = ~seen_[i] & valid;
            occ_[i] |= u;
            paint_[i] |= u;
        }
        for (auto it = overlay_.begin(); it != overlay_.end();) {
            const int i = it->first;
            const int x = i % size_x, y = (i / size_x) % size_y, z = i / (size_x * size_y);
            if (!((seen_[word(x, y, z)] >> (x & 31)) & 1u)) it = overlay_.erase(it);
            else ++it;
        }
    }
};

}  // namespace arvx
#endif
