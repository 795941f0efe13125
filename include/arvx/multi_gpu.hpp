// multi_gpu.hpp -- the reference's carve() over several GPUs of one node from one process
// (links libarvx_mgpu.so: libarvx.so + RCCL; include/arvx/arvx_mgpu.h).
//
//   arvx::carve(intr, model, views, devices)      reference src/VoxelCarving.h:19
//
// Same effect on `model` as arvx::carve(intr, model, views): device r carves the 8-plane
// groups r, r + n, ... of the grid, one RCCL collective merges the bit-packed occupancy
// (all-reduce, or an all-gather of compressed packets), and the model takes occupied and
// seen planes back from the devices that own them.  The merged occupancy stays on every
// device for consumers that want it there (MultiGpuCarver::occupancy_device_ptr).
#ifndef ARVX_MULTI_GPU_HPP
#define ARVX_MULTI_GPU_HPP

#include <vector>

#include "arvx/arvx_mgpu.h"
#include "arvx/voxel_carving.hpp"

namespace arvx {

// the devices, their contexts and the communicator: keep one for repeated carves of one grid
class MultiGpuCarver {
   public:
    MultiGpuCarver(const std::vector<int> &devices, int X, int Y, int Z, float voxel_size)
        : X_(X), Y_(Y), Z_(Z), s_(voxel_size) {
        detail::check(arvx_mgpu_create(&m_, devices.data(), (int)devices.size(), X, Y, Z,
                                       voxel_size),
                      "arvx_mgpu_create");
    }
    ~MultiGpuCarver() { arvx_mgpu_destroy(m_); }
    MultiGpuCarver(const MultiGpuCarver &) = delete;
    MultiGpuCarver &operator=(const MultiGpuCarver &) = delete;

    // carve `model` by all views; merge = ARVX_MERGE_ALLREDUCE | ARVX_MERGE_COMPRESSED.
    // Returns whether a compressed merge fell back to the all-reduce.
    bool carve(const Intrinsics &intr, Model &model, const std::vector<View> &views,
               int merge = ARVX_MERGE_ALLREDUCE) {
        const int V = (int)views.size();
        if (!V) throw Error(ARVX_ERR_INVALID, "no views");
        // the host planes are copied by the grid size the handle was made for
        if (model.getX() != X_ || model.getY() != Y_ || model.getZ() != Z_ ||
            model.getSize() != s_)
            throw Error(ARVX_ERR_INVALID, "model and MultiGpuCarver differ in grid or voxel size");
        for (const View &v : views)
            if (!v.mask.data || v.mask.width != views[0].mask.width ||
                v.mask.height != views[0].mask.height ||
                v.mask.channels != views[0].mask.channels || v.mask.stride != views[0].mask.stride)
                throw Error(ARVX_ERR_INVALID, "all masks must share one size and layout");
        std::vector<float> M((size_t)V * 12), cam((size_t)V * 3);
        std::vector<const uint8_t *> masks(V);
        for (int i = 0; i < V; ++i) {
            if (views[i].has_M)
                for (int k = 0; k < 12; ++k) M[12 * (size_t)i + k] = views[i].M[k];
            else
                detail::check(arvx_compose_projection(intr.K, views[i].pose, &M[12 * (size_t)i]),
                              "arvx_compose_projection");
            cam[3 * (size_t)i] = views[i].pose[3];
            cam[3 * (size_t)i + 1] = views[i].pose[7];
            cam[3 * (size_t)i + 2] = views[i].pose[11];
            masks[i] = views[i].mask.data;
        }
        const Image &m0 = views[0].mask;
        detail::check(arvx_mgpu_set_views(m_, V, M.data(), cam.data(), masks.data(), m0.width,
                                          m0.height, m0.channels, m0.stride),
                      "arvx_mgpu_set_views");
        if (model.pristine())
            detail::check(arvx_mgpu_state_reset(m_), "arvx_mgpu_state_reset");
        else
            detail::check(arvx_mgpu_state_upload_planes(m_, model.occ_plane(), model.seen_plane()),
                          "arvx_mgpu_state_upload_planes");
        int fell_back = 0;
        detail::check(arvx_mgpu_carve(m_, 0, merge, &fell_back), "arvx_mgpu_carve");
        detail::check(arvx_mgpu_state_download_planes(m_, model.occ_plane_for_writing(),
                                                      model.seen_plane_for_writing()),
                      "arvx_mgpu_state_download_planes");
        model.planes_replaced();
        return fell_back != 0;
    }
    arvx_mgpu *handle() { return m_; }

   private:
    arvx_mgpu *m_ = nullptr;
    int X_, Y_, Z_;
    float s_;
};

inline void carve(const Intrinsics &intr, Model &model, const std::vector<View> &views,
                  const std::vector<int> &devices, int merge = ARVX_MERGE_ALLREDUCE) {
    std::cout << "LOG - VC: starting carving process (version 1)." << std::endl;
    detail::timing(kStageCarving, true);
    MultiGpuCarver carver(devices, model.getX(), model.getY(), model.getZ(), model.getSize());
    carver.carve(intr, model, views, merge);
    detail::timing(kStageCarving, false);
    std::cout << "LOG - VC: carving complete." << std::endl;
}

}  // namespace arvx
#endif
