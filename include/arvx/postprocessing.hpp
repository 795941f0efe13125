// postprocessing.hpp -- the reference's applyClosure(Model*, int kernelSize)
// (src/Postprocessing3d.h:10, src/Postprocessing3d.cpp:4-100) over the GPU library.
//
// Same name, arguments, return value and log lines.  As in the reference the
// result is one dilation with a kernelSize^3 box (its erosion half tests w < 0
// and never fires); new voxels get the mean RGBA of their occupied
// neighbours.  Limitation: explicit colours must have w == 1 (every colour the
// reference's own pipeline produces has).  Runs on the model's own device context: state and
// the colour list of the colour pass are already there (src/main.cpp:282-299).
#ifndef ARVX_POSTPROCESSING_HPP
#define ARVX_POSTPROCESSING_HPP

#include "arvx/voxel_carving.hpp"

namespace arvx {

inline int applyClosure(Model *model, int kernelSize) {
    std::cout << "LOG - PP: starting postprocessing." << std::endl;
    if (kernelSize % 2 != 1) {
        std::cerr << "Invalid kernel size for post processing, skipping..." << std::endl;
        return -1;
    }
    detail::timing(kStagePostProcessing, true);
    arvx_ctx *ctx = model->device();
    // painted voxels: when the paint is exactly "not seen" (handleUnseen was the last thing that
    // touched the state) the device derives it itself; otherwise the byte plane carries it (bit2)
    const bool painted = model->painted();
    if (painted && !model->paint_is_unseen()) {
        const std::vector<uint8_t> st = model->byte_state();
        detail::check(arvx_state_upload(ctx, st.data()), "arvx_state_upload");
        model->set_colors_on_device(false);
    }
    if (!model->colors_on_device()) {
        std::vector<int64_t> idx;
        std::vector<float> rgb;
        if (!model->sorted_colors(idx, rgb)) {
            std::cerr << "LOG(ERR) - GPU: applyClosure needs explicit colours with w == 1"
                      << std::endl;
            throw Error(ARVX_ERR_INVALID, "explicit colour with w != 1");
        }
        detail::check(arvx_colors_upload(ctx, (int64_t)idx.size(), idx.data(), rgb.data()),
                      "arvx_colors_upload");
        model->set_colors_on_device(true);
    }
    std::cout << "LOG - PP: starting dilution." << std::endl;
    detail::check(arvx_closure(ctx, kernelSize, (painted && model->paint_is_unseen()) ? 1 : 0),
                  "arvx_closure");
    int64_t n = 0;
    detail::check(arvx_closure_count(ctx, &n), "arvx_closure_count");
    // straight into the model's own arrays (recycled page-locked memory, host_pool.hpp)
    HostVector<int> fidx((size_t)n);
    HostVector<Vec4f> frgba((size_t)n);
    if (n) detail::check(arvx_closure_download32(ctx, fidx.data(), frgba[0].data()),
                         "arvx_closure_download32");
    std::cout << "LOG - PP: starting erosion." << std::endl;  // a no-op in the reference too
    // the filled voxels: occupied on the device already, with their colours on the host now;
    // the context keeps its closure result, so the model goes back to "host is current" and a
    // second closure starts from a fresh upload
    model->device_changed_keep_paint();
    model->set_closure_list(std::move(fidx), std::move(frgba), true);
    model->closure_applied();
    detail::timing(kStagePostProcessing, false);
    std::cout << "LOG - PP: postprocessing completed." << std::endl;
    return 0;
}

}  // namespace arvx
#endif
