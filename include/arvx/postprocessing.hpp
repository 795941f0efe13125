// postprocessing.hpp -- the reference's applyClosure(Model*, int kernelSize)
// (src/Postprocessing3d.h:10, src/Postprocessing3d.cpp:4-100) over the GPU library.
//
// Same name, arguments, return value and log lines.  As in the reference the
// result is one dilation with a kernelSize^3 box (its erosion half tests w < 0
// and never fires); new voxels get the mean RGBA of their occupied
// neighbours.  Limitation: explicit colours must have w == 1 (every colour the
// reference's own pipeline produces has).
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
    arvx_ctx *ctx = nullptr;
    detail::check(arvx_ctx_create(&ctx, 0, model->getX(), model->getY(), model->getZ(),
                                  model->getSize()),
                  "arvx_ctx_create");
    struct Guard {
        arvx_ctx *c;
        ~Guard() { arvx_ctx_destroy(c); }
    } guard{ctx};
    // bit2 of the host state = painted UNSEEN_COLOR by handleUnseen(): keep it
    detail::check(arvx_state_upload(ctx, model->state_data()), "arvx_state_upload");
    const auto cols = model->sorted_colors();
    std::vector<int64_t> idx(cols.size());
    std::vector<float> rgb(cols.size() * 3);
    for (size_t k = 0; k < cols.size(); ++k) {
        idx[k] = cols[k].first;
        rgb[3 * k] = cols[k].second.x();
        rgb[3 * k + 1] = cols[k].second.y();
        rgb[3 * k + 2] = cols[k].second.z();
    }
    detail::check(arvx_colors_upload(ctx, (int64_t)idx.size(), idx.data(), rgb.data()),
                  "arvx_colors_upload");
    std::cout << "LOG - PP: starting dilution." << std::endl;
    detail::check(arvx_closure(ctx, kernelSize, 0), "arvx_closure");
    int64_t n = 0;
    detail::check(arvx_closure_count(ctx, &n), "arvx_closure_count");
    std::vector<int64_t> fidx((size_t)n);
    std::vector<float> frgba((size_t)n * 4);
    if (n) detail::check(arvx_closure_download(ctx, fidx.data(), frgba.data()),
                         "arvx_closure_download");
    std::cout << "LOG - PP: starting erosion." << std::endl;  // a no-op in the reference too
    for (int64_t k = 0; k < n; ++k)
        model->set_flat((int)fidx[k], Vec4f(frgba[4 * k], frgba[4 * k + 1], frgba[4 * k + 2],
                                            frgba[4 * k + 3]));
    detail::timing(kStagePostProcessing, false);
    std::cout << "LOG - PP: postprocessing completed." << std::endl;
    return 0;
}

}  // namespace arvx
#endif
