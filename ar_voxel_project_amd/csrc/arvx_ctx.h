// arvx_ctx.h -- the opaque context behind the C-ABI (include/arvx/arvx.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace arvx {

int fail_hip(hipError_t e, const char *what, const char *file, int line);
int fail_msg(int code, const char *msg);

#define ARVX_HIP(call)                                                        \
    do {                                                                      \
        hipError_t arvx_e_ = (call);                                          \
        if (arvx_e_ != hipSuccess)                                            \
            return ::arvx::fail_hip(arvx_e_, #call, __FILE__, __LINE__);      \
    } while (0)

struct Ctx {
    int device = 0;
    int X = 0, Y = 0, Z = 0;  // full grid
    int z0 = 0, z1 = 0;       // slab held here
    float s = 0.f;
    size_t nvox = 0;  // slab voxels

    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    uint8_t *d_state_own = nullptr;
    uint8_t *d_state = nullptr;  // own or bound
    unsigned long long *d_stats = nullptr;
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;

    // views
    bool views_ready = false;
    bool has_campos = false;
    int V = 0, W = 0, H = 0;
    int bgWords = 0, satStride = 0;
    float *d_M = nullptr;
    float *d_campos = nullptr;
    uint32_t *d_bg = nullptr;
    int *d_sat = nullptr;
    std::vector<float> h_M, h_campos;

    // colour pass
    uint8_t *d_images = nullptr;  // V x H x W x 3, BGR
    bool images_ready = false;
    int *d_surf_index = nullptr;      // compacted flat indices (slab-local)
    float *d_surf_rgb = nullptr;      // 3 floats per surface voxel
    uint8_t *d_surf_has = nullptr;    // 1 if the voxel received >= 1 sample
    int64_t surf_count = 0;           // occupied non-inner voxels found
    int64_t surf_capacity = 0;
    bool color_ready = false;

    void free_views() {
        if (d_M) (void)hipFree(d_M);
        if (d_campos) (void)hipFree(d_campos);
        if (d_bg) (void)hipFree(d_bg);
        if (d_sat) (void)hipFree(d_sat);
        d_M = d_campos = nullptr;
        d_bg = nullptr;
        d_sat = nullptr;
        views_ready = false;
    }
    void free_color() {
        if (d_images) (void)hipFree(d_images);
        if (d_surf_index) (void)hipFree(d_surf_index);
        if (d_surf_rgb) (void)hipFree(d_surf_rgb);
        if (d_surf_has) (void)hipFree(d_surf_has);
        d_images = nullptr;
        d_surf_index = nullptr;
        d_surf_rgb = nullptr;
        d_surf_has = nullptr;
        images_ready = color_ready = false;
        surf_count = surf_capacity = 0;
    }
};

}  // namespace arvx

struct arvx_ctx : arvx::Ctx {};
