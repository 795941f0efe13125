// mc_mesh_kernels.h -- the triangles of the reference's marchingCubes() on the device, for the
// models the device can hold: every w is 0 or 1 (src/Model.h:90-91; what carve, the colour
// pass, handleUnseen and the closure produce).
//
// Input: the cell list of mc_kernels.h (cells cut by the surface, with cube index, in the
// reference's visiting order).  Per cell (one thread), Polygonise + ProcessVoxel
// (src/MarchingCubes.h:478-578) reduce to: a cut edge joins an occupied corner and an empty one,
// so its vertex SNAPS to the occupied corner and takes that voxel's colour (VertexInterp,
// :432-441) -- integer lattice coordinates; every triangle of Bourke's table row gets three
// fresh vertices, and its face colour is round((c0 + c1 + c1) / 3) per channel in fp32: the
// third corner's colour is the second's (:506, kept) and MeanColorFloats rounds half away from
// zero (:414-416).
//
// A voxel's colour as Model::get would return it (src/main.cpp:262-299 order): UNSEEN_COLOR if
// handleUnseen painted it, else the closure's mean colour if the closure filled it, else the
// colour pass's result if it has one, else MODEL_COLOR.  The two sparse lists are ordered
// compactions of bit planes the context keeps with per-word ranks (bitplane_kernels.h,
// sparse_find): one lookup each; the voxel's seen bit comes straight from its record.
#pragma once

#include "arvx/mc_tables.hpp"
#include "arvx_device.h"
#include "bitplane_kernels.h"
#include "state_kernels.h"

namespace arvx {

struct McTriTable {
    int8_t e[256][16];  // edge numbers, three per triangle, -1 terminated
    int8_t n[256];      // triangles per cube index
};

constexpr McTriTable make_mc_tri_table() {
    McTriTable t{};
    for (int i = 0; i < 256; ++i) {
        int k = 0;
        for (const char *s = mc::kTriangles[i]; *s; ++s) t.e[i][k++] = (int8_t)mc::hex_digit(*s);
        t.n[i] = (int8_t)(k / 3);
        for (; k < 16; ++k) t.e[i][k] = -1;
    }
    return t;
}

__constant__ McTriTable kMcTri = make_mc_tri_table();

struct McMeshParams {
    CarveParams g;         // the state records (whole grid)
    const unsigned long long *paint;  // voxels painted UNSEEN_COLOR by the host (null: none)
    int apply_unseen;      // never-seen voxels are painted UNSEEN_COLOR
    SparseList col;        // colour pass: plane + rank, rgb, has-sample flag
    const float *col_rgb;
    const uint8_t *col_has;
    SparseList clo;        // closure: plane + rank, rgba
    const float4 *clo_rgba;
};

// rgb of an occupied voxel
__device__ inline float3 mc_voxel_rgb(const McMeshParams &p, int x, int y, int z) {
    const int X = p.g.X, Y = p.g.Y, XW = (X + 63) >> 6;
    if (plane_bit(p.paint, X, Y, x, y, z) || (p.apply_unseen && !(rec_state(p.g, x, y, z) & 2u)))
        return make_float3(204.f, 0.f, 0.f);
    const size_t row = (size_t)z * Y + y;
    int k = sparse_find(p.clo, XW, x, row);
    if (k >= 0) return make_float3(p.clo_rgba[k].x, p.clo_rgba[k].y, p.clo_rgba[k].z);
    k = sparse_find(p.col, XW, x, row);
    if (k >= 0 && p.col_has[k])
        return make_float3(p.col_rgb[3 * k], p.col_rgb[3 * k + 1], p.col_rgb[3 * k + 2]);
    return make_float3(50.f, 168.f, 141.f);
}

__global__ __launch_bounds__(256) void mc_tri_count_kernel(const int4 *__restrict__ cells,
                                                           long long n, int *__restrict__ counts) {
    const long long c = (long long)blockIdx.x * 256 + threadIdx.x;
    if (c < n) counts[c] = kMcTri.n[cells[c].w & 255];
}

__device__ __forceinline__ unsigned mc_mean3(float a, float b, float c) {
    const float s = (a + b) + c;
    return (unsigned)roundf(s / 3.f);
}

constexpr int kMeshMaxTris = 5;  // Bourke's table: at most five triangles per cell

// One WAVE per workgroup, one cell per lane.  The triangles of 64 consecutive cells are one
// contiguous range of the output (tri_offset is the exclusive scan of their counts), so the
// lanes build theirs in LDS and the wave then streams the range out in whole lines: 36 bytes of
// vertices and 24 bytes of face record per triangle leave as coalesced dword stores instead of
// twelve scattered ones per triangle.
//   verts  9 floats per triangle (three fresh vertices, voxel units)
//   faces  6 uints per triangle: vertex numbers 3t, 3t+1, 3t+2, then r, g, b -- the layout of
//          the C++ layer's Triangle (include/arvx/marching_cubes.hpp), reference
//          src/MarchingCubes.h:19-31
// Only the first two corners' colours are ever used (col[2] = col[1], :506): two voxel lookups
// per triangle.
__global__ __launch_bounds__(64) void mc_mesh_kernel(const McMeshParams p,
                                                     const int4 *__restrict__ cells, long long n,
                                                     const long long *__restrict__ tri_offset,
                                                     float *__restrict__ verts,
                                                     unsigned *__restrict__ faces) {
    __shared__ float s_vert[64 * kMeshMaxTris * 9];
    __shared__ unsigned s_rgb[64 * kMeshMaxTris * 3];
    const int lane = threadIdx.x;
    const long long c = (long long)blockIdx.x * 64 + lane;
    const long long c0 = (long long)blockIdx.x * 64;
    const long long clast = (c0 + 63 < n - 1) ? c0 + 63 : n - 1;
    const long long base = tri_offset[c0];
    // corner i of ProcessVoxel (src/MarchingCubes.h:537-552); bit i of idx set = NOT in the model
    // x offsets of corners 0..7: 1 0 0 1 1 0 0 1; y: 0 0 1 1 0 0 1 1; z: 0 0 0 0 1 1 1 1
    constexpr unsigned kCx = 0x99u, kCy = 0xCCu, kCz = 0xF0u;
    if (c < n) {
        const int4 cell = cells[c];
        const int idx = cell.w & 255;
        int t = (int)(tri_offset[c] - base);
        const int8_t *row = kMcTri.e[idx];
        for (int k = 0; row[k] >= 0; k += 3, ++t) {
            int vx[3], vy[3], vz[3];
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int e = row[k + v];
                const int a = e & 7, b = mc::kSecondCorner[e];  // e % 8 and its partner (:491)
                const int cn = ((idx >> a) & 1) ? b : a;         // the corner that is in the model
                vx[v] = cell.x + (int)((kCx >> cn) & 1u);
                vy[v] = cell.y + (int)((kCy >> cn) & 1u);
                vz[v] = cell.z + (int)((kCz >> cn) & 1u);
                float *o = s_vert + 9 * t + 3 * v;
                o[0] = (float)vx[v];
                o[1] = (float)vy[v];
                o[2] = (float)vz[v];
            }
            const float3 q0 = mc_voxel_rgb(p, vx[0], vy[0], vz[0]);
            const float3 q1 = mc_voxel_rgb(p, vx[1], vy[1], vz[1]);  // col[2] = col[1], :506
            s_rgb[3 * t] = mc_mean3(q0.x, q1.x, q1.x);
            s_rgb[3 * t + 1] = mc_mean3(q0.y, q1.y, q1.y);
            s_rgb[3 * t + 2] = mc_mean3(q0.z, q1.z, q1.z);
        }
    }
    __syncthreads();
    const int ntri = (int)(tri_offset[clast] - base) + kMcTri.n[cells[clast].w & 255];
    float *vo = verts + 9 * base;
    for (int i = lane; i < 9 * ntri; i += 64) vo[i] = s_vert[i];
    unsigned *fo = faces + 6 * base;
    for (int i = lane; i < 6 * ntri; i += 64) {
        const int t = i / 6, f = i - 6 * t;
        fo[i] = f < 3 ? (unsigned)(3 * (base + t) + f) : s_rgb[3 * t + f - 3];
    }
}

}  // namespace arvx
