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
#include "mc_kernels.h"  // kMcTri
#include "state_kernels.h"

namespace arvx {

struct McMeshParams {
    CarveParams g;         // the state records (whole grid, or slab + halo)
    const unsigned long long *paint;  // voxels painted UNSEEN_COLOR by the host (null: none)
    int apply_unseen;      // never-seen voxels are painted UNSEEN_COLOR
    SparseList col;        // colour pass: index, (r, g, b, has-sample flag)
    const float4 *col_rgba;
    SparseList clo;        // closure: index, rgba
    const float4 *clo_rgba;
};

// rgb of an occupied voxel (z: global plane; records, paint plane and lists are indexed by the
// context's local planes)
__device__ inline float3 mc_voxel_rgb(const McMeshParams &p, int x, int y, int zg) {
    const int X = p.g.X, Y = p.g.Y, XW = (X + 63) >> 6;
    const int z = zg - p.g.zoff;
    if (plane_bit(p.paint, X, Y, x, y, z) || (p.apply_unseen && !rec_seen(p.g, x, y, z)))
        return make_float3(204.f, 0.f, 0.f);
    const size_t row = (size_t)z * Y + y;
    int k = sparse_find(p.clo, XW, x, row);
    if (k >= 0) return make_float3(p.clo_rgba[k].x, p.clo_rgba[k].y, p.clo_rgba[k].z);
    k = sparse_find(p.col, XW, x, row);
    if (k >= 0) {
        const float4 c = p.col_rgba[k];
        if (c.w != 0.f) return make_float3(c.x, c.y, c.z);
    }
    return make_float3(50.f, 168.f, 141.f);
}

__device__ __forceinline__ unsigned mc_mean3(float a, float b, float c) {
    const float s = (a + b) + c;
    return (unsigned)roundf(s / 3.f);
}

constexpr int kMeshMaxTris = 5;  // Bourke's table: at most five triangles per cell

// The triangles of the 64 consecutive cells of a wave are one contiguous range of the output
// (tri_offset is the exclusive scan of their counts).  Three steps per wave, through LDS:
//   A  lane = cell: the cell's triangles as packed descriptors (corner of the cell, nine corner
//      offset bits) -- no memory traffic beyond the table row, read as ONE 16-byte load;
//   B  lane = triangle: the two voxel-colour lookups and the face colour.  Cells have one to
//      five triangles; per triangle every lane is busy, per cell the wave would wait for its
//      slowest lane;
//   C  the range streams out in whole lines: 36 bytes of vertices and 24 bytes of face record
//      per triangle leave as coalesced dword stores.
//   verts  9 floats per triangle (three fresh vertices, voxel units)
//   faces  6 uints per triangle: vertex numbers 3t, 3t+1, 3t+2, then r, g, b -- the layout of
//          the C++ layer's Triangle (include/arvx/marching_cubes.hpp), reference
//          src/MarchingCubes.h:19-31
// Only the first two corners' colours are ever used (col[2] = col[1], :506).
// n_cap: room in `cells`; n_dev: the cell list's length on the device; tri_cap: room for triangles
__global__ __launch_bounds__(256) void mc_mesh_kernel(const McMeshParams p,
                                                      const int4 *__restrict__ cells, long long n_cap,
                                                      const long long *__restrict__ n_dev, long long tri_cap,
                                                      const int *__restrict__ tri_offset,
                                                      float *__restrict__ verts,
                                                      unsigned *__restrict__ faces) {
    // per wave and triangle: {x + 1 | (y + 1) << 16, z + 1 | offset bits << 16, r, g, b}
    __shared__ uint32_t s_tri[4][64 * kMeshMaxTris][5];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n = *n_dev < n_cap ? *n_dev : n_cap;
    const long long c0 = ((long long)blockIdx.x * 4 + wave) * 64;
    if (c0 >= n) return;  // (no workgroup-wide barrier below: a wave only touches its own part)
    const long long c = c0 + lane;
    const long long clast = (c0 + 63 < n - 1) ? c0 + 63 : n - 1;
    const long long base = tri_offset[c0];
    const int ntri = (int)(tri_offset[clast] - base) + kMcTri.n[cells[clast].w & 255];
    uint32_t(*mine)[5] = s_tri[wave];
    // corner i of ProcessVoxel (src/MarchingCubes.h:537-552); bit i of idx set = NOT in the model
    // x offsets of corners 0..7: 1 0 0 1 1 0 0 1; y: 0 0 1 1 0 0 1 1; z: 0 0 0 0 1 1 1 1
    constexpr unsigned kCx = 0x99u, kCy = 0xCCu, kCz = 0xF0u;
    if (c < n) {  // ---- A
        const int4 cell = cells[c];
        const int idx = cell.w & 255;
        const int t0 = (int)(tri_offset[c] - base), nt = kMcTri.n[idx];
        const uint4 rw = *reinterpret_cast<const uint4 *>(kMcTri.e[idx]);  // 16 edge numbers
        const unsigned long long elo = (unsigned long long)rw.x | ((unsigned long long)rw.y << 32),
                                 ehi = (unsigned long long)rw.z | ((unsigned long long)rw.w << 32);
        // (cell coordinates start at -1: stored + 1; grids are at most 16384 wide)
        const uint32_t w0 = (uint32_t)(cell.x + 1) | ((uint32_t)(cell.y + 1) << 16);
        for (int k = 0; k < nt; ++k) {
            unsigned bits = 0;  // bit 3 v + axis: offset of vertex v along the axis
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                const int j = 3 * k + v;
                const int e = (int)(((j < 8 ? elo >> (8 * j) : ehi >> (8 * (j - 8)))) & 15ull);
                const int a = e & 7, b = mc::kSecondCorner[e];  // e % 8 and its partner (:491)
                const int cn = ((idx >> a) & 1) ? b : a;         // the corner that is in the model
                bits |= (((kCx >> cn) & 1u) | (((kCy >> cn) & 1u) << 1) | (((kCz >> cn) & 1u) << 2))
                        << (3 * v);
            }
            mine[t0 + k][0] = w0;
            mine[t0 + k][1] = (uint32_t)(cell.z + 1) | (bits << 16);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the wave's LDS writes have landed
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < ntri; t += 64) {  // ---- B
        const uint32_t w0 = mine[t][0], w1 = mine[t][1];
        const int x = (int)(w0 & 0xffffu) - 1, y = (int)(w0 >> 16) - 1, z = (int)(w1 & 0xffffu) - 1;
        const unsigned bits = w1 >> 16;
        const float3 q0 = mc_voxel_rgb(p, x + (int)(bits & 1u), y + (int)((bits >> 1) & 1u),
                                       z + (int)((bits >> 2) & 1u));
        const float3 q1 = mc_voxel_rgb(p, x + (int)((bits >> 3) & 1u), y + (int)((bits >> 4) & 1u),
                                       z + (int)((bits >> 5) & 1u));  // col[2] = col[1], :506
        mine[t][2] = mc_mean3(q0.x, q1.x, q1.x);
        mine[t][3] = mc_mean3(q0.y, q1.y, q1.y);
        mine[t][4] = mc_mean3(q0.z, q1.z, q1.z);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float *vo = verts + 9 * base;  // ---- C
    for (int i = lane; i < 9 * ntri; i += 64) {
        const int t = i / 9, comp = i - 9 * t, axis = comp % 3;
        if (base + t >= tri_cap) continue;  // (no room: the caller repeats the launch with more)
        const uint32_t w0 = mine[t][0], w1 = mine[t][1];
        const int origin = axis == 0 ? (int)(w0 & 0xffffu) : axis == 1 ? (int)(w0 >> 16)
                                                                        : (int)(w1 & 0xffffu);
        vo[i] = (float)(origin - 1 + (int)((w1 >> (16 + comp)) & 1u));
    }
    unsigned *fo = faces + 6 * base;
    for (int i = lane; i < 6 * ntri; i += 64) {
        const int t = i / 6, f = i - 6 * t;
        if (base + t >= tri_cap) continue;
        fo[i] = f < 3 ? (unsigned)(3 * (base + t) + f) : mine[t][f - 1];
    }
}

}  // namespace arvx
