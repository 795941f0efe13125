// mc_tables.hpp -- marching-cubes case tables.
//
// kTriangles[cubeIndex]: Paul Bourke's triangle table ("Polygonising a scalar field", 1994,
// public domain), one hex digit per edge number, three digits per triangle -- the table the
// reference's Polygonise walks (src/MarchingCubes.h:147-405, :502-509); the data lives in
// mc_triangles.inc (generated and checked by tools/make_mc_tables.py).  Cube index bit i is
// set when corner i is OUTSIDE the model (value < threshold, :479-484); edge e joins corner
// e % 8 and kSecondCorner[e] (:491).  The edge table of the reference (:112-145) is not
// stored: an edge is cut exactly when its two corners differ, edge_mask() below.
#ifndef ARVX_MC_TABLES_HPP
#define ARVX_MC_TABLES_HPP

namespace arvx {
namespace mc {

constexpr int kSecondCorner[12] = {1, 2, 3, 0, 5, 6, 7, 4, 4, 5, 6, 7};

// edgeTable[cubeIndex] of the reference: bit e set when edge e is cut by the surface
constexpr int edge_mask(int cubeIndex) {
    int m = 0;
    for (int e = 0; e < 12; ++e)
        if (((cubeIndex >> (e % 8)) & 1) != ((cubeIndex >> kSecondCorner[e]) & 1)) m |= 1 << e;
    return m;
}

constexpr int hex_digit(char c) { return c <= '9' ? c - '0' : c - 'a' + 10; }

static constexpr const char *kTriangles[256] = {
#include "arvx/mc_triangles.inc"
};

}  // namespace mc
}  // namespace arvx
#endif
