// Stands where the reference's src/MarchingCubes.h stood: marchingCubes, SimpleMesh, Triangle,
// ProcessVoxel, Vector3f at global scope (src/MarchingCubes.h:12,19,33,532,596) over libarvx.so.
#ifndef ARVX_DROPIN_MARCHING_CUBES_H
#define ARVX_DROPIN_MARCHING_CUBES_H
#include "Model.h"
#include "arvx/marching_cubes.hpp"

using Vector3f = arvx::Vec3f;  // Eigen::Vector3f itself where Eigen is installed
using arvx::marchingCubes;
using arvx::ProcessVoxel;
using SimpleMesh = arvx::SimpleMesh;
using Triangle = arvx::Triangle;
#endif
