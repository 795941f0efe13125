// Stands where the reference's src/VoxelCarving.h stood: carve / fastCarve with the reference's
// cv::Mat signatures (src/VoxelCarving.h:19,31), defined inline over libarvx.so.  The including
// file must have seen the reference's PoseEstimation.h (estimatePoseFromImage) first, as
// src/main.cpp:5-7 has.
#ifndef ARVX_DROPIN_VOXEL_CARVING_H
#define ARVX_DROPIN_VOXEL_CARVING_H
#include "Model.h"
#include "arvx/opencv_dropin.hpp"
#endif
