// Stands where the reference's src/ColorReconstruction.h stood: reconstructClosestColor /
// reconstructAvgColor (src/ColorReconstruction.h:131,142) over libarvx.so.
#ifndef ARVX_DROPIN_COLOR_RECONSTRUCTION_H
#define ARVX_DROPIN_COLOR_RECONSTRUCTION_H
#include "Model.h"
#include "arvx/opencv_dropin.hpp"
#endif
