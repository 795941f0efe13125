// Stands where the reference's src/Postprocessing3d.h stood: int applyClosure(Model*, int)
// (src/Postprocessing3d.h:10) over libarvx.so.
#ifndef ARVX_DROPIN_POSTPROCESSING3D_H
#define ARVX_DROPIN_POSTPROCESSING3D_H
#include "Model.h"
#include "arvx/postprocessing.hpp"

using arvx::applyClosure;
#endif
