// Stands where the reference's src/Model.h stood: with this directory
// on the include path and src/Model.{h,cpp} removed, `#include "Model.h"` yields the same global
// names -- Model, DCLR, Vector4f, MODEL_COLOR, UNSEEN_COLOR (src/Model.h:68-91,93) -- over
// libarvx.so.
#ifndef ARVX_DROPIN_MODEL_H
#define ARVX_DROPIN_MODEL_H
#include "arvx/model.hpp"

using Vector4f = arvx::Vec4f;  // Eigen::Vector4f itself where Eigen is installed (vec_types.hpp)
using Model = arvx::Model;
using DCLR = arvx::DCLR;
#ifndef MODEL_COLOR
#define MODEL_COLOR Vector4f(50, 168, 141, 1)
#define UNSEEN_COLOR Vector4f(204, 0, 0, 1)
#endif
#endif
