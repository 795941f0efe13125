// NOT OpenCV: see ../core.hpp in this directory.
#include "opencv2/core.hpp"
