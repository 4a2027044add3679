// placeholder translation unit (LO-RANSAC triangulation kernels land here)
#include "common.hpp"
