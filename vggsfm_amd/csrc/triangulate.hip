// LO-RANSAC triangulation kernels (under construction in this commit: entry points report UNSUPPORTED)
#include "common.hpp"
extern "C" {
size_t vgg_triangulate_workspace_bytes(int S, int N, int H, int lo_num) { return 0; }
int vgg_triangulate_tracks(const double*, const double*, const uint8_t*, const int32_t*, int, int, int, int, double,
                           double, double*, int64_t*, uint8_t*, void*, void*) { return VGG_ERR_UNSUPPORTED; }
}
