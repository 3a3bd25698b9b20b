// homography.h — host geometry of the keypoint path (see homography.cpp).
#pragma once
#include <cstdint>

namespace stk {
namespace geom {
int find_homography(const float* src_pts, const float* dst_pts, int n, int method, double thr, double* H,
                    uint8_t* mask_out, int* found);
}  // namespace geom
}  // namespace stk
