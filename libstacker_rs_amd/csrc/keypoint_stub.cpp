// temporary: keypoint path entry points (replaced by keypoint.cpp)
#include "keypoint.h"
namespace stk {
struct KeypointWorkspace { int dummy; };
KeypointWorkspace* keypoint_workspace_create() { return new KeypointWorkspace(); }
void keypoint_workspace_destroy(KeypointWorkspace* k) { delete k; }
}
extern "C" {
stk_status stk_keypoint_match(stk_ctx*, const stk_frames*, const stk_keypoint_params*, float, stk_image_f32*, int32_t*, stk_frame_stats*) { return STK_NOT_IMPLEMENTED; }
stk_status stk_keypoint_match_shard(stk_ctx*, const stk_frames*, const stk_keypoint_params*, float, int32_t, stk_image_f32*, int32_t*, int32_t*, stk_frame_stats*) { return STK_NOT_IMPLEMENTED; }
stk_status stk_orb_detect_and_compute(stk_ctx*, const uint8_t*, int32_t, int32_t, int32_t, int32_t, float*, uint8_t*, int32_t*) { return STK_NOT_IMPLEMENTED; }
stk_status stk_bf_knn2_hamming(stk_ctx*, const uint8_t*, int32_t, const uint8_t*, int32_t, int32_t*) { return STK_NOT_IMPLEMENTED; }
stk_status stk_find_homography(stk_ctx*, const float*, const float*, int32_t, int32_t, double, double*, uint8_t*, int32_t*) { return STK_NOT_IMPLEMENTED; }
}
