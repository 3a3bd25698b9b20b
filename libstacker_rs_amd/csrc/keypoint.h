// keypoint.h — ORB / brute-force Hamming / RANSAC-homography side of the engine (keypoint_match path).
#pragma once
#include "common.h"

namespace stk {
struct KeypointWorkspace;
KeypointWorkspace* keypoint_workspace_create();
void keypoint_workspace_destroy(KeypointWorkspace*);
}  // namespace stk
