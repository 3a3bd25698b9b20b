// keypoint.h — ORB / brute-force Hamming / RANSAC-homography side of the engine (keypoint_match path).
#pragma once
#include "common.h"

namespace stk {

constexpr int ORB_LEVELS = 8;
constexpr int ORB_NFEATURES = 500;
constexpr int ORB_EDGE = 31;
constexpr int ORB_FAST_THRESHOLD = 20;
constexpr int ORB_KNN_MAX_TRAIN = 65536; // train rows per set of launch_knn2_hamming (the index shares a 32-bit key with the distance)
constexpr int ORB_SEL_CAP = 4096;        // short-list entries per level (2 n_l + ties)

struct OrbCandidate { int xy; int score; };                               // x | y << 16
struct OrbSelected { int xy; int score; float harris; int m01, m10; int pad; };
struct OrbLevelState { int hist[256]; int n_cand; int threshold; int n_sel; int pad; };
struct OrbUmax { int u[16]; };
struct Gauss7 { float k[7]; };
struct OrbPyramid { int w[ORB_LEVELS], h[ORB_LEVELS]; size_t ofs[ORB_LEVELS]; size_t total; };
// per-level geometry of a pyramid for the all-levels launches: tiles of the FAST kernel (128 x 32 px) in level order
struct OrbLevelTable {
    int w[ORB_LEVELS], h[ORB_LEVELS], tiles_x[ORB_LEVELS], tile_ofs[ORB_LEVELS + 1];
    unsigned pyr_ofs[ORB_LEVELS], cand_ofs[ORB_LEVELS];
    int cand_cap[ORB_LEVELS], keep[ORB_LEVELS];
};
struct OrbFinalKeypoint { int level, cx, cy; float cos_a, sin_a; int frame, row; };   // row: descriptor row to write
// a level's kept corners after the Harris cull (KeyPointsFilter::retainBest: the n_l best and everything that ties with the
// n_l-th), in the order (harris descending, y, x): orb_cull_all_kernel
struct OrbKept { int xy; float harris; int m01, m10; };
constexpr int ORB_KEEP_PACK = 256;       // kept corners per level the device-side cull hands over (n_l <= ~110 + ties; else the host culls that level)
constexpr int ORB_CULL_MAX = 512;        // short-list length the device-side cull sorts (2 n_l + ties; else the host culls that level)

struct KeypointWorkspace;
class HostPool;
void host_pool_destroy(HostPool*);
KeypointWorkspace* keypoint_workspace_create();
void keypoint_workspace_destroy(KeypointWorkspace*);

// The ORB launchers are batched over frames: n_frames images whose per-frame arrays sit `*_stride` elements apart.
// `tab`: the step's coefficient tables (launch_resize_tables: resize_tables_ints(dw, dh) ints, 16-byte aligned) — then `src` needs 11
// readable bytes behind its last row —, or null (tables computed per tile)
hipError_t launch_resize_exact(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, hipStream_t s,
                               int n_frames = 1, size_t frame_stride = 0, const int* tab = nullptr);
hipError_t launch_resize_tables(int sw, int sh, int dw, int dh, int* tab, hipStream_t s);
size_t resize_tables_ints(int dw, int dh);
hipError_t launch_resize_area_u8(const uint8_t* src, int sw, int sh, uint8_t* dst, int dw, int dh, hipStream_t s);
hipError_t launch_resize_area(const void* src, int depth, int sw, int sh, void* dst, int dw, int dh, hipStream_t s);   // depth 8 or 32 (f32 grey)
// scale_image's target size (utils.rs:186-214): the SMALLER dimension becomes scale_down, `as i32` truncation
inline bool scaled_size(int w, int h, float scale_down, int& nw, int& nh) {
    const double sf = w < h ? (double)scale_down / (double)w : (double)scale_down / (double)h;
    nw = (int)((double)w * sf); nh = (int)((double)h * sf);
    // (may EXCEED the input: the smaller dimension becomes scale_down, and the callers' only check is scale_down < WIDTH
    // (lib.rs:377, 876) — a landscape frame with height < scale_down < width is enlarged, resize(INTER_AREA) then runs its
    // bilinear emulation: resize_area_up_kernel)
    return nw > 0 && nh > 0 && (int64_t)nw * nh < ((int64_t)1 << 31);
}
hipError_t launch_fast_level(const uint8_t* img, int w, int h, int thr, int edge, int keep, uint8_t* score,
                             OrbLevelState* st, OrbCandidate* cand, int cap, OrbSelected* sel, int sel_cap,
                             const OrbUmax& um, hipStream_t s, int n_frames = 1, size_t pyr_stride = 0,
                             size_t states_stride = 0, size_t cand_stride = 0, size_t sel_stride = 0);
// kept / kept_cnt (null: no device-side cull): per (frame, level) ORB_KEEP_PACK entries and a count (-1: that level is left to the host)
hipError_t launch_fast_all(const uint8_t* pyr, const OrbLevelTable& L, int thr, int edge, OrbLevelState* st, OrbCandidate* cand,
                           OrbSelected* sel, int sel_cap, const OrbUmax& um, hipStream_t s, int n_frames, size_t pyr_stride,
                           size_t states_stride, size_t cand_stride, size_t sel_stride, OrbKept* kept = nullptr, int* kept_cnt = nullptr);
hipError_t launch_gauss7(const uint8_t* src, int w, int h, const Gauss7& k, float* tmp, uint8_t* dst, hipStream_t s,
                         int n_frames = 1, size_t pyr_stride = 0, size_t tmp_stride = 0);
hipError_t upload_orb_pattern(const signed char* p);
hipError_t launch_brief(const uint8_t* pyr_blur, const OrbPyramid& pyr, const OrbFinalKeypoint* kps, int n, uint8_t* desc,
                        hipStream_t s, size_t pyr_stride = 0);
// blur (7x7, sigma 2) of the window a keypoint's descriptor reads + rotated BRIEF, from the UNBLURRED pyramid: one wave per keypoint
hipError_t launch_brief_patch(const uint8_t* pyr_img, const OrbPyramid& pyr, const OrbFinalKeypoint* kps, int n, const Gauss7& k,
                              uint8_t* desc, hipStream_t s, size_t pyr_stride);
// n_sets train sets against one query set; set k = rows [k * train_stride, k * train_stride + train_counts[k])
hipError_t launch_knn2_hamming(const uint8_t* query, int nq, const uint8_t* train, int nt, int* out, hipStream_t s,
                               int n_sets = 1, const int* train_counts = nullptr, size_t train_stride = 0);

}  // namespace stk
