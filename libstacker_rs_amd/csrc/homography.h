// homography.h — calib3d::findHomography (lib.rs:267-276) of the keypoint path, batched over the frames of a shard.
//
// Split of the work:
//   host   (homography.cpp)          cv::RNG sample sequence + the sample admissibility test, the sequential
//                                    best-so-far / RANSACUpdateNumIters replay over the scores the device returns
//   device (kernels_homography.hip)  every 4-point model of a round for every frame in ONE launch (one wavefront per
//                                    model: closed-form projective-basis solve in f64, f32 reprojection test, inlier count
//                                    by ballot or least-median by bit-wise bisection), then ONE launch that re-derives the
//                                    winning model's inlier mask, solves the normalised DLT on the inliers and runs the
//                                    Levenberg-Marquardt polish (one wavefront per frame, moment sums by wave reduction)
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

struct stk_ctx;

namespace stk {
namespace geom {

constexpr int HG_MAX_POINTS = 4096;      // correspondences per problem (ORB yields <= 500 + ties; MAX_KP rows per frame)

struct HgPoint { float Mx, My, mx, my; };   // one correspondence M -> m (findHomography's srcPoints -> dstPoints)

struct HgFrame {             // one estimation problem, as the model kernel sees it in one round
    int pt_ofs, n;           // points[pt_ofs .. pt_ofs + n)
    int hyp_ofs, n_hyp;      // samples[hyp_ofs .. hyp_ofs + n_hyp), scores likewise
    int err_ofs;             // LMEDS: first float of this problem's n_hyp x n error scratch
    float thr2;              // RANSAC: (float)(thr * thr)
};
struct HgSample { int idx[4]; };

struct HgJob {               // one problem, as the refinement kernel sees it
    int pt_ofs, n;
    int idx[4];              // winning sample (mode 1)
    float thr2;              // inlier threshold of the winning model (mode 1)
    int mode;                // 0: every point, DLT then LM when n > 4 (method 0, or n == 4)
                             // 1: inliers of the winning sample's model, DLT + LM (RANSAC / LMEDS)
                             // -1: nothing to do (no model found)
};
struct HgResult { double H[9]; int found; int n_inliers; int lm_iterations; int dlt_degenerate; };

hipError_t launch_hg_models(const HgPoint* pts, const HgFrame* frames, int n_frames, int max_hyp, const HgSample* samples,
                            int lmeds, float* err_scratch, int* scores, hipStream_t s);
hipError_t launch_hg_refine(const HgPoint* pts, const HgJob* jobs, int n_frames, HgResult* results, uint8_t* masks,
                            hipStream_t s);

// ---- host driver ----------------------------------------------------------------------------------------------------
struct HgProblem { const float* from_pts; const float* to_pts; int n; uint8_t* mask_or_null; };
struct HgOutcome {
    int rc = 0;              // 0 ok (see found), 3 arguments OpenCV rejects (n < 4, unknown method), 7 method not implemented
    int found = 0;           // 0: OpenCV would return an empty Mat
    double H[9] = {0};
    int n_inliers = 0;
    int models_evaluated = 0;
};
struct HgWorkspace;
HgWorkspace* hg_workspace_create();
void hg_workspace_destroy(HgWorkspace*);
// findHomography(from, to, method, thr, mask) for `count` independent problems. Returns a HIP/driver failure as a status
// (message in ctx); per-problem argument errors are reported in out[i].rc.
int find_homography_batch(stk_ctx* ctx, hipStream_t stream, HgWorkspace* ws, const HgProblem* probs, int count, int method,
                          double thr, HgOutcome* out);

}  // namespace geom
}  // namespace stk
