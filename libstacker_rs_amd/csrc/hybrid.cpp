// hybrid.cpp — BASELINE.json configs[4]: "16-bit stack, mixed ORB-seeded ECC refine". An EXTENSION beyond the reference
// (both of its paths reject 16-bit input); defined in SURVEY §8d as:
//   grey16 = BGR2GRAY (16U formula)            ORB on grey8 = (grey16 + 128) / 257
//   RANSAC homography (f64, frame i -> frame 0, as keypoint_match computes it) normalised to h22 = 1, cast to f32
//   = the INITIAL warp of findTransformECC (OpenCV supports a non-identity start) on float(grey16), 32FC1 branch
//   fold with convertTo alpha = 1/65535 (declared deviation from the literal 1/255 of utils.rs:133)
// 8-bit stacks take the same route with grey8 = grey and alpha = 1/255. A frame whose homography cannot be estimated is
// not dropped: its ECC simply starts from the identity, as ecc_match would. ECC failures abort like ecc_match.
#include <cmath>
#include <cstring>

#include "context.h"

extern "C" {

stk_status stk_hybrid_match_shard(stk_ctx* ctx, const stk_frames* frames, const stk_keypoint_params* kp_params,
                                  const stk_ecc_params* ecc_params, int32_t add_reference, stk_image_f32* sum,
                                  int32_t* n_added, stk_frame_stats* stats) {
    stk_status st = check_frames(ctx, frames, true);
    if (st) return st;
    if (!kp_params || !ecc_params) return fail(ctx, STK_INVALID_PARAMS, "null params");
    if (frames->depth != 8 && frames->depth != 16) return fail(ctx, STK_INVALID_PARAMS, "hybrid match takes 8- or 16-bit BGR frames");
    if (frames->channels != 3) return fail(ctx, STK_INVALID_PARAMS, "hybrid match takes 3-channel frames");
    if (ecc_params->motion_type != STK_MOTION_HOMOGRAPHY)
        return fail(ctx, STK_INVALID_PARAMS, "hybrid match seeds a homography: motion type must be Homography");
    const int n = frames->n;
    timing_begin(ctx);
    std::vector<KpAlign> al;
    std::vector<const void*> dev;
    int n0 = 0;
    if ((st = keypoint_align_impl(ctx, frames, kp_params, 0.f, true, al, &n0, dev))) return st;
    // seeds: H / h22 in f32, identity where no homography was found
    std::vector<float> seeds((size_t)n * 9, 0.f);
    for (int i = 0; i < n; i++) {
        float* sd = seeds.data() + (size_t)i * 9;
        sd[0] = sd[4] = sd[8] = 1.f;
        if (i > 0 && al[i].ok && std::fabs(al[i].H[8]) > 1e-12) {
            for (int k = 0; k < 9; k++) sd[k] = (float)(al[i].H[k] / al[i].H[8]);
            sd[8] = 1.f;
        }
    }
    // the frames already sit on the device (uploaded once by the alignment step if they came from the host)
    stk_frames devf = *frames;
    std::vector<void*> devp(dev.size());
    for (size_t i = 0; i < dev.size(); i++) devp[i] = const_cast<void*>(dev[i]);
    devf.data = devp.data();
    devf.location = STK_DEVICE;
    const double alpha = frames->depth == 16 ? 1.0 / 65535.0 : 1.0 / 255.0;
    const stk_timing kp_t = ctx->timing;                      // upload + ORB figures of the alignment step above
    st = ecc_shard_impl(ctx, &devf, ecc_params, 0.f, add_reference, sum, n_added, stats, seeds.data(), alpha, true);
    if (st) return st;
    ctx->timing.h2d_ms = kp_t.h2d_ms; ctx->timing.h2d_bytes = kp_t.h2d_bytes;
    ctx->timing.fast_ms = kp_t.fast_ms; ctx->timing.fast_launches = kp_t.fast_launches; ctx->timing.fast_pixels = kp_t.fast_pixels;
    if (stats)
        for (int i = 0; i < n; i++) { stats[i].n_keypoints = i ? al[i].n_keypoints : n0; stats[i].n_matches = al[i].n_matches; stats[i].n_inliers = al[i].n_inliers; }
    return STK_OK;
}

stk_status stk_hybrid_match(stk_ctx* ctx, const stk_frames* frames, const stk_keypoint_params* kp_params,
                            const stk_ecc_params* ecc_params, stk_image_f32* out, stk_frame_stats* stats) {
    if (ctx && ctx->multi) return multi_match(ctx, 2, frames, kp_params, ecc_params, 0.f, out, nullptr, stats);
    stk_status st = check_frames(ctx, frames, true);
    if (st) return st;
    if ((st = image_check(ctx, out, frames->width, frames->height, 3))) return st;
    if (out->row_stride_bytes) return fail(ctx, STK_INVALID_PARAMS, "output must be tightly packed");
    (void)hipSetDevice(ctx->device);
    const size_t nel = (size_t)frames->width * frames->height * 3;
    stk_image_f32 sum = *out;
    if (out->location != STK_DEVICE) {
        HIP_TRY(ctx->acc.reserve(nel * sizeof(float)));
        sum.data = ctx->acc.as<float>(); sum.location = STK_DEVICE;
    }
    int32_t added = 0;
    if ((st = stk_hybrid_match_shard(ctx, frames, kp_params, ecc_params, 1, &sum, &added, stats))) return st;
    return stk_finalize_mean(ctx, &sum, frames->n, out);      // img / n as f64 (lib.rs:836-839)
}

}  // extern "C"
