// mixed.cpp — keypoint_match on a stack whose frames differ in size. The reference reads every file on its own, runs ORB at
// the frame's own size and warps it into the FIRST frame's size (lib.rs:166, 200-204, 290-299: warp_perspective's dsize is
// the first image's). A stack of one geometry — every BASELINE configuration — goes through the batched, lane-parallel
// pipeline of keypoint.cpp; this is the same sequence of stages frame by frame, through the same stage-level entry points
// (stk_grey, stk_orb_detect_and_compute, stk_bf_knn2_hamming, stk_find_homography) and the same fold kernels, for the
// stacks that one cannot take. Each frame's result is what the pipeline computes for it (its stages are batch-invariant;
// tests/test_gpu_mixed.py: a uniform stack through this route equals stk_keypoint_match bit for bit).
// ecc_match on such a stack fails in the reference (cv::add of different sizes, lib.rs:809): STK_BACKEND_ERROR.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "context.h"

namespace {

struct Match { int q, t; float d; };
constexpr int MIXED_MAX_KP = 4096;      // rows of the stage-level ORB call (ORB keeps 500 + ties)

}  // namespace

extern "C" {

stk_status stk_keypoint_match_mixed(stk_ctx* ctx, const stk_frames* frames, const stk_frame_geometry* geometry,
                                    const stk_keypoint_params* params, float scale_down_width, stk_image_f32* out,
                                    int32_t* dropped_out, stk_frame_stats* stats) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (!frames || !frames->data || frames->n < 0) return fail(ctx, STK_INVALID_PARAMS, "null frames");
    if (frames->n == 0) return fail(ctx, STK_NOT_ENOUGH_FILES, "Not enough files");
    const int n = frames->n;
    bool uniform = geometry == nullptr;
    if (geometry) {
        uniform = true;
        for (int i = 0; i < n; i++) {
            if (geometry[i].width <= 0 || geometry[i].height <= 0) return fail(ctx, STK_INVALID_PARAMS, "frame " + std::to_string(i) + ": empty geometry");
            uniform = uniform && geometry[i].width == geometry[0].width && geometry[i].height == geometry[0].height &&
                      geometry[i].row_stride_bytes == geometry[0].row_stride_bytes;
        }
    }
    if (uniform) {
        stk_frames f = *frames;
        if (geometry) { f.width = geometry[0].width; f.height = geometry[0].height; f.row_stride_bytes = geometry[0].row_stride_bytes; }
        return stk_keypoint_match(ctx, &f, params, scale_down_width, out, dropped_out, stats);
    }
    // (a multi-device context takes such a stack on its own device alone, frame by frame: the owning context is member 0)
    if (!params) return fail(ctx, STK_INVALID_PARAMS, "null params");
    if (frames->channels != 3 && frames->channels != 4) return fail(ctx, STK_BACKEND_ERROR, "cvtColor(BGR2GRAY): frames must have 3 or 4 channels (utils.rs:136)");
    const int cn = frames->channels;
    if (frames->depth != 8) return fail(ctx, STK_BACKEND_ERROR, "ORB: only 8-bit images are supported");
    if (params->border_mode < 0 || params->border_mode > 4)
        return fail(ctx, params->border_mode == STK_BORDER_TRANSPARENT ? STK_NOT_IMPLEMENTED : STK_BACKEND_ERROR, "unsupported border mode");
    const int dw = geometry[0].width, dh = geometry[0].height;
    const bool scaled = scale_down_width > 0;
    if (scaled && scale_down_width >= (float)dw)        // lib.rs:377-382: against the first image only
        return fail(ctx, STK_INVALID_PARAMS, "scale_down_to was larger (or equal) to the full image width: full_size:" +
                                              std::to_string(dw) + ", scale_down_to:" + std::to_string(scale_down_width));
    stk_status st;
    if ((st = image_check(ctx, out, dw, dh, cn))) return st;
    if (out->row_stride_bytes) return fail(ctx, STK_INVALID_PARAMS, "output must be tightly packed");
    (void)hipSetDevice(ctx->device);
    timing_begin(ctx);
    const size_t nel = (size_t)dw * dh * cn;
    float* sum = out->data;
    if (out->location != STK_DEVICE) {
        HIP_TRY(ctx->acc.reserve(nel * sizeof(float)));
        sum = ctx->acc.as<float>();
    }
    if (stats) std::memset(stats, 0, sizeof(stk_frame_stats) * (size_t)n);

    // one frame on the device at a time (host frames are uploaded once, into the context's frame buffer)
    size_t max_bytes = 0, max_px = 0;
    for (int i = 0; i < n; i++) {
        const size_t rb = geometry[i].row_stride_bytes ? geometry[i].row_stride_bytes : (size_t)geometry[i].width * cn;
        if (rb < (size_t)geometry[i].width * cn) return fail(ctx, STK_INVALID_PARAMS, "frame " + std::to_string(i) + ": row stride below the row's bytes");
        max_bytes = std::max(max_bytes, rb * (size_t)geometry[i].height);
        max_px = std::max(max_px, (size_t)geometry[i].width * geometry[i].height);
    }
    const bool host = frames->location == STK_HOST;
    if (host) HIP_TRY(ctx->frames.reserve(max_bytes));
    size_t max_small = 0;
    if (scaled)
        for (int i = 0; i < n; i++) {
            int ew, eh;
            if (!stk::scaled_size(geometry[i].width, geometry[i].height, scale_down_width, ew, eh))
                return fail(ctx, STK_INVALID_PARAMS, "frame " + std::to_string(i) + ": scale_down_width gives an empty image");
            max_small = std::max(max_small, (size_t)ew * eh);
        }
    HIP_TRY(ctx->blur_tmp.reserve(max_px + 256 + max_small));      // the frame's grey image (u8), and behind it its scale_image
    uint8_t* grey = ctx->blur_tmp.as<uint8_t>();
    uint8_t* small = grey + ((max_px + 255) & ~(size_t)255);

    std::vector<float> kp0((size_t)MIXED_MAX_KP * 7), kp((size_t)MIXED_MAX_KP * 7);
    std::vector<uint8_t> de0((size_t)MIXED_MAX_KP * 32), de((size_t)MIXED_MAX_KP * 32);
    int n0 = 0, dropped = 0, added = 0;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < n; i++) {
        const int w = geometry[i].width, h = geometry[i].height;
        const size_t rb = geometry[i].row_stride_bytes ? geometry[i].row_stride_bytes : (size_t)w * cn;
        if (!frames->data[i]) return fail(ctx, STK_INVALID_PARAMS, "frame " + std::to_string(i) + ": null data");
        const void* dev = frames->data[i];
        if (host) {
            HIP_TRY(hipMemcpyAsync(ctx->frames.p, frames->data[i], rb * (size_t)h, hipMemcpyHostToDevice, ctx->stream));
            dev = ctx->frames.p;
        }
        stk_frames one{};
        const void* one_ptr = dev;
        one.data = &one_ptr; one.n = 1; one.width = w; one.height = h; one.channels = cn; one.depth = 8;
        one.location = STK_DEVICE; one.row_stride_bytes = rb;
        if ((st = stk_grey(ctx, &one, grey))) return st;                                     // utils.rs:136-142
        float* kps = i == 0 ? kp0.data() : kp.data();
        uint8_t* des = i == 0 ? de0.data() : de.data();
        int nk = 0;
        int ow = w, oh = h;                                  // the size ORB runs at
        const uint8_t* orb_in = grey;
        if (scaled) {                                        // utils::scale_image on this frame's grey (lib.rs:389, 429)
            if ((st = stk_scale_image_grey(ctx, grey, w, h, STK_DEVICE, scale_down_width, small, &ow, &oh))) return st;
            orb_in = small;
        }
        if ((st = stk_orb_detect_and_compute(ctx, orb_in, ow, oh, STK_DEVICE, MIXED_MAX_KP, kps, des, &nk))) return st;   // lib.rs:161-175, 200-204
        double H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        bool ok = true;
        int n_matches = 0, n_inliers = 0;
        if (i == 0) n0 = nk;
        else {
            ok = false;
            std::vector<Match> ms;
            if (n0 > 0) {
                std::vector<int> knn((size_t)n0 * 4);
                if ((st = stk_bf_knn2_hamming(ctx, de0.data(), n0, de.data(), nk, knn.data()))) return st;       // lib.rs:208-219
                for (int q = 0; q < n0; q++) {
                    if (knn[q * 4] < 0 || knn[q * 4 + 2] < 0) continue;                                          // m.len() == 2
                    const float d0 = (float)knn[q * 4 + 1], d1 = (float)knn[q * 4 + 3];
                    if (d0 < params->match_ratio * d1) ms.push_back({q, knn[q * 4], d0});                        // Lowe ratio lib.rs:224
                }
                std::stable_sort(ms.begin(), ms.end(), [](const Match& a, const Match& b) { return a.d < b.d; });   // lib.rs:233
                const size_t keep = (size_t)std::round((float)ms.size() * params->match_keep_ratio);              // lib.rs:235
                if (keep < ms.size()) ms.resize(keep);
            }
            n_matches = (int)ms.size();
            if (ms.size() >= 5) {                                                                                // lib.rs:240
                std::vector<float> sp(ms.size() * 2), dp(ms.size() * 2);
                for (size_t k = 0; k < ms.size(); k++) {
                    sp[2 * k] = kp0[(size_t)ms[k].q * 7]; sp[2 * k + 1] = kp0[(size_t)ms[k].q * 7 + 1];          // src_pts: frame 0  lib.rs:245-253
                    dp[2 * k] = kp[(size_t)ms[k].t * 7]; dp[2 * k + 1] = kp[(size_t)ms[k].t * 7 + 1];            // dst_pts: frame i  lib.rs:256-264
                }
                int found = 0;
                std::vector<uint8_t> mask(ms.size());
                // find_homography(dst_pts, src_pts): frame i -> frame 0 (lib.rs:267-276)
                const int mth = params->method;
                const bool known = mth == STK_METHOD_LEAST_SQUARES || mth == STK_METHOD_LMEDS || mth == STK_METHOD_RANSAC || mth == STK_METHOD_RHO || (mth >= 32 && mth <= 38);
                // (an unknown method: findHomography throws and the frame is skipped, lib.rs:275; RHO / USAC: STK_NOT_IMPLEMENTED from the stage)
                if (known && (st = stk_find_homography(ctx, dp.data(), sp.data(), (int)ms.size(), mth, params->ransac_reproj_threshold,
                                                       H, mask.data(), &found))) return st;
                if (found) {
                    const double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
                    ok = std::fabs(det) >= 1e-6;                                                                 // lib.rs:284 / 521 (on the small-image H)
                    if (ok && scaled) {                      // adjust_homography_for_scale_f64(h_small, THIS frame's small grey, THIS frame): utils.rs:229-239
                        const double sx = (double)w / (double)ow, sy = (double)h / (double)oh;
                        H[2] *= sx; H[5] *= sy; H[6] /= sx; H[7] /= sy;
                    }
                    for (uint8_t m : mask) n_inliers += m != 0;
                }
            }
        }
        if (stats) {
            stats[i].status = ok ? 0 : 1; stats[i].n_keypoints = nk; stats[i].n_matches = n_matches; stats[i].n_inliers = ok ? n_inliers : 0;
            for (int k = 0; k < 9; k++) stats[i].warp[k] = ok ? H[k] : I3[k];
        }
        if (!ok) { dropped++; continue; }
        // warp_perspective(frame i, H, dsize = the first frame's) + add (lib.rs:290-316); frame 0 under the identity (lib.rs:194-196)
        std::vector<WarpFrame> wf(1);
        make_warp_frame(wf[0], dev, i == 0 ? I3 : H, 0);
        if ((st = warp_fold(ctx, wf, 8, w, h, cn, rb, 1.0 / 255.0, i == 0 ? STK_BORDER_CONSTANT : params->border_mode,
                            i == 0 ? nullptr : params->border_value, 0, sum, (size_t)dw * cn, added > 0 ? 1 : 0, dw, dh))) return st;
        added++;
    }
    if (dropped_out) *dropped_out = dropped;
    if (added <= 0)   // lib.rs:324
        return fail(ctx, STK_INVALID_PARAMS, "All images discarded: try modifying KeyPointMatchParameters::match_distance_threshold");
    stk_image_f32 s = *out;
    s.data = sum; s.location = STK_DEVICE;
    return stk_finalize_mean(ctx, &s, n - dropped, out);            // lib.rs:342: img / (n - dropped)
}

}  // extern "C"
