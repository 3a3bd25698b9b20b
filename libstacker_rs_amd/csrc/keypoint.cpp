// keypoint.cpp — keypoint_match (lib.rs:146-353) on the GPU engine: per frame ORB on device
// (kernels_orb.hip) -> brute-force Hamming 2-NN on device -> Lowe ratio / stable sort / truncate and
// RANSAC homography on the host (a few hundred points, homography.cpp) -> one fused
// warpPerspective + accumulate launch over all kept frames (kernels_warp.hip).
//
// Drop semantics: the reference's `return Ok(None)` inside the Rayon fold wipes the segment
// accumulator and never counts the drop, which makes its output schedule-dependent.
// This implements the DOCUMENTED behaviour (lib.rs:98): a frame whose homography cannot be estimated
// (<5 matches lib.rs:240, find_homography failure lib.rs:275, bad shape lib.rs:279, |det| < 1e-6
// lib.rs:284) is skipped and counted; the sum is divided by n - dropped.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>

#include "context.h"
#include "homography.h"
#include "host_pool.h"
#include "upload.h"
#include "orb_pattern.h"

using namespace stk;

// HIP_TRY against an explicit context (the lanes of keypoint_align_impl run the same code on two contexts)
#define HIP_TRY_C(c, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail((c), STK_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

namespace stk {

struct KeypointWorkspace {
    DevBuf pyr, score, blur, tmpf, cand, sel, states, final_kps, desc0, desc, knn, gfull, counts, rtab, kept, kept_cnt;
    bool pattern_uploaded = false;
    int rtab_w = 0, rtab_h = 0;             // level-0 size the resize tables in `rtab` were computed for
    size_t rtab_ofs[ORB_LEVELS] = {};       // ints into rtab: tables of the step level l - 1 -> l
    OrbSelected* host_sel = nullptr;        // pinned: [frames][ORB_LEVELS][ORB_PACK] head of every short list
    OrbLevelState* host_states = nullptr;   // pinned: [frames][ORB_LEVELS]
    OrbKept* host_kept = nullptr;           // pinned: [frames][ORB_LEVELS][ORB_KEEP_PACK] what the device-side cull kept, in order
    int* host_kept_cnt = nullptr;           // pinned: [frames][ORB_LEVELS] (-1: that level's short list is culled here)
    OrbFinalKeypoint* host_final = nullptr; // pinned: the kept keypoints of a batch, back to back (read by an async copy)
    size_t host_final_cap = 0;
    int* host_knn = nullptr;                // pinned
    size_t host_knn_cap = 0, host_frames_cap = 0;
};

KeypointWorkspace* keypoint_workspace_create() { return new KeypointWorkspace(); }
void keypoint_workspace_destroy(KeypointWorkspace* k) {
    if (!k) return;
    for (DevBuf* b : {&k->pyr, &k->score, &k->blur, &k->tmpf, &k->cand, &k->sel, &k->states, &k->final_kps, &k->desc0, &k->desc, &k->knn, &k->gfull, &k->counts, &k->rtab, &k->kept, &k->kept_cnt})
        b->release();
    if (k->host_sel) (void)hipHostFree(k->host_sel);
    if (k->host_states) (void)hipHostFree(k->host_states);
    if (k->host_kept) (void)hipHostFree(k->host_kept);
    if (k->host_kept_cnt) (void)hipHostFree(k->host_kept_cnt);
    if (k->host_knn) (void)hipHostFree(k->host_knn);
    if (k->host_final) (void)hipHostFree(k->host_final);
    delete k;
}

}  // namespace stk

namespace {

inline int cv_round_f(float v) { return (int)std::lrintf(v); }
inline int cv_round_d(double v) { return (int)std::lrint(v); }
inline int cv_floor_f(float v) { int i = (int)v; return i - (i > v); }

// cv::fastAtan2 (degrees)
float fast_atan2(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
    const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
    const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
    const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
    const float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

struct HostKeypoint { float x, y, size, angle, response; int octave; int lx, ly; };

struct OrbGeometry {
    OrbPyramid pyr;
    float scale[ORB_LEVELS];
    int nfeatures[ORB_LEVELS];
    OrbUmax umax;
    Gauss7 g7;
    size_t cand_ofs[ORB_LEVELS], cand_cap[ORB_LEVELS], cand_total;
};

void orb_geometry(int w, int h, OrbGeometry& g) {
    size_t ofs = 0, cofs = 0;
    for (int l = 0; l < ORB_LEVELS; l++) {
        const float scale = (float)std::pow((double)1.2f, (double)l);      // getScale(level, 0, 1.2f)
        const float inv = 1.0f / scale;
        g.scale[l] = scale;
        g.pyr.w[l] = cv_round_f((float)w * inv); g.pyr.h[l] = cv_round_f((float)h * inv);
        g.pyr.ofs[l] = ofs;
        ofs += ((size_t)g.pyr.w[l] * g.pyr.h[l] + 255) & ~(size_t)255;
        g.cand_ofs[l] = cofs;
        g.cand_cap[l] = (size_t)g.pyr.w[l] * g.pyr.h[l] / 4 + 64;            // strict 3x3 maxima: at most one per 2x2
        cofs += g.cand_cap[l];
    }
    g.pyr.total = ofs; g.cand_total = cofs;
    const float factor = (float)(1.0 / (double)1.2f);
    float nd = ORB_NFEATURES * (1 - factor) / (1 - (float)std::pow((double)factor, (double)ORB_LEVELS));
    int sum = 0;
    for (int l = 0; l < ORB_LEVELS - 1; l++) { g.nfeatures[l] = cv_round_f(nd); sum += g.nfeatures[l]; nd *= factor; }
    g.nfeatures[ORB_LEVELS - 1] = std::max(ORB_NFEATURES - sum, 0);
    const int half = 15;
    int um[17] = {0};
    const int vmax = cv_floor_f(half * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(half * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; v++) um[v] = cv_round_d(std::sqrt((double)half * half - v * v));
    for (int v = half, v0 = 0; v >= vmin; --v) { while (um[v0] == um[v0 + 1]) ++v0; um[v] = v0; ++v0; }
    for (int v = 0; v < 16; v++) g.umax.u[v] = um[v];
    double kd[7], ks = 0;
    for (int i = 0; i < 7; i++) { const double x = i - 3; kd[i] = std::exp(-0.5 * x * x / 4.0); ks += kd[i]; }
    for (int i = 0; i < 7; i++) g.g7.k[i] = (float)(kd[i] * (1.0 / ks));
}

constexpr size_t MAX_KP = 4096;   // descriptor rows per frame (500 + ties)
constexpr int ORB_PACK = 512;     // short-list entries per level fetched in the one strided copy (2 n_l + ties fit; else a 2nd copy)

}  // namespace

namespace stk {

void host_pool_destroy(HostPool* p) { delete p; }

}  // namespace stk

namespace {

// STK_KP_TRACE=1 in the environment: a line on stderr at every host-visible stage boundary of the keypoint path (context,
// microseconds within the current second, stage) — how the lanes' host steps and the stream interleave
static const bool KP_TRACE = getenv("STK_KP_TRACE") != nullptr;
static double kp_now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define KPT(c, what) do { if (KP_TRACE) fprintf(stderr, "[kp %p] %10.1f %s\n", (void*)(c), kp_now() - 1e6 * std::floor(kp_now() / 1e6), what); } while (0)

// run `fn(i)` for i in [0, n) on the context's host pool (pure host work: no HIP calls inside)
template <typename F>
void parallel_for(stk_ctx* ctx, int n, int threads, F fn) {
    if (threads <= 1 || n <= 1) { for (int i = 0; i < n; i++) fn(i); return; }
    if (ctx->shared_pool) {                                  // a member of a multi-device context: one pool for all members
        const std::function<void(int)> f = fn;
        ctx->shared_pool->run(n, f);
        return;
    }
    if (!ctx->host_pool || ctx->host_pool->size() != threads - 1) {
        host_pool_destroy(ctx->host_pool);
        ctx->host_pool = new HostPool(threads - 1);          // the caller is the remaining thread
    }
    const std::function<void(int)> f = fn;
    ctx->host_pool->run(n, f);
}

// ORB on `n_frames` 8-bit grey images that already sit in level 0 of the workspace pyramids (frame f at
// pyr + f * g.pyr.total). Every device stage is ONE launch per level for all frames; the host steps in between
// (Harris cull, ordering, angles) run on `threads` host threads. Descriptors are left on the device in `desc_dev`,
// frame f in rows [f * MAX_KP, f * MAX_KP + out[f].size()) — by work still QUEUED on `s` when this returns.
stk_status orb_run(stk_ctx* ctx, KeypointWorkspace* ws, hipStream_t s, const OrbGeometry& g, int n_frames, int threads,
                   uint8_t* desc_dev, std::vector<std::vector<HostKeypoint>>& out, hipStream_t tail = nullptr) {
    uint8_t* pyr = ws->pyr.as<uint8_t>();
    uint8_t* score = ws->score.as<uint8_t>();
    OrbLevelState* st = ws->states.as<OrbLevelState>();
    const size_t PT = g.pyr.total, SELF = (size_t)ORB_SEL_CAP * ORB_LEVELS;
    const size_t tmp_stride = (size_t)g.pyr.w[0] * g.pyr.h[0];
    HIP_TRY(hipMemsetAsync(st, 0, sizeof(OrbLevelState) * ORB_LEVELS * n_frames, s));
    for (int l = 1; l < ORB_LEVELS; l++)
        HIP_TRY(launch_resize_exact(pyr + g.pyr.ofs[l - 1], g.pyr.w[l - 1], g.pyr.h[l - 1], pyr + g.pyr.ofs[l], g.pyr.w[l], g.pyr.h[l], s,
                                    n_frames, PT, ctx->opt_orb_resize_tables ? ws->rtab.as<int>() + ws->rtab_ofs[l] : nullptr));
    const bool timed = ctx->opt_profile >= 1;
    if (timed) HIP_TRY(hipEventRecord(ctx->ev[6], s));
    // FAST + NMS + short list: all levels in four launches when every level qualifies for the tiled kernel, else level by level
    OrbLevelTable L{};
    bool all_ok = g.pyr.total < ((size_t)1 << 32) && g.cand_total < ((size_t)1 << 32);
    L.tile_ofs[0] = 0;
    for (int l = 0; l < ORB_LEVELS; l++) {
        const int lw = g.pyr.w[l], lh = g.pyr.h[l];
        L.w[l] = lw; L.h[l] = lh; L.pyr_ofs[l] = (unsigned)g.pyr.ofs[l]; L.cand_ofs[l] = (unsigned)g.cand_ofs[l];
        L.cand_cap[l] = (int)g.cand_cap[l]; L.keep[l] = 2 * g.nfeatures[l];
        L.tiles_x[l] = std::max(1, (lw + 127) / 128);
        const bool has_interior = lw > 2 * ORB_EDGE && lh > 2 * ORB_EDGE;      // otherwise runByImageBorder leaves nothing
        L.tile_ofs[l + 1] = L.tile_ofs[l] + (has_interior ? L.tiles_x[l] * ((lh + 31) / 32) : 0);
        if (lw <= 6 || lh <= 6) all_ok = false;
    }
    hipError_t fe = hipErrorNotSupported;
    if (all_ok) {
        fe = launch_fast_all(pyr, L, ORB_FAST_THRESHOLD, ORB_EDGE, st, ws->cand.as<OrbCandidate>(), ws->sel.as<OrbSelected>(), ORB_SEL_CAP,
                             g.umax, s, n_frames, PT, ORB_LEVELS, g.cand_total, SELF,
                             ctx->opt_orb_device_cull ? ws->kept.as<OrbKept>() : nullptr, ctx->opt_orb_device_cull ? ws->kept_cnt.as<int>() : nullptr);
        if (fe != hipSuccess && fe != hipErrorNotSupported) return fail(ctx, STK_HIP_ERROR, std::string("FAST: ") + hipGetErrorString(fe));
    }
    for (int l = 0; l < ORB_LEVELS; l++) {
        const int lw = g.pyr.w[l], lh = g.pyr.h[l];
        if (lw <= 6 || lh <= 6) continue;
        ctx->timing.fast_launches += 1;
        ctx->timing.fast_pixels += (int64_t)lw * lh * n_frames;
        if (fe == hipSuccess) continue;
        HIP_TRY(launch_fast_level(pyr + g.pyr.ofs[l], lw, lh, ORB_FAST_THRESHOLD, ORB_EDGE, 2 * g.nfeatures[l], score + g.pyr.ofs[l],
                                  st + l, ws->cand.as<OrbCandidate>() + g.cand_ofs[l], (int)g.cand_cap[l],
                                  ws->sel.as<OrbSelected>() + (size_t)l * ORB_SEL_CAP, ORB_SEL_CAP, g.umax, s,
                                  n_frames, PT, ORB_LEVELS, g.cand_total, SELF));
    }
    if (timed) HIP_TRY(hipEventRecord(ctx->ev[7], s));
    // The short lists go to the host on a SECOND stream (the prep stream, idle on this path; not the copy stream, which
    // may be full of frame uploads), behind an event on the FAST kernels, so that the blur kernels queued next on the
    // compute stream do not wait behind 8 MB of D2H traffic.
    hipStream_t ds = ctx->prep_stream;
    HIP_TRY(hipEventRecord(ctx->gate_ev2, s));
    HIP_TRY(hipStreamWaitEvent(ds, ctx->gate_ev2, 0));
    HIP_TRY(hipMemcpyAsync(ws->host_states, st, sizeof(OrbLevelState) * ORB_LEVELS * n_frames, hipMemcpyDeviceToHost, ds));
    // device-side cull (the all-levels launch took it along): what was kept, in order, and a count per (frame, level) — a third of
    // the bytes of the short lists; else the head of every short list in one strided copy (rows = (frame, level), ORB_PACK
    // of ORB_SEL_CAP entries each)
    const bool device_cull = fe == hipSuccess && ctx->opt_orb_device_cull;
    if (device_cull) {
        HIP_TRY(hipMemcpyAsync(ws->host_kept_cnt, ws->kept_cnt.p, sizeof(int) * ORB_LEVELS * n_frames, hipMemcpyDeviceToHost, ds));
        HIP_TRY(hipMemcpyAsync(ws->host_kept, ws->kept.p, sizeof(OrbKept) * ORB_KEEP_PACK * ORB_LEVELS * n_frames, hipMemcpyDeviceToHost, ds));
    } else
        HIP_TRY(hipMemcpy2DAsync(ws->host_sel, sizeof(OrbSelected) * ORB_PACK, ws->sel.p, sizeof(OrbSelected) * ORB_SEL_CAP,
                                 sizeof(OrbSelected) * ORB_PACK, (size_t)ORB_LEVELS * n_frames, hipMemcpyDeviceToHost, ds));
    HIP_TRY(hipEventRecord(ctx->gate_ev, ds));
    // The 7x7 blur of every level (the descriptor stage's input) does not depend on the host's Harris cull: it is queued
    // now and runs while the host works on the short lists (the host waits for the copies above only, not for the stream).
    // (orb_patch_blur = 0 only: by default the descriptor kernel blurs the 45 x 40 window it reads, launch_brief_patch)
    const bool patch_blur = ctx->opt_orb_patch_blur;
    if (!patch_blur) {
        HIP_TRY(ws->blur.reserve(g.pyr.total * (size_t)n_frames + 64));
        HIP_TRY(ws->tmpf.reserve(tmp_stride * sizeof(float) * (size_t)n_frames));
        for (int l = 0; l < ORB_LEVELS; l++)
            HIP_TRY(launch_gauss7(pyr + g.pyr.ofs[l], g.pyr.w[l], g.pyr.h[l], g.g7, ws->tmpf.as<float>(), ws->blur.as<uint8_t>() + g.pyr.ofs[l], s,
                                  n_frames, PT, tmp_stride));
    }
    KPT(ctx, "orb: enqueued A, waiting D2H");
    HIP_TRY(hipEventSynchronize(ctx->gate_ev));
    KPT(ctx, "orb: D2H done");
    if (timed) ctx->timing.fast_ms += ev_ms(ctx->ev[6], ctx->ev[7]);
    // rare: a level with more short-listed corners than ORB_PACK (many tied FAST scores) is fetched whole
    std::vector<std::vector<OrbSelected>> big((size_t)n_frames * ORB_LEVELS);
    for (int f = 0; f < n_frames; f++)
        for (int l = 0; l < ORB_LEVELS; l++) {
            const int n = ws->host_states[f * ORB_LEVELS + l].n_sel;
            if (n > ORB_SEL_CAP) return fail(ctx, STK_PROCESSING_ERROR, "ORB: more tied FAST corners than the short list holds");
            if (device_cull ? ws->host_kept_cnt[f * ORB_LEVELS + l] < 0 : n > ORB_PACK) {
                auto& v = big[(size_t)f * ORB_LEVELS + l];
                v.resize(n);
                HIP_TRY(hipMemcpy(v.data(), ws->sel.as<OrbSelected>() + (size_t)f * SELF + (size_t)l * ORB_SEL_CAP,
                                  sizeof(OrbSelected) * n, hipMemcpyDeviceToHost));
            }
        }

    out.assign(n_frames, {});
    std::vector<std::vector<OrbFinalKeypoint>> fins(n_frames);
    parallel_for(ctx, n_frames, threads, [&](int f) {
        std::vector<HostKeypoint>& o = out[f];
        std::vector<OrbFinalKeypoint>& fin = fins[f];
        size_t n_short = 0;
        for (int l = 0; l < ORB_LEVELS; l++) n_short += (size_t)ws->host_states[f * ORB_LEVELS + l].n_sel;
        o.reserve(n_short); fin.reserve(n_short);            // (the cull keeps about half: one allocation each instead of ten)
        std::vector<OrbSelected> v;
        std::vector<float> resp;
        for (int l = 0; l < ORB_LEVELS; l++) {
            const int n = ws->host_states[f * ORB_LEVELS + l].n_sel;
            const auto& bv = big[(size_t)f * ORB_LEVELS + l];
            const int kc = device_cull ? ws->host_kept_cnt[f * ORB_LEVELS + l] : -1;
            if (kc >= 0) {                                     // culled and ordered on the device
                const OrbKept* kp = ws->host_kept + ((size_t)f * ORB_LEVELS + l) * ORB_KEEP_PACK;
                v.resize(kc);
                for (int i = 0; i < kc; i++) { v[i].xy = kp[i].xy; v[i].score = 0; v[i].harris = kp[i].harris; v[i].m01 = kp[i].m01; v[i].m10 = kp[i].m10; v[i].pad = 0; }
            } else {
                const OrbSelected* src = !bv.empty() ? bv.data() : ws->host_sel + ((size_t)f * ORB_LEVELS + l) * ORB_PACK;
                // cull to n_l by the Harris response (KeyPointsFilter::retainBest: everything >= the n_l-th best stays, ties
                // included), then a deterministic order
                const int keep = g.nfeatures[l];
                float thr = -FLT_MAX;
                if (keep == 0) thr = FLT_MAX;
                else if (keep > 0 && n > keep) {
                    resp.resize(n);
                    for (int i = 0; i < n; i++) resp[i] = src[i].harris;
                    std::nth_element(resp.begin(), resp.begin() + (keep - 1), resp.end(), std::greater<float>());
                    thr = resp[keep - 1];
                }
                v.clear();
                for (int i = 0; i < n; i++) if (keep != 0 && src[i].harris >= thr) v.push_back(src[i]);
                std::sort(v.begin(), v.end(), [](const OrbSelected& p, const OrbSelected& q) {
                    if (p.harris != q.harris) return p.harris > q.harris;
                    const int py = p.xy >> 16, qy = q.xy >> 16;
                    if (py != qy) return py < qy;
                    return (p.xy & 0xffff) < (q.xy & 0xffff);
                });
            }
            const float sc = g.scale[l], inv = 1.f / sc, size = 31 * sc;
            for (const OrbSelected& k : v) {
                HostKeypoint hk;
                hk.lx = k.xy & 0xffff; hk.ly = k.xy >> 16; hk.octave = l;
                hk.response = k.harris;
                hk.angle = fast_atan2((float)k.m01, (float)k.m10);
                hk.size = size;
                hk.x = (float)hk.lx * sc; hk.y = (float)hk.ly * sc;
                o.push_back(hk);
                // computeOrbDescriptors: centre = cvRound(pt * (1/scale)), a = cos(angle deg->rad), b = sin
                float ang = hk.angle;
                ang *= (float)(3.14159265358979323846 / 180.f);
                OrbFinalKeypoint fk;
                fk.level = l; fk.cx = cv_round_f(hk.x * inv); fk.cy = cv_round_f(hk.y * inv);
                fk.cos_a = (float)std::cos(ang); fk.sin_a = (float)std::sin(ang);
                fk.frame = f; fk.row = 0;
                fin.push_back(fk);
            }
        }
        if (o.size() > MAX_KP) { o.resize(MAX_KP); fin.resize(MAX_KP); }
        for (size_t k = 0; k < fin.size(); k++) fin[k].row = (int)((size_t)f * MAX_KP + k);
    });
    KPT(ctx, "orb: cull done");
    // all frames' keypoints back to back in the workspace's pinned buffer: the copy and the descriptor kernel are only
    // ENQUEUED here — the caller synchronises the stream once, behind whatever it queues next (the 2-NN match)
    size_t n_all = 0;
    for (auto& fin : fins) n_all += fin.size();
    if (n_all == 0) return STK_OK;
    if (ws->host_final_cap < n_all) {
        if (ws->host_final) (void)hipHostFree(ws->host_final);
        ws->host_final = nullptr; ws->host_final_cap = 0;
        const size_t cap = std::max(n_all + n_all / 4, (size_t)1024);
        HIP_TRY(hipHostMalloc((void**)&ws->host_final, sizeof(OrbFinalKeypoint) * cap, hipHostMallocDefault));
        ws->host_final_cap = cap;
    }
    n_all = 0;
    for (auto& fin : fins) { std::memcpy(ws->host_final + n_all, fin.data(), sizeof(OrbFinalKeypoint) * fin.size()); n_all += fin.size(); }
    HIP_TRY(ws->final_kps.reserve(sizeof(OrbFinalKeypoint) * n_all));
    // the descriptor stage goes to `tail` when the caller has one (a high-priority stream: everything it depends on has been
    // waited for by the host above); the whole-level blur path stays on `s`, behind its blur launches
    hipStream_t bs = patch_blur && tail ? tail : s;
    HIP_TRY(hipMemcpyAsync(ws->final_kps.p, ws->host_final, sizeof(OrbFinalKeypoint) * n_all, hipMemcpyHostToDevice, bs));
    if (patch_blur) HIP_TRY(launch_brief_patch(pyr, g.pyr, ws->final_kps.as<OrbFinalKeypoint>(), (int)n_all, g.g7, desc_dev, bs, PT));
    else HIP_TRY(launch_brief(ws->blur.as<uint8_t>(), g.pyr, ws->final_kps.as<OrbFinalKeypoint>(), (int)n_all, desc_dev, bs, PT));
    return STK_OK;
}

stk_status orb_prepare(stk_ctx* ctx, KeypointWorkspace* ws, int w, int h, OrbGeometry& g, int n_frames = 1) {
    orb_geometry(w, h, g);
    const size_t F = (size_t)std::max(n_frames, 1);
    HIP_TRY(ws->pyr.reserve(g.pyr.total * F + 64));       // + slack: the tiled kernels read whole aligned dwords
    HIP_TRY(ws->score.reserve(g.pyr.total * F + 64));
    HIP_TRY(ws->cand.reserve(g.cand_total * sizeof(OrbCandidate) * F));
    HIP_TRY(ws->sel.reserve(sizeof(OrbSelected) * ORB_SEL_CAP * ORB_LEVELS * F));
    HIP_TRY(ws->states.reserve(sizeof(OrbLevelState) * ORB_LEVELS * F));
    HIP_TRY(ws->kept.reserve(sizeof(OrbKept) * ORB_KEEP_PACK * ORB_LEVELS * F));
    HIP_TRY(ws->kept_cnt.reserve(sizeof(int) * ORB_LEVELS * F));
    if (ws->host_frames_cap < F) {
        if (ws->host_sel) (void)hipHostFree(ws->host_sel);
        if (ws->host_states) (void)hipHostFree(ws->host_states);
        if (ws->host_kept) (void)hipHostFree(ws->host_kept);
        if (ws->host_kept_cnt) (void)hipHostFree(ws->host_kept_cnt);
        ws->host_sel = nullptr; ws->host_states = nullptr; ws->host_kept = nullptr; ws->host_kept_cnt = nullptr; ws->host_frames_cap = 0;
        HIP_TRY(hipHostMalloc((void**)&ws->host_kept, sizeof(OrbKept) * ORB_KEEP_PACK * ORB_LEVELS * F, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&ws->host_kept_cnt, sizeof(int) * ORB_LEVELS * F, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&ws->host_sel, sizeof(OrbSelected) * ORB_PACK * ORB_LEVELS * F, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc((void**)&ws->host_states, sizeof(OrbLevelState) * ORB_LEVELS * F, hipHostMallocDefault));
        ws->host_frames_cap = F;
    }
    // coefficient tables of the seven pyramid steps: a function of the level sizes only
    if (ws->rtab_w != w || ws->rtab_h != h) {
        size_t n_int = 0;
        for (int l = 1; l < ORB_LEVELS; l++) { ws->rtab_ofs[l] = n_int; n_int += resize_tables_ints(g.pyr.w[l], g.pyr.h[l]); }
        HIP_TRY(ws->rtab.reserve(sizeof(int) * n_int));
        for (int l = 1; l < ORB_LEVELS; l++)
            HIP_TRY(launch_resize_tables(g.pyr.w[l - 1], g.pyr.h[l - 1], g.pyr.w[l], g.pyr.h[l], ws->rtab.as<int>() + ws->rtab_ofs[l], ctx->stream));
        ws->rtab_w = w; ws->rtab_h = h;
    }
    // the BRIEF pattern lives in __constant__ memory of the module: uploaded once
    if (!ws->pattern_uploaded) { HIP_TRY(upload_orb_pattern(ORB_BIT_PATTERN_31)); ws->pattern_uploaded = true; }
    return STK_OK;
}

// device bytes one frame needs in the batched ORB workspace (pyramid x3, f32 scratch, candidates, short lists, descriptors)
size_t orb_bytes_per_frame(const OrbGeometry& g) {
    return g.pyr.total * 2 + g.cand_total * sizeof(OrbCandidate) +
           sizeof(OrbSelected) * ORB_SEL_CAP * ORB_LEVELS + MAX_KP * 32;
}


struct Match { int q, t; float d; };

}  // namespace

extern "C" {

stk_status stk_orb_detect_and_compute(stk_ctx* ctx, const uint8_t* grey, int32_t width, int32_t height, int32_t location,
                                      int32_t max_keypoints, float* keypoints, uint8_t* descriptors, int32_t* n_keypoints) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (!grey || width <= 0 || height <= 0 || !keypoints || !descriptors || !n_keypoints || max_keypoints <= 0)
        return fail(ctx, STK_INVALID_PARAMS, "bad arguments");
    if (width >= 65536 || height >= 32768) return fail(ctx, STK_INVALID_PARAMS, "image too large for ORB");
    (void)hipSetDevice(ctx->device);
    OrbGeometry g;
    stk_status st = orb_prepare(ctx, ctx->kp, width, height, g);
    if (st) return st;
    KeypointWorkspace* ws = ctx->kp;
    HIP_TRY(ws->desc.reserve(MAX_KP * 32));
    HIP_TRY(hipMemcpyAsync(ws->pyr.p, grey, (size_t)width * height,
                           location == STK_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
    std::vector<std::vector<HostKeypoint>> kpsv;
    if ((st = orb_run(ctx, ws, ctx->stream, g, 1, 1, ws->desc.as<uint8_t>(), kpsv))) return st;
    const std::vector<HostKeypoint>& kps = kpsv[0];
    const int n = (int)std::min<size_t>(kps.size(), (size_t)max_keypoints);
    for (int i = 0; i < n; i++) {
        float* o = keypoints + (size_t)i * 7;
        o[0] = kps[i].x; o[1] = kps[i].y; o[2] = kps[i].size; o[3] = kps[i].angle; o[4] = kps[i].response;
        o[5] = (float)kps[i].octave; o[6] = -1.f;
    }
    if (n > 0) HIP_TRY(hipMemcpyAsync(descriptors, ws->desc.p, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *n_keypoints = n;
    return STK_OK;
}

// scale_image on a grey image of `depth` 8 (8UC1) or 32 (32FC1): the two depths a grey can have where the reference shrinks it
static stk_status scale_image_grey_impl(stk_ctx* ctx, const void* grey, int depth, int32_t width, int32_t height, int32_t location,
                                        float scale_down, void* out, int32_t* new_width, int32_t* new_height) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (!grey || !out || width <= 0 || height <= 0 || !new_width || !new_height) return fail(ctx, STK_INVALID_PARAMS, "bad arguments");
    int nw, nh;
    if (!(scale_down > 0) || !scaled_size(width, height, scale_down, nw, nh))
        return fail(ctx, STK_BACKEND_ERROR, "resize(INTER_AREA): scale_down must give a non-empty image");
    (void)hipSetDevice(ctx->device);
    KeypointWorkspace* ws = ctx->kp;
    const size_t el = (size_t)depth / 8, ib = (size_t)width * height * el, ob = (size_t)nw * nh * el;
    const void* src = grey; void* dst = out;
    if (location == STK_HOST) {
        HIP_TRY(ws->gfull.reserve(ib)); HIP_TRY(ws->score.reserve(ob));
        HIP_TRY(hipMemcpyAsync(ws->gfull.p, grey, ib, hipMemcpyHostToDevice, ctx->stream));
        src = ws->gfull.p; dst = ws->score.p;
    }
    HIP_TRY(launch_resize_area(src, depth, width, height, dst, nw, nh, ctx->stream));
    if (location == STK_HOST) HIP_TRY(hipMemcpyAsync(out, dst, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *new_width = nw; *new_height = nh;
    return STK_OK;
}

stk_status stk_scale_image_grey(stk_ctx* ctx, const uint8_t* grey, int32_t width, int32_t height, int32_t location,
                                float scale_down, uint8_t* out, int32_t* new_width, int32_t* new_height) {
    return scale_image_grey_impl(ctx, grey, 8, width, height, location, scale_down, out, new_width, new_height);
}
stk_status stk_scale_image_grey_f32(stk_ctx* ctx, const float* grey, int32_t width, int32_t height, int32_t location,
                                    float scale_down, float* out, int32_t* new_width, int32_t* new_height) {
    return scale_image_grey_impl(ctx, grey, 32, width, height, location, scale_down, out, new_width, new_height);
}

stk_status stk_bf_knn2_hamming(stk_ctx* ctx, const uint8_t* query, int32_t n_query, const uint8_t* train, int32_t n_train,
                               int32_t* out) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (n_query < 0 || n_train < 0 || (n_query && (!query || !out)) || (n_train && !train)) return fail(ctx, STK_INVALID_PARAMS, "bad arguments");
    if (n_query == 0) return STK_OK;
    if (n_train > ORB_KNN_MAX_TRAIN) return fail(ctx, STK_INVALID_PARAMS, "bf_knn2_hamming: more than 65536 train rows");
    (void)hipSetDevice(ctx->device);
    KeypointWorkspace* ws = ctx->kp;
    HIP_TRY(ws->desc0.reserve((size_t)n_query * 32));
    HIP_TRY(ws->desc.reserve((size_t)std::max(n_train, 1) * 32));
    HIP_TRY(ws->knn.reserve((size_t)n_query * 16));
    HIP_TRY(hipMemcpyAsync(ws->desc0.p, query, (size_t)n_query * 32, hipMemcpyHostToDevice, ctx->stream));
    if (n_train) HIP_TRY(hipMemcpyAsync(ws->desc.p, train, (size_t)n_train * 32, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(launch_knn2_hamming(ws->desc0.as<uint8_t>(), n_query, ws->desc.as<uint8_t>(), n_train, ws->knn.as<int>(), ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, ws->knn.p, (size_t)n_query * 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STK_OK;
}

stk_status stk_find_homography(stk_ctx* ctx, const float* src_pts, const float* dst_pts, int32_t n, int32_t method,
                               double thr, double* H, uint8_t* inlier_mask, int32_t* found) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (!src_pts || !dst_pts || !H || !found) return fail(ctx, STK_INVALID_PARAMS, "bad arguments");
    (void)hipSetDevice(ctx->device);
    *found = 0;
    geom::HgProblem pr{src_pts, dst_pts, n, inlier_mask};
    geom::HgOutcome o;
    const int st = geom::find_homography_batch(ctx, ctx->stream, ctx->hg, &pr, 1, method, thr, &o);
    if (st) return (stk_status)st;
    if (o.rc == 7) return fail(ctx, STK_NOT_IMPLEMENTED, "findHomography: RHO is not implemented");
    if (method >= 32 && method <= 38) return fail(ctx, STK_NOT_IMPLEMENTED, "findHomography: the USAC methods are not implemented");
    if (o.rc != 0) return fail(ctx, STK_BACKEND_ERROR, "findHomography: needs at least 4 point pairs and a known method");
    *found = o.found;
    for (int k = 0; k < 9; k++) H[k] = o.H[k];
    return STK_OK;
}

}  // extern "C"

stk_status keypoint_align_impl(stk_ctx* ctx, const stk_frames* frames, const stk_keypoint_params* params, float scale_down_width,
                               bool reduce16, std::vector<KpAlign>& out, int* n_ref_keypoints, std::vector<const void*>& dev,
                               const KpFramesFinal* on_final) {
    stk_status st = check_frames(ctx, frames, true);
    if (st) return st;
    if (!params) return fail(ctx, STK_INVALID_PARAMS, "null params");
    (void)hipSetDevice(ctx->device);
    if (frames->depth != 8 && !(reduce16 && frames->depth == 16))   // ORB::detectAndCompute asserts an 8-bit image
        return fail(ctx, STK_BACKEND_ERROR, "ORB: only 8-bit images are supported");
    if (reduce16 && frames->depth == 16 && scale_down_width > 0) return fail(ctx, STK_NOT_IMPLEMENTED, "scale_down_width with 16-bit frames");
    // findHomography's methods: 0 / LMEDS / RANSAC here; RHO and OpenCV >= 4.5's USAC family (32 .. 38) exist in the reference's
    // OpenCV and are not restated -> STK_NOT_IMPLEMENTED. Any other value makes findHomography throw ("Unknown estimation method"),
    // which keypoint_match turns into a SKIPPED frame (lib.rs:275: Err(_) => return Ok(None)): every moving frame is dropped then.
    bool drop_all_moving = false;
    if (params->method != STK_METHOD_RANSAC && params->method != STK_METHOD_LEAST_SQUARES && params->method != STK_METHOD_LMEDS) {
        if (params->method == STK_METHOD_RHO)
            return fail(ctx, STK_NOT_IMPLEMENTED, "findHomography: RHO is not implemented");
        if (params->method >= 32 && params->method <= 38)
            return fail(ctx, STK_NOT_IMPLEMENTED, "findHomography: the USAC methods are not implemented");
        drop_all_moving = true;
    }
    if (params->border_mode < 0 || params->border_mode > 4)
        return fail(ctx, params->border_mode == STK_BORDER_TRANSPARENT ? STK_NOT_IMPLEMENTED : STK_BACKEND_ERROR, "unsupported border mode");
    const int w = frames->width, h = frames->height, n = frames->n;
    const int cn = frames->channels;                     // 3, or 4 (BGRA: grey from B, G, R — cvtColor ignores the fourth channel)
    if (w >= 65536 || h >= 32768) return fail(ctx, STK_INVALID_PARAMS, "image too large for ORB");
    // keypoint_match_scale_down (lib.rs:355-601): ORB and the homography on INTER_AREA-shrunk greys
    const bool scaled = scale_down_width > 0;
    int ew = w, eh = h;
    if (scaled) {
        if (scale_down_width >= (float)w)   // lib.rs:377-382
            return fail(ctx, STK_INVALID_PARAMS, "scale_down_to was larger (or equal) to the full image width: full_size:" +
                                                  std::to_string(w) + ", scale_down_to:" + std::to_string(scale_down_width));
        if (!scaled_size(w, h, scale_down_width, ew, eh)) return fail(ctx, STK_INVALID_PARAMS, "scale_down_width gives an empty image");
    }

    // device pointers of the frames, returned to the caller. Host-fed stacks cross PCIe on the copy stream while the ORB
    // batches that have arrived are processed: the ORB batch is then capped to one upload batch (+ the reference frame).
    const size_t rb = frame_row_bytes(frames), fb = rb * (size_t)h;
    const bool host_fed = frames->location == STK_HOST;
    dev.resize(n);
    AsyncUpload up;
    if (host_fed) {
        HIP_TRY(ctx->frames.reserve(fb * (size_t)n));
        for (int i = 0; i < n; i++) dev[i] = ctx->frames.as<uint8_t>() + fb * (size_t)i;
        if ((st = up.start(ctx, frames, ctx->frames.p, fb, ctx->opt_upload_batch))) return st;
    } else {
        for (int i = 0; i < n; i++) dev[i] = frames->data[i];
    }
    OrbGeometry g;
    orb_geometry(ew, eh, g);
    const int threads = ctx->opt_kp_workers;
    const bool depth16 = frames->depth == 16;
    const double fix_sx = (double)w / (double)ew, fix_sy = (double)h / (double)eh;   // adjust_homography_for_scale_f64

    std::vector<KpAlign>& results = out;
    results.assign(n, KpAlign{});

    // LANES (round 3). A third of a keypoint step is host work between device stages (Harris cull and ordering, match
    // filter, RANSAC sampling): device-resident stacks are therefore cut into kp_lanes runs of frames (3; at least 8
    // frames each) that go through the whole pipeline side by side — the calling thread on this context, helper threads on
    // hidden contexts of the same device (own streams, events and workspaces: keypoint.cpp's code runs on them unchanged) —
    // so that one lane's kernels fill the other lanes' host gaps. Every frame is independent of the others (lib.rs:185-290
    // is the body of a Rayon map): the per-frame results do not depend on the lane. The reference frame belongs to lane 0;
    // the other lanes wait for its keypoints (host) and descriptors (an event on lane 0's stream) before their first match.
    // The device takes the lanes' FAST launches (each fills it) more or less one after the other, so the lanes drift apart
    // by themselves; what stays exposed is the LAST lane's host tail (cull -> BRIEF -> 2-NN -> filter -> RANSAC, ~0.75 ms
    // of latency for ~0.15 ms of kernels), which is why more, shorter lanes help up to 4-6 and the fold follows the lanes
    // (`on_final`). 64 x 1080p: one pipeline 3.3-3.4 ms, 2 lanes 3.1-3.2, 4 lanes + following fold 2.8-2.9 ms per stack; at the
    // end of the round, with the tail shortened (device cull, 8-lane 2-NN, priority stream): 2 lanes 2.45, 3 lanes 2.36, 4 lanes 2.42.
    // Cutting a LANE into several ORB batches instead costs more than it hides (3.5-4.7 ms: every batch has its own syncs).
    struct RefShare {
        std::mutex m;
        std::condition_variable cv;
        bool ready = false, failed = false;
        std::vector<HostKeypoint> kp0;
        int n0 = 0;
        const uint8_t* desc0 = nullptr;       // lane 0's device buffer
        hipEvent_t ev = nullptr;              // recorded on lane 0's stream behind the copy into desc0
    } ref;
    HIP_TRY(hipEventCreateWithFlags(&ref.ev, hipEventDisableTiming));
    struct EvGuard { hipEvent_t e; ~EvGuard() { (void)hipEventDestroy(e); } } ref_ev_guard{ref.ev};

    const bool two_lanes_cfg = ctx->opt_kp_lanes >= 2 && !host_fed && n >= 16;
    const int n_lanes = two_lanes_cfg ? std::max(2, std::min({ctx->opt_kp_lanes, STK_MAX_KP_LANES, n / 8})) : 1;
    // one lane: frames [lo, hi) of the stack through ORB -> 2-NN -> match filter -> findHomography on context `c`
    auto run_lane = [&](stk_ctx* c, int lo, int hi) -> stk_status {
        stk_status st;
        (void)hipSetDevice(c->device);
        const int cnt = hi - lo;
        // frames per ORB batch: the whole lane when it fits its share of a 32 GiB workspace — all lanes together, the helper
        // contexts' workspaces add up (it does for every BASELINE config but the 1024-frame one) — else chunks
        int batch = (int)std::max<size_t>(1, std::min<size_t>((size_t)cnt, (((size_t)32 << 30) / (size_t)n_lanes) / orb_bytes_per_frame(g)));
        batch = (cnt + (cnt + batch - 1) / batch - 1) / ((cnt + batch - 1) / batch);     // even batches: 256 frames as 128 + 128, not 245 + 11
        if (host_fed) batch = std::min(batch, 1 + c->opt_upload_batch);      // measured at 64 x 1080p: 9-frame ORB batches 11.4 ms, 17: 12.8, 65: 15.3
        if ((st = orb_prepare(c, c->kp, ew, eh, g, batch))) return st;
        KeypointWorkspace* ws = c->kp;
        hipStream_t s = c->stream;
        struct TailIdle { hipStream_t t; ~TailIdle() { if (t) (void)hipStreamSynchronize(t); } } tail_idle{c->tail_stream};   // nothing of this lane stays queued on its tail stream, whatever the exit
        if (scaled) HIP_TRY_C(c, ws->gfull.reserve((size_t)w * h));
        HIP_TRY_C(c, ws->desc0.reserve(MAX_KP * 32));
        HIP_TRY_C(c, ws->desc.reserve(MAX_KP * 32 * (size_t)batch));
        HIP_TRY_C(c, ws->knn.reserve(MAX_KP * 16 * (size_t)batch));
        HIP_TRY_C(c, ws->counts.reserve(sizeof(int) * (size_t)batch));
        if (ws->host_knn_cap < MAX_KP * 4 * (size_t)batch) {
            if (ws->host_knn) (void)hipHostFree(ws->host_knn);
            ws->host_knn = nullptr; ws->host_knn_cap = 0;
            HIP_TRY_C(c, hipHostMalloc((void**)&ws->host_knn, MAX_KP * 16 * (size_t)batch, hipHostMallocDefault));
            ws->host_knn_cap = MAX_KP * 4 * (size_t)batch;
        }
        // grey of a frame into level 0 of pyramid `slot`, through scale_image when scaling (utils.rs:186-214)
        auto grey_level0 = [&](const void* frame, int slot) -> stk_status {
            uint8_t* l0 = ws->pyr.as<uint8_t>() + (size_t)slot * g.pyr.total;
            if (depth16) { HIP_TRY_C(c, launch_bgr16_to_grey8(frame, w, h, rb, l0, s)); return STK_OK; }   // grey16 -> (g + 128) / 257
            if (!scaled) { HIP_TRY_C(c, launch_grey(frame, 8, w, h, rb, l0, s, 1, 0, 0, cn)); return STK_OK; }
            HIP_TRY_C(c, launch_grey(frame, 8, w, h, rb, ws->gfull.p, s, 1, 0, 0, cn));
            HIP_TRY_C(c, launch_resize_area_u8(ws->gfull.as<uint8_t>(), w, h, l0, ew, eh, s));
            return STK_OK;
        };
        const std::vector<HostKeypoint>* kp0 = nullptr;       // the reference frame's keypoints / descriptors, once known
        int n0 = 0;
        const uint8_t* desc0 = nullptr;
        // Batches of frames go through ORB together (one launch per stage and level for the whole batch); lane 0's first batch
        // starts with the reference frame, whose descriptors are kept in desc0 (lib.rs:161-175).
        for (int b0 = lo; b0 < hi; b0 += batch) {
            const int nb = std::min(batch, hi - b0);
            if (host_fed && (st = up.wait_frame(b0 + nb - 1, s))) return st;
            // level 0 of every pyramid of the batch: one launch when the frames are evenly spaced in memory (a tensor), else per frame
            bool even = !scaled && nb > 1;
            const ptrdiff_t fstep = nb > 1 ? (const uint8_t*)dev[b0 + 1] - (const uint8_t*)dev[b0] : 0;
            for (int k = 1; even && k + 1 < nb; k++) even = ((const uint8_t*)dev[b0 + k + 1] - (const uint8_t*)dev[b0 + k]) == fstep;
            if (even && fstep > 0 && depth16) HIP_TRY_C(c, launch_bgr16_to_grey8(dev[b0], w, h, rb, ws->pyr.as<uint8_t>(), s, nb, (size_t)fstep, g.pyr.total));
            else if (even && fstep > 0) HIP_TRY_C(c, launch_grey(dev[b0], 8, w, h, rb, ws->pyr.p, s, nb, (size_t)fstep, g.pyr.total, cn));
            else
                for (int k = 0; k < nb; k++)
                    if ((st = grey_level0(dev[b0 + k], k))) return st;
            std::vector<std::vector<HostKeypoint>> kps;
            KPT(c, "lane: batch start");
            // The lane's TAIL — descriptors, 2-NN, homography: microseconds of kernels between host steps — runs on a high-priority
            // stream: on the engine's stream these launches queue behind the other lanes' FAST and fold launches, which fill
            // the device (a 8 us model-scoring launch took 30-160 us to come back, a 58 us refinement 80-145): 2.41 -> 2.38 ms per
            // 64 x 1080p stack, six A/B pairs in one call — the priority helps a launch onto the device, not through it. Only where
            // other lanes compete: on a host-fed stack (one lane, the copy engines busy with uploads) the same stream COSTS 0.4 ms per
            // batch (9.1 -> 12.4 ms per 64 x 1080p stack). The blur-whole-
            // level path keeps everything on s (the descriptors there depend on launches queued on s).
            const hipStream_t ts = c->opt_orb_patch_blur && c->opt_kp_tail_priority && c->tail_stream && !host_fed && n_lanes > 1 ? c->tail_stream : s;
            if ((st = orb_run(c, ws, s, g, nb, threads, ws->desc.as<uint8_t>(), kps, ts))) return st;   // descriptors: queued on ts
            int first = 0;                                         // first moving frame of this batch
            if (b0 == 0) {
                std::lock_guard<std::mutex> lk(ref.m);
                ref.kp0 = kps[0];
                ref.n0 = (int)ref.kp0.size();
                if (ref.n0 > 0) HIP_TRY_C(c, hipMemcpyAsync(ws->desc0.p, ws->desc.p, (size_t)ref.n0 * 32, hipMemcpyDeviceToDevice, ts));
                HIP_TRY_C(c, hipEventRecord(ref.ev, ts));
                ref.desc0 = ws->desc0.as<uint8_t>();
                ref.ready = true;
                ref.cv.notify_all();
                first = 1;
            }
            if (!kp0) {
                std::unique_lock<std::mutex> lk(ref.m);
                ref.cv.wait(lk, [&]() { return ref.ready || ref.failed; });
                if (ref.failed) return STK_PROCESSING_ERROR;       // lane 0 reports its own error
                kp0 = &ref.kp0; n0 = ref.n0; desc0 = ref.desc0;
                if (lo != 0) HIP_TRY_C(c, hipStreamWaitEvent(ts, ref.ev, 0));  // the other lane's copy into desc0
            }
            const int n_mov = nb - first;
            if (n_mov <= 0) { if (ts != s) HIP_TRY_C(c, hipStreamSynchronize(ts)); continue; }
            const int* knn_host = ws->host_knn;
            if (n0 > 0) {
                // knn_match(query = frame-0 descriptors, train = frame-i descriptors, k = 2) for the whole batch  lib.rs:208-219
                std::vector<int> cntv(nb);
                for (int k = 0; k < nb; k++) cntv[k] = (int)kps[k].size();
                HIP_TRY_C(c, hipMemcpyAsync(ws->counts.p, cntv.data(), sizeof(int) * nb, hipMemcpyHostToDevice, ts));
                HIP_TRY_C(c, launch_knn2_hamming(desc0, n0, ws->desc.as<uint8_t>() + (size_t)first * MAX_KP * 32, 0,
                                                 ws->knn.as<int>(), ts, n_mov, ws->counts.as<int>() + first, MAX_KP));
                HIP_TRY_C(c, hipMemcpyAsync(ws->host_knn, ws->knn.p, (size_t)n_mov * n0 * 16, hipMemcpyDeviceToHost, ts));
                HIP_TRY_C(c, hipStreamSynchronize(ts));
            }
            KPT(c, "lane: knn done");
            // C2/C3 on host threads: Lowe ratio, stable sort, truncate, point gather (lib.rs:221-264)
            std::vector<std::vector<float>> from_pts(n_mov), to_pts(n_mov);
            parallel_for(c, n_mov, threads, [&](int m) {
                const int i = b0 + first + m;
                const std::vector<HostKeypoint>& kp = kps[first + m];
                KpAlign& R = results[i];
                R.n_keypoints = (int)kp.size();
                std::vector<Match> ms;
                if (n0 > 0) {
                    const int* knn = knn_host + (size_t)m * n0 * 4;
                    for (int q = 0; q < n0; q++) {
                        if (knn[q * 4] < 0 || knn[q * 4 + 2] < 0) continue;                 // m.len() == 2
                        const float d0 = (float)knn[q * 4 + 1], d1 = (float)knn[q * 4 + 3];
                        if (d0 < params->match_ratio * d1) ms.push_back({q, knn[q * 4], d0});   // Lowe ratio lib.rs:224
                    }
                    std::stable_sort(ms.begin(), ms.end(), [](const Match& a, const Match& b) { return a.d < b.d; });   // lib.rs:233
                    const size_t keep = (size_t)std::round((float)ms.size() * params->match_keep_ratio);              // lib.rs:235
                    if (keep < ms.size()) ms.resize(keep);
                }
                R.n_matches = (int)ms.size();
                if (ms.size() < 5) return;                                                   // lib.rs:240: dropped
                std::vector<float>& dp = from_pts[m];                                        // find_homography(dst_pts, src_pts): frame i -> frame 0
                std::vector<float>& sp = to_pts[m];
                sp.resize(ms.size() * 2); dp.resize(ms.size() * 2);
                for (size_t k = 0; k < ms.size(); k++) {
                    sp[2 * k] = (*kp0)[ms[k].q].x; sp[2 * k + 1] = (*kp0)[ms[k].q].y;        // src_pts: frame 0  lib.rs:245-253
                    dp[2 * k] = kp[ms[k].t].x; dp[2 * k + 1] = kp[ms[k].t].y;                // dst_pts: frame i  lib.rs:256-264
                }
            });
            KPT(c, "lane: match host done");
            // D1 on the device for the whole batch (lib.rs:267-276), D2 checks on the host (lib.rs:279-287)
            std::vector<geom::HgProblem> probs;
            std::vector<int> owner;
            for (int m = 0; m < n_mov; m++)
                if (!from_pts[m].empty()) { probs.push_back({from_pts[m].data(), to_pts[m].data(), (int)from_pts[m].size() / 2, nullptr}); owner.push_back(m); }
            std::vector<geom::HgOutcome> outc(probs.size());
            if (!probs.empty() && !drop_all_moving) {           // (an unknown method: findHomography throws, the frame is skipped — `found` stays 0)
                const int hst = geom::find_homography_batch(c, ts, c->hg, probs.data(), (int)probs.size(), params->method,
                                                            params->ransac_reproj_threshold, outc.data());
                if (hst) return (stk_status)hst;
            }
            KPT(c, "lane: homography done");
            for (size_t k = 0; k < probs.size(); k++) {
                KpAlign& R = results[b0 + first + owner[k]];
                const geom::HgOutcome& o = outc[k];
                if (o.rc != 0 || !o.found) continue;                                         // Err(_) | empty -> skip  lib.rs:275-282
                double* H = R.H;
                for (int q = 0; q < 9; q++) H[q] = o.H[q];
                const double det = H[0] * (H[4] * H[8] - H[5] * H[7]) - H[1] * (H[3] * H[8] - H[5] * H[6]) + H[2] * (H[3] * H[7] - H[4] * H[6]);
                if (std::fabs(det) < 1e-6) continue;                                         // lib.rs:284 / 521 (on the small-image H)
                if (scaled) { H[2] *= fix_sx; H[5] *= fix_sy; H[6] /= fix_sx; H[7] /= fix_sy; }   // utils.rs:236-239
                R.n_inliers = o.n_inliers;
                R.ok = true;
            }
            if (ts != s && (probs.empty() || drop_all_moving)) HIP_TRY_C(c, hipStreamSynchronize(ts));   // (no homography call ended the batch on ts)
        }
        HIP_TRY_C(c, hipStreamSynchronize(s));
        return STK_OK;
    };

    // lane 0 always (the calling thread); helper lanes for device-resident stacks large enough to be worth more pipelines
    std::vector<int> cut(n_lanes + 1);
    for (int k = 0; k <= n_lanes; k++) cut[k] = (int)(((int64_t)n * k + n_lanes - 1) / n_lanes);
    std::vector<stk_status> lane_st(n_lanes, STK_OK);
    // frames become final lane by lane; `on_final` gets them in stack order (a lane's range once every earlier lane is through)
    struct FinalOrder { std::mutex m; std::vector<int> done; int next = 0; stk_status st = STK_OK; } fin;
    fin.done.assign(n_lanes, 0);
    auto lane_finished = [&](int k, stk_status lane_status) {
        if (!on_final) return;
        std::lock_guard<std::mutex> lk(fin.m);
        fin.done[k] = lane_status == STK_OK ? 1 : -1;
        while (fin.next < n_lanes && fin.done[fin.next] == 1 && fin.st == STK_OK) {
            fin.st = (*on_final)(cut[fin.next], cut[fin.next + 1]);
            fin.next++;
        }
    };
    std::vector<std::thread> helpers;
    struct LaneJoin {                                        // also when lane 0 leaves by an exception: wake the helpers, then join them
        std::vector<std::thread>& t; RefShare& r;
        ~LaneJoin() {
            bool any = false;
            for (auto& th : t) any = any || th.joinable();
            if (!any) return;
            { std::lock_guard<std::mutex> lk(r.m); if (!r.ready) r.failed = true; }
            r.cv.notify_all();
            for (auto& th : t) if (th.joinable()) th.join();
        }
    } lane_join{helpers, ref};
    if (n_lanes > 1) {
        // all lanes share this context's host pool (it serves concurrent callers); it must exist before a helper starts
        if (threads > 1 && !ctx->shared_pool && (!ctx->host_pool || ctx->host_pool->size() != threads - 1)) {
            host_pool_destroy(ctx->host_pool);
            ctx->host_pool = new HostPool(threads - 1);
        }
        // the frames may have been produced on this context's stream just now: the helpers' streams start behind it
        HIP_TRY(hipEventRecord(ctx->gate_ev, ctx->stream));
        helpers.reserve(n_lanes - 1);
        for (int k = 1; k < n_lanes; k++) {
            stk_ctx*& hk = ctx->lanes[k - 1];
            if (!hk) {
                if ((st = stk_create(ctx->device, &hk))) return fail(ctx, st, "keypoint lane: helper context creation failed");
                (void)hipSetDevice(ctx->device);
            }
            hk->opt_kp_workers = ctx->opt_kp_workers; hk->opt_orb_patch_blur = ctx->opt_orb_patch_blur; hk->opt_orb_resize_tables = ctx->opt_orb_resize_tables; hk->opt_orb_device_cull = ctx->opt_orb_device_cull; hk->opt_kp_tail_priority = ctx->opt_kp_tail_priority; hk->opt_profile = ctx->opt_profile;
            hk->opt_upload_batch = ctx->opt_upload_batch;
            timing_begin(hk);
            if (threads > 1) hk->shared_pool = ctx->shared_pool ? ctx->shared_pool : ctx->host_pool;
            HIP_TRY(hipStreamWaitEvent(hk->stream, ctx->gate_ev, 0));
            stk_ctx* h = hk;
            helpers.emplace_back([&, h, k]() {
                try { lane_st[k] = run_lane(h, cut[k], cut[k + 1]); lane_finished(k, lane_st[k]); }
                catch (const std::exception& e) { lane_st[k] = fail(h, STK_PROCESSING_ERROR, std::string("helper lane: ") + e.what()); }
            });
        }
    }
    st = run_lane(ctx, 0, cut[1]);
    if (st) { std::lock_guard<std::mutex> lk(ref.m); ref.failed = true; ref.cv.notify_all(); }
    lane_finished(0, st);
    for (auto& th : helpers) if (th.joinable()) th.join();
    (void)hipSetDevice(ctx->device);
    if (st) return st;
    for (int k = 1; k < n_lanes; k++)
        if (lane_st[k]) return fail(ctx, lane_st[k], std::string(stk_last_error(ctx->lanes[k - 1])) + " [helper lane]");
    if (fin.st) return fin.st;
    for (int k = 1; k < n_lanes; k++) {                       // the FAST figures of all lanes in one place
        const stk_ctx* hk = ctx->lanes[k - 1];
        ctx->timing.fast_ms = std::max(ctx->timing.fast_ms, hk->timing.fast_ms);    // they ran side by side
        ctx->timing.fast_launches += hk->timing.fast_launches;
        ctx->timing.fast_pixels += hk->timing.fast_pixels;
    }
    const int n0 = ref.n0;

    if (n_ref_keypoints) *n_ref_keypoints = n0;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (host_fed) {
        double h2d = 0;
        if ((st = up.finish(&h2d))) return st;
        ctx->timing.h2d_ms = h2d; ctx->timing.h2d_bytes = (int64_t)(fb * (size_t)n);
    }
    return STK_OK;
}

extern "C" {

stk_status stk_keypoint_match_shard(stk_ctx* ctx, const stk_frames* frames, const stk_keypoint_params* params,
                                    float scale_down_width, int32_t add_reference, stk_image_f32* sum,
                                    int32_t* n_added, int32_t* n_dropped, stk_frame_stats* stats) {
    stk_status st = check_frames(ctx, frames, true);
    if (st) return st;
    const int w = frames->width, h = frames->height, n = frames->n, cn = frames->channels;
    if ((st = image_check(ctx, sum, w, h, cn))) return st;
    if (sum->location != STK_DEVICE) return fail(ctx, STK_INVALID_PARAMS, "shard sum must be device memory");
    timing_begin(ctx);
    HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    std::vector<KpAlign> results;
    int n0 = 0;
    std::vector<const void*> dev;
    const size_t rb = frame_row_bytes(frames);
    // The fold follows the alignment lane by lane (KpFramesFinal): the frames of a lane are warped into the sum as soon as
    // that lane and every earlier one are through, while later lanes are still in their host steps. Launches accumulate in
    // stack order, so the f32 sum is the one a single launch over all frames gives (each pixel: ((0 + f_a) + f_b) + ...).
    std::vector<WarpFrame> wf(n);                               // entries [0, n_wf) are used; never reallocated (async copies read it)
    int n_wf = 0, dropped = 0;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    HIP_TRY(ctx->warpframes.reserve(sizeof(WarpFrame) * (size_t)n));
    const bool timed = ctx->opt_profile >= 1;
    // The fold's event pairs come from a pool the context keeps (two hipEventCreate per lane on every call sat
    // inside the lanes' critical section); `used` of them belong to this call.
    std::vector<std::pair<hipEvent_t, hipEvent_t>>& fold_ev = ctx->fold_ev;
    size_t fold_used = 0;
    // Declared LAST among the guards, so that it runs FIRST on every way out: nothing of this call is still queued on the
    // stream when `wf` (read by asynchronous copies) or anything else above goes away.
    struct StreamIdle { hipStream_t s; ~StreamIdle() { (void)hipStreamSynchronize(s); } } wf_outlives_its_copies{ctx->stream};
    const KpFramesFinal fold_range = [&](int lo, int hi) -> stk_status {
        (void)hipSetDevice(ctx->device);
        const int first = n_wf;
        for (int i = lo; i < hi; i++) {
            if (i == 0) { if (add_reference) make_warp_frame(wf[n_wf++], dev[0], I3, 0); continue; }
            if (!results[i].ok) { dropped++; continue; }
            make_warp_frame(wf[n_wf++], dev[i], results[i].H, 0);
        }
        const int cnt = n_wf - first;
        if (cnt == 0) return STK_OK;
        for (int k = first; k < n_wf; k++) wf[k].flags = warp_frame_flags(wf[k].src, wf[k].M, rb, w, h, 0);
        if (timed) {
            if (fold_used == fold_ev.size()) {
                hipEvent_t a = nullptr, b = nullptr;
                HIP_TRY(hipEventCreate(&a)); HIP_TRY(hipEventCreate(&b));
                fold_ev.emplace_back(a, b);
            }
            HIP_TRY(hipEventRecord(fold_ev[fold_used].first, ctx->stream));
        }
        HIP_TRY(hipMemcpyAsync(ctx->warpframes.as<WarpFrame>() + first, wf.data() + first, sizeof(WarpFrame) * (size_t)cnt, hipMemcpyHostToDevice, ctx->stream));
        const stk_status fs = warp_fold_enqueue(ctx, cnt, 8, w, h, cn, rb, 1.0 / 255.0, params->border_mode, params->border_value, 0, sum->data,
                                                image_stride_floats(sum), first > 0 ? 1 : 0, first);
        if (fs) return fs;
        if (timed) { HIP_TRY(hipEventRecord(fold_ev[fold_used].second, ctx->stream)); fold_used++; }
        return STK_OK;
    };
    if ((st = keypoint_align_impl(ctx, frames, params, scale_down_width, false, results, &n0, dev, &fold_range))) return st;
    if (stats) {
        std::memset(stats, 0, sizeof(stk_frame_stats) * n);
        stats[0].n_keypoints = n0; stats[0].warp[0] = stats[0].warp[4] = stats[0].warp[8] = 1;
        for (int i = 1; i < n; i++) {
            const KpAlign& R = results[i];
            stats[i].status = R.ok ? 0 : 1; stats[i].n_keypoints = R.n_keypoints; stats[i].n_matches = R.n_matches; stats[i].n_inliers = R.n_inliers;
            for (int k = 0; k < 9; k++) stats[i].warp[k] = R.H[k];
        }
    }
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    if (n_wf == 0) HIP_TRY(hipMemsetAsync(sum->data, 0, image_stride_floats(sum) * h * sizeof(float), ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->timing.align_ms = ev_ms(ctx->ev[0], ctx->ev[1]);       // includes the folds that ran under later lanes
    ctx->timing.warp_ms = 0;
    for (size_t k = 0; k < fold_used; k++) ctx->timing.warp_ms += ev_ms(fold_ev[k].first, fold_ev[k].second);
    if (n_added) *n_added = (int32_t)n_wf;
    if (n_dropped) *n_dropped = dropped;
    return STK_OK;
}

stk_status stk_keypoint_match(stk_ctx* ctx, const stk_frames* frames, const stk_keypoint_params* params,
                              float scale_down_width, stk_image_f32* out, int32_t* dropped, stk_frame_stats* stats) {
    if (ctx && ctx->multi) return multi_match(ctx, 1, frames, params, nullptr, scale_down_width, out, dropped, stats);
    stk_status st = check_frames(ctx, frames, true);
    if (st) return st;
    if ((st = image_check(ctx, out, frames->width, frames->height, frames->channels))) return st;
    if (out->row_stride_bytes) return fail(ctx, STK_INVALID_PARAMS, "output must be tightly packed");
    (void)hipSetDevice(ctx->device);
    const size_t nel = (size_t)frames->width * frames->height * frames->channels;
    stk_image_f32 sum = *out;
    if (out->location != STK_DEVICE) {
        HIP_TRY(ctx->acc.reserve(nel * sizeof(float)));
        sum.data = ctx->acc.as<float>(); sum.location = STK_DEVICE;
    }
    int32_t added = 0, ndrop = 0;
    if ((st = stk_keypoint_match_shard(ctx, frames, params, scale_down_width, 1, &sum, &added, &ndrop, stats))) return st;
    if (dropped) *dropped = ndrop;
    if (added <= 0)   // lib.rs:324
        return fail(ctx, STK_INVALID_PARAMS, "All images discarded: try modifying KeyPointMatchParameters::match_distance_threshold");
    const stk_timing keep = ctx->timing;
    st = stk_finalize_mean(ctx, &sum, frames->n - ndrop, out);   // lib.rs:342: img / (n - dropped)
    const double fin = ctx->timing.finalize_ms;
    ctx->timing = keep; ctx->timing.finalize_ms = fin;
    return st;
}

}  // extern "C"
