// stacker.cpp — host runtime and C ABI (include/stacker.h) of the MI355X align-and-stack engine.
//
// Mirrors the reference's drivers: ecc_match_no_scaling (lib.rs:719-847) and
// keypoint_match_no_scale (lib.rs:146-353). Where the reference runs a Rayon map-reduce over
// frame indices with one f32 accumulator per worker thread, this runtime keeps every frame of the
// shard resident in HBM, runs the alignment of ALL frames as a device-side work queue and then
// folds every warped frame into one accumulator in a single launch.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <functional>
#include <mutex>

#include "context.h"
#include "upload.h"

using namespace stk;

size_t frame_row_bytes(const stk_frames* f) {
    return f->row_stride_bytes ? f->row_stride_bytes : (size_t)f->width * f->channels * (f->depth / 8);
}

// Bring the frames of a stack into HBM (no copy when they already are).
stk_status resolve_frames(stk_ctx* ctx, const stk_frames* f, std::vector<const void*>& dev) {
    const size_t rb = frame_row_bytes(f), fb = rb * f->height;
    dev.resize(f->n);
    if (f->location == STK_DEVICE) { for (int i = 0; i < f->n; i++) dev[i] = f->data[i]; return STK_OK; }
    HIP_TRY(ctx->frames.reserve(fb * f->n));
    for (int i = 0; i < f->n; i++) {
        void* d = ctx->frames.as<uint8_t>() + fb * i;
        HIP_TRY(hipMemcpyAsync(d, f->data[i], fb, hipMemcpyHostToDevice, ctx->stream));
        dev[i] = d;
    }
    return STK_OK;
}

stk_status check_frames(stk_ctx* ctx, const stk_frames* f, bool need_bgr) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (!f || f->n <= 0 || !f->data) return fail(ctx, STK_NOT_ENOUGH_FILES, "Not enough files");
    if (f->width <= 0 || f->height <= 0) return fail(ctx, STK_INVALID_PARAMS, "bad frame geometry");
    if (f->depth != 8 && f->depth != 16 && f->depth != 32) return fail(ctx, STK_INVALID_PARAMS, "depth must be 8, 16 or 32");
    // cvtColor(BGR2GRAY) takes 3 or 4 channels (utils.rs:136-142); IMREAD_UNCHANGED hands a PNG's alpha plane through (utils.rs:132)
    if (need_bgr && f->channels != 3 && f->channels != 4)
        return fail(ctx, STK_BACKEND_ERROR, "cvtColor(BGR2GRAY): frames must have 3 or 4 channels (utils.rs:136)");
    if (f->channels != 1 && f->channels != 3 && f->channels != 4) return fail(ctx, STK_INVALID_PARAMS, "channels must be 1, 3 or 4");
    if ((size_t)f->width * f->height > (size_t)1 << 30) return fail(ctx, STK_INVALID_PARAMS, "frame too large");
    return STK_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" {

const char* stk_version(void) { return "libstacker_rs_amd 0.1 (gfx950)"; }

stk_status stk_create(int32_t device_id, stk_ctx** out) {
    if (!out) return STK_INVALID_PARAMS;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return STK_HIP_ERROR;
    if (hipSetDevice(device_id) != hipSuccess) return STK_HIP_ERROR;
    stk_ctx* ctx = new stk_ctx();
    ctx->device = device_id;
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return STK_HIP_ERROR; }
    ctx->stream = ctx->own_stream;
    if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->prep_stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->ecc_stream2, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->gate_ev, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->gate_ev2, hipEventDisableTiming) != hipSuccess) { delete ctx; return STK_HIP_ERROR; }
    {
        int least = 0, greatest = 0;                          // (numerically lower = higher priority)
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
        if (hipStreamCreateWithPriority(&ctx->tail_stream, hipStreamNonBlocking, greatest) != hipSuccess) {
            (void)hipGetLastError();                           // no priorities on this device: the lanes then keep one stream
            ctx->tail_stream = nullptr;
        }
    }
    for (auto& e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { delete ctx; return STK_HIP_ERROR; }
    for (auto& e : ctx->poll_ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { delete ctx; return STK_HIP_ERROR; }
    if (hipHostMalloc((void**)&ctx->host_done, 64, hipHostMallocDefault) != hipSuccess) { delete ctx; return STK_HIP_ERROR; }
    if (const char* e = getenv("STK_ECC_GROUPS")) ctx->opt_ecc_groups = std::max(0, std::min(2, atoi(e)));   // test hook: the default of the option
    ctx->kp = keypoint_workspace_create();
    ctx->hg = geom::hg_workspace_create();
    *out = ctx;
    return STK_OK;
}

void stk_destroy(stk_ctx* ctx) {
    if (!ctx) return;
    multi_destroy(ctx);                    // the other devices' contexts, RCCL communicators
    for (stk_ctx*& h : ctx->lanes) if (h) { stk_destroy(h); h = nullptr; }
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (DevBuf* b : {&ctx->frames, &ctx->ref, &ctx->blur_tmp, &ctx->templates, &ctx->slots, &ctx->queue, &ctx->results,
                      &ctx->partials, &ctx->warpframes, &ctx->acc, &ctx->scratch, &ctx->init_warps, &ctx->frameptrs})
        b->release();
    keypoint_workspace_destroy(ctx->kp);
    geom::hg_workspace_destroy(ctx->hg);
    host_pool_destroy(ctx->host_pool);
    for (auto& e : ctx->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->poll_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ctx->prof_ev) if (e) (void)hipEventDestroy(e);
    for (auto& pr : ctx->fold_ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (ctx->host_done) (void)hipHostFree(ctx->host_done);
    if (ctx->files_block) { if (ctx->files_block_pinned) (void)hipHostFree(ctx->files_block); else std::free(ctx->files_block); }
    for (auto& e : ctx->upload_events) if (e) (void)hipEventDestroy(e);
    if (ctx->gate_ev) (void)hipEventDestroy(ctx->gate_ev);
    if (ctx->gate_ev2) (void)hipEventDestroy(ctx->gate_ev2);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->prep_stream) (void)hipStreamDestroy(ctx->prep_stream);
    if (ctx->ecc_stream2) (void)hipStreamDestroy(ctx->ecc_stream2);
    if (ctx->tail_stream) (void)hipStreamDestroy(ctx->tail_stream);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* stk_last_error(const stk_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

stk_status stk_set_stream(stk_ctx* ctx, void* hip_stream) {
    if (!ctx) return STK_INVALID_PARAMS;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return STK_OK;
}

stk_status stk_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return STK_INVALID_PARAMS;
    *out = nullptr;
    return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? STK_OK : STK_HIP_ERROR;
}
void stk_host_free(void* p) { if (p) (void)hipHostFree(p); }

stk_status stk_get_timing(const stk_ctx* ctx, stk_timing* out) {
    if (!ctx || !out) return STK_INVALID_PARAMS;
    *out = ctx->timing;
    return STK_OK;
}

stk_status stk_set_option(stk_ctx* ctx, const char* name, int64_t value) {
    if (!ctx || !name) return STK_INVALID_PARAMS;
    for (int i = 1; i < multi_member_count(ctx); i++) {          // a multi-device context: every device gets the knob
        const stk_status st = set_option_one(multi_member(ctx, i), name, value);
        if (st) return fail(ctx, st, stk_last_error(multi_member(ctx, i)));
    }
    return set_option_one(ctx, name, value);
}

}  // extern "C"

stk_status set_option_one(stk_ctx* ctx, const char* name, int64_t value) {
    const std::string n(name);
    if (n == "ecc_slots") { if (value < 0 || value > 256) return fail(ctx, STK_INVALID_PARAMS, "ecc_slots out of range"); ctx->opt_ecc_slots = (int)value; }
    else if (n == "warp_subpixel_bits") { if (value != 0 && value != 5) return fail(ctx, STK_INVALID_PARAMS, "warp_subpixel_bits must be 0 or 5"); ctx->opt_subpixel_bits = (int)value; }
    else if (n == "profile") ctx->opt_profile = (int)value;
    else if (n == "ecc_chunk") { if (value < 0 || value > 64) return fail(ctx, STK_INVALID_PARAMS, "ecc_chunk out of range"); ctx->opt_ecc_chunk = (int)value; }
    else if (n == "ecc_ring") ctx->opt_ecc_ring = value != 0;
    else if (n == "ecc_groups") { if (value < 0 || value > 2) return fail(ctx, STK_INVALID_PARAMS, "ecc_groups must be 0 (auto), 1 or 2"); ctx->opt_ecc_groups = (int)value; }
    else if (n == "ecc_ring_lookahead") { if (value < 1 || value > 5) return fail(ctx, STK_INVALID_PARAMS, "ecc_ring_lookahead must be 1..5"); ctx->opt_ecc_ring_lookahead = (int)value; }
    else if (n == "ecc_variant") { if (value != 0 && value != 3) return fail(ctx, STK_INVALID_PARAMS, "ecc_variant must be 3 (production) or 0 (direct cross-check)"); ctx->opt_ecc_variant = (int)value; }
    else if (n == "profile_stride") { if (value < 1 || value > 1024) return fail(ctx, STK_INVALID_PARAMS, "profile_stride out of range"); ctx->opt_profile_stride = (int)value; }
    else if (n == "kp_lanes") { if (value < 1 || value > STK_MAX_KP_LANES) return fail(ctx, STK_INVALID_PARAMS, "kp_lanes must be 1..8"); ctx->opt_kp_lanes = (int)value; }
    else if (n == "orb_resize_tables") ctx->opt_orb_resize_tables = value != 0;
    else if (n == "kp_tail_priority") ctx->opt_kp_tail_priority = value != 0;
    else if (n == "orb_device_cull") ctx->opt_orb_device_cull = value != 0;
    else if (n == "orb_patch_blur") ctx->opt_orb_patch_blur = value != 0;
    else if (n == "kp_workers") { if (value < 1 || value > 16) return fail(ctx, STK_INVALID_PARAMS, "kp_workers out of range"); ctx->opt_kp_workers = (int)value; }
    else if (n == "warp_tune") ctx->opt_warp_tune = (int)value;
    else if (n == "prep_stream") ctx->opt_prep_stream = value != 0;
    else if (n == "prep_overlap") ctx->opt_prep_overlap = value != 0;
    else if (n == "upload_batch") { if (value < 1 || value > 1024) return fail(ctx, STK_INVALID_PARAMS, "upload_batch out of range"); ctx->opt_upload_batch = (int)value; }
    else if (n == "ecc_blocks") { if (value != 0 && (value < 8 || value > 65536)) return fail(ctx, STK_INVALID_PARAMS, "ecc_blocks out of range"); ctx->opt_ecc_blocks = (int)value; }
    else return fail(ctx, STK_INVALID_PARAMS, "unknown option " + n);
    return STK_OK;
}


float ev_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }

// ---------------------------------------------------------------------------------------------
// ECC machinery shared by stk_ecc_match_shard and stk_find_transform_ecc
// ---------------------------------------------------------------------------------------------
constexpr int REF_PLANES = 8;     // I, gx, gy, (gx, gy) x 2 floats, (I, gx, gy) x 3 floats
struct EccPlan {
    int w, h, n_templates, motion;
    int templ_row_stride;
    size_t templ_plane_stride;
    int ref_stride;
    size_t ref_plane_floats;
    int n_slots, nb, nsums;
};

static stk_status ecc_validate(stk_ctx* ctx, const stk_ecc_params* p, EccCriteria& crit) {
    if (!p) return fail(ctx, STK_INVALID_PARAMS, "null params");
    if (p->motion_type < 0 || p->motion_type > 3) return fail(ctx, STK_INVALID_PARAMS, "bad motion type");
    // TermCriteria with neither COUNT nor EPS: OpenCV's CV_Assert fails -> OpenCvError (utils.rs:159-170)
    if (!p->has_max_count && !p->has_epsilon)
        return fail(ctx, STK_BACKEND_ERROR, "findTransformECC: criteria needs COUNT and/or EPS");
    if (p->has_max_count && p->max_count < 0) return fail(ctx, STK_BACKEND_ERROR, "findTransformECC: maxCount < 0");
    if (p->has_epsilon && p->epsilon < 0) return fail(ctx, STK_BACKEND_ERROR, "findTransformECC: epsilon < 0");
    if (p->gauss_filt_size <= 0 || p->gauss_filt_size % 2 == 0)
        return fail(ctx, STK_BACKEND_ERROR, "GaussianBlur: kernel size must be odd and positive");
    if (p->gauss_filt_size > 63) return fail(ctx, STK_NOT_IMPLEMENTED, "gauss_filt_size > 63 is not supported");
    crit.n_iter = p->has_max_count ? p->max_count : 200;
    crit.eps = p->has_epsilon ? p->epsilon : -1;
    return STK_OK;
}

static stk_status ecc_plan(stk_ctx* ctx, int w, int h, int n_templates, int motion, EccPlan& pl) {
    pl.w = w; pl.h = h; pl.n_templates = n_templates; pl.motion = motion;
    pl.templ_row_stride = (w + 3) & ~3;
    pl.templ_plane_stride = (size_t)pl.templ_row_stride * h;
    pl.ref_stride = (w + 2 * REF_PAD + 3) & ~3;
    pl.ref_plane_floats = (size_t)pl.ref_stride * (h + 2 * REF_PAD);
    const int P = motion == STK_MOTION_HOMOGRAPHY ? 8 : motion == STK_MOTION_AFFINE ? 6 : motion == STK_MOTION_EUCLIDEAN ? 3 : 2;
    pl.nsums = ecc_nsums(P);
    // Frames in flight per launch ("slots"). Every (iterate, solve) launch pair costs ~25 us of solve latency and launch
    // gaps whatever it carries, so the more frames share it the better; and the stack should go through the slots in EQUAL
    // rounds (63 frames in 48 slots leave a second round of 15). Round 3, re-measured with the column-walking kernel (A/B
    // within one call, ms per stack): what costs most is a SECOND round — its frames enter as the first round's converge,
    // all within a few launches of each other, and every launch from there to the end is part empty. 4K: 63 frames in one
    // round of 63 slots 14.4-14.6, in two rounds of 32 15.7-16.0; 95 frames 95 slots 21.4-21.8, 2 x 48 22.2-22.4; 127 frames
    // 127 slots 28.1, 3 x 43 28.4; 255 frames 3 x 85 55.9-56.1, 73 / 102 slots 55.9 / 56.2, 5 x 51 56.1-56.5, 6 x 43 56.6-57.0,
    // 1 x 255 56.1-56.4 (nothing hides the templates' preparation), 2 x 128 57.4-57.6, 4 x 64 56.9-57.3. So: one round up to
    // 128 frames; beyond, at least three equal rounds of at most 96 (4K class) — 64 slots per round for frames up to 1080p
    // as before (63 x 1080p: 32 slots 3.91 ms, 48 4.30, 63 3.76).
    // Workgroups per frame: a function of the FRAME SIZE only (288 at 4K, see below), so a frame's f32 summation partition
    // depends on nothing else: its warp is bit-identical however the stack is sharded over GPUs and however many frames
    // happen to share the launch.
    int slots = ctx->opt_ecc_slots;
    if (slots <= 0) {
        const int nt = std::max(n_templates, 1);
        if ((size_t)w * h <= (size_t)1920 * 1088) {
            const int rounds = (nt + 63) / 64;
            slots = (nt + rounds - 1) / rounds;
        } else if (nt <= 128) slots = nt;
        else {
            const int rounds = std::max(3, (nt + 95) / 96);
            slots = (nt + rounds - 1) / rounds;
        }
    }
    pl.n_slots = std::max(1, std::min(slots, std::max(n_templates, 1)));
    // Round 3: a wavefront walks ~112 rows of a 64-pixel column strip before it folds its 66 sums across the lanes (266
    // instructions against 67 per row): 288 workgroups is that at 4K, but a 1080p frame cut into 264 left 31 rows per wave
    // and 13 % of the pass in the folds. Now the workgroup count follows the frame: (strips x rows) / (4 waves x 112 rows),
    // at most 288, at least 64 where the frame has that many 4-row groups (a lone small frame still wants parallelism).
    // 64 x 1080p: 2.60 -> 2.41 ms of alignment with 72 workgroups per frame. Still a function of the frame size only.
    const int units = (h + 3) / 4;                           // 4-row groups of the frame
    const long long strip_rows = (long long)((w + 63) / 64) * h;
    int nb = ctx->opt_ecc_blocks > 0 ? ctx->opt_ecc_blocks / pl.n_slots
                                     : (int)std::max<long long>(std::min(64, units), std::min<long long>(288, strip_rows / (4 * 112)));
    nb = std::max(8, std::min(units, nb));
    nb = std::max(8, (nb / 8) * 8);                          // multiple of 8: XCD-aware block decode
    pl.nb = nb;
    HIP_TRY(ctx->ref.reserve(pl.ref_plane_floats * REF_PLANES * sizeof(float) + 4096));   // (+ the overhang of a ring row behind the last plane)
    HIP_TRY(ctx->blur_tmp.reserve(pl.templ_plane_stride * sizeof(float)));
    // (+ 4 rows: the column pass prefetches template rows up to three past the end of a strip, also behind the last row of the last frame)
    HIP_TRY(ctx->templates.reserve(pl.templ_plane_stride * sizeof(float) * std::max(n_templates, 1) + 4 * (size_t)pl.templ_row_stride * sizeof(float) + 1024));
    HIP_TRY(ctx->slots.reserve(sizeof(EccSlot) * pl.n_slots));
    HIP_TRY(ctx->queue.reserve(sizeof(EccQueue)));
    HIP_TRY(ctx->results.reserve(sizeof(EccFrameResult) * std::max(n_templates, 1)));
    HIP_TRY(ctx->partials.reserve(sizeof(double) * pl.n_slots * ((size_t)pl.nb * pl.nsums + ECC_MAX_SUMS) + sizeof(int) * pl.n_slots));
    return STK_OK;
}

// frame-0 side: grey -> blur -> zero-padded I/gx/gy planes
static stk_status ecc_prepare_reference(stk_ctx* ctx, const EccPlan& pl, const void* img, int depth, int cn,
                                        size_t stride_bytes, int gauss) {
    HIP_TRY(launch_grey_blur(img, depth, cn, pl.w, pl.h, stride_bytes, gauss, ctx->blur_tmp.as<float>(), pl.templ_row_stride, ctx->stream));
    // the zero border is written once per geometry: ref_planes_kernel only ever writes the interior, so a stack of the same
    // size as the last one (every step of a shard's life) finds the border as it left it (171 MB of memset = 22 us at 4K)
    if (ctx->ref_zeroed_ptr != ctx->ref.p || ctx->ref_zeroed_w != pl.w || ctx->ref_zeroed_h != pl.h) {
        HIP_TRY(hipMemsetAsync(ctx->ref.p, 0, pl.ref_plane_floats * REF_PLANES * sizeof(float), ctx->stream));
        ctx->ref_zeroed_ptr = ctx->ref.p; ctx->ref_zeroed_w = pl.w; ctx->ref_zeroed_h = pl.h;
    }
    float* base = ctx->ref.as<float>() + (size_t)REF_PAD * pl.ref_stride + REF_PAD;
    float* gxy = ctx->ref.as<float>() + 3 * pl.ref_plane_floats + 2 * ((size_t)REF_PAD * pl.ref_stride + REF_PAD);
    float* igg = ctx->ref.as<float>() + 5 * pl.ref_plane_floats + 3 * ((size_t)REF_PAD * pl.ref_stride + REF_PAD);
    HIP_TRY(launch_ref_planes(ctx->blur_tmp.as<float>(), pl.templ_row_stride, pl.w, pl.h, base, base + pl.ref_plane_floats,
                              base + 2 * pl.ref_plane_floats, gxy, igg, pl.ref_stride, ctx->stream));
    return STK_OK;
}

// run the device-side iteration queue to completion; results copied to `res`
// `feed`, when given, is the producer of a host-fed stack: feed(false) enqueues the preparation of every batch of frames
// that has arrived since the last call (raising the queue's ready count behind it) and returns the number of templates
// whose preparation has been enqueued so far; feed(true) first blocks until at least one more batch has arrived.
using EccFeed = std::function<stk_status(bool block, int* enqueued)>;
// `on_done`, when given, is called once the device-side queue is known to have drained, BEFORE the results travel to the
// host: the caller enqueues there what may follow the alignment in stream order without the host's help (the fold).
using EccDone = std::function<stk_status()>;
static stk_status ecc_run(stk_ctx* ctx, const EccPlan& pl, EccCriteria crit, const float* init_warps_dev,
                          std::vector<EccFrameResult>& res, const EccFeed* feed = nullptr, const EccDone* on_done = nullptr) {
    res.resize(pl.n_templates);
    if (pl.n_templates == 0) return STK_OK;
    // for (i = 1; i <= nIter && fabs(rho - last_rho) >= eps; i++) with rho = -1, last_rho = -eps: not even the first
    // iteration runs when |eps - 1| < eps, i.e. eps > 0.5 (and never for nIter < 1)
    if (crit.n_iter >= 1 && !(std::fabs(-1.0 - (-crit.eps)) >= crit.eps)) crit.n_iter = 0;
    EccIterArgs a{};
    const float* base = ctx->ref.as<float>() + (size_t)REF_PAD * pl.ref_stride + REF_PAD;
    const float* gxy = ctx->ref.as<float>() + 3 * pl.ref_plane_floats + 2 * ((size_t)REF_PAD * pl.ref_stride + REF_PAD);
    const float* igg = ctx->ref.as<float>() + 5 * pl.ref_plane_floats + 3 * ((size_t)REF_PAD * pl.ref_stride + REF_PAD);
    a.ref = RefPlanes{base, base + pl.ref_plane_floats, base + 2 * pl.ref_plane_floats, gxy, igg, pl.ref_stride, pl.w, pl.h};
    a.templates = ctx->templates.as<float>();
    a.templ_plane_stride = pl.templ_plane_stride;
    a.templ_row_stride = pl.templ_row_stride;
    a.tw = pl.w; a.th = pl.h;
    a.slots = ctx->slots.as<EccSlot>();
    a.n_slots = pl.n_slots;
    a.nb = pl.nb;
    a.ring = ctx->opt_ecc_ring;
    a.ring_lookahead = ctx->opt_ecc_ring_lookahead;
    a.partials = ctx->partials.as<double>();
    a.sums = a.partials + (size_t)pl.n_slots * pl.nb * pl.nsums;
    a.tickets = reinterpret_cast<int*>(a.sums + (size_t)pl.n_slots * ECC_MAX_SUMS);
    EccQueue* q = ctx->queue.as<EccQueue>();
    EccFrameResult* r = ctx->results.as<EccFrameResult>();
    a.ring_fallbacks = &q->ring_fallbacks;
    a.slot0 = 0;
    HIP_TRY(launch_ecc_init(a.slots, pl.n_slots, a.tickets, q, pl.n_templates, r, init_warps_dev, ctx->stream, feed ? 0 : -1));
    int fed = feed ? 0 : pl.n_templates;
    if (feed) {
        // the prep stream may raise `ready` only after the queue exists
        HIP_TRY(hipEventRecord(ctx->gate_ev, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->prep_stream, ctx->gate_ev, 0));
        stk_status fs = (*feed)(crit.n_iter >= 1 ? false : true, &fed);
        if (fs) return fs;
        while (crit.n_iter < 1 && fed < pl.n_templates) { if ((fs = (*feed)(true, &fed))) return fs; }   // no iterations: just drain the producer
    }
    if (crit.n_iter >= 1) {
        // Enqueue chunks of (iterate, solve) launches; keep two chunks in flight and poll the
        // device-side completion counter behind each. Launches after completion are no-ops.
        // (0 = by frame size: the launches behind the last converged frame are empty but not free, ~13 us a pair, and up to two
        // chunks of them are queued by the time the host hears of it: 2 for 4K-class frames, whose pairs are long enough to
        // poll after every second one — 32 x 4K 7.72-7.75 -> 7.63-7.69 ms per stack —, 4 for smaller frames)
        const int chunk = ctx->opt_ecc_chunk > 0 ? ctx->opt_ecc_chunk : ((size_t)pl.w * pl.h > (size_t)1920 * 1088 ? 2 : 4);
        // Two slot GROUPS (option ecc_groups = 2): the slots are cut in two halves with their own (iterate, solve) launch sequences
        // on two streams, so that one half's solve launch (one workgroup per slot, ~20 us of an otherwise idle device) and
        // the fill / drain of its iteration launches run under the other half's iteration launch. The queue, the results and
        // the counters are shared (device-scope atomics); which slot a frame lands in never shows in its bits.
        // Measured (A/B in one call): 64 x 1080p 3.01 -> 2.88 ms per stack; 256 x 4K 56.7 -> 56.0 ms; 32 x 4K 7.66 -> 7.83 ms (two
        // half-filled tails). Default (0 = auto): two groups for frames up to 1080p with at least 32 slots when every template exists
        // before the first launch, else one. With
        // per-launch event pairs (profile = 2) always one: a bracketed launch must not share the device with another.
        // (not while frames are still arriving: host-fed, 64 x 1080p from pinned memory 9.2 -> 11.1 ms with two sequences)
        const int want = ctx->opt_ecc_groups ? ctx->opt_ecc_groups : ((size_t)pl.w * pl.h <= (size_t)1920 * 1088 && pl.n_slots >= 32 && !feed ? 2 : 1);
        const int groups = (want >= 2 && pl.n_slots >= 8 && ctx->opt_profile < 2) ? 2 : 1;
        EccIterArgs ag[2] = {a, a};
        hipStream_t sg[2] = {ctx->stream, ctx->ecc_stream2};
        if (groups == 2) {
            ag[0].n_slots = (pl.n_slots + 1) / 2;
            ag[1].slot0 = ag[0].n_slots; ag[1].n_slots = pl.n_slots - ag[0].n_slots;
            HIP_TRY(hipEventRecord(ctx->gate_ev2, ctx->stream));                  // behind ecc_init
            HIP_TRY(hipStreamWaitEvent(sg[1], ctx->gate_ev2, 0));
        }
        struct SecondStreamIdle { hipStream_t s; bool on; ~SecondStreamIdle() { if (on) (void)hipStreamSynchronize(s); } } second_idle{sg[1], groups == 2};
        int inflight = 0, head = 0;
        long long launched = 0;
        const long long max_launches = 2 * ((long long)crit.n_iter * pl.n_templates + 2 * chunk) + 4;
        bool done = false;
        size_t prof_used = 0;
        if (ctx->opt_profile >= 2 && ctx->prof_ev.empty()) {
            ctx->prof_ev.resize(8192);
            for (auto& e : ctx->prof_ev) HIP_TRY(hipEventCreate(&e));
        }
        ctx->host_done[0] = ctx->host_done[1] = 0;
        int last_done = 0;
        while (!done) {
            if (feed && fed < pl.n_templates) {
                // everything prepared so far has been aligned and the two chunks in flight cannot find work: wait for PCIe
                const bool starved = last_done >= fed && inflight == 0;
                stk_status fs = (*feed)(starved, &fed);
                if (fs) return fs;
            }
            while (inflight < 2) {
                for (int c = 0; c < chunk; c++) {
                    // per-launch timing (profile >= 2) brackets every `profile_stride`-th pixel pass with an event pair: an
                    // event between two kernels keeps them from being dispatched back to back (~5 % of the step if every
                    // launch is bracketed), so bench.py samples
                    const bool timed = ctx->opt_profile >= 2 && prof_used + 2 <= ctx->prof_ev.size() &&
                                       (ctx->timing.ecc_iter_launches + c) % ctx->opt_profile_stride == 0;
                    if (timed) HIP_TRY(hipEventRecord(ctx->prof_ev[prof_used], ctx->stream));
                    HIP_TRY(launch_ecc_iter(ag[0], pl.motion, ctx->opt_ecc_variant, sg[0]));
                    if (timed) { HIP_TRY(hipEventRecord(ctx->prof_ev[prof_used + 1], ctx->stream)); prof_used += 2; }
                    HIP_TRY(launch_ecc_solve(ag[0], pl.motion, crit, q, r, sg[0], init_warps_dev));
                    if (groups == 2) {
                        HIP_TRY(launch_ecc_iter(ag[1], pl.motion, ctx->opt_ecc_variant, sg[1]));
                        HIP_TRY(launch_ecc_solve(ag[1], pl.motion, crit, q, r, sg[1], init_warps_dev));
                    }
                }
                launched += chunk;
                ctx->timing.ecc_iter_launches += chunk;
                const int slot = (head + inflight) & 1;
                HIP_TRY(hipMemcpyAsync(&ctx->host_done[slot], &q->frames_done, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(hipEventRecord(ctx->poll_ev[slot], ctx->stream));
                inflight++;
            }
            HIP_TRY(hipEventSynchronize(ctx->poll_ev[head]));
            last_done = ctx->host_done[head];
            if (last_done >= pl.n_templates) done = true;
            head ^= 1; inflight--;
            if (feed && !done && last_done >= fed && fed < pl.n_templates) {
                // starved: let the chunk still in flight drain, then block on the producer at the top of the loop
                while (inflight > 0) { HIP_TRY(hipEventSynchronize(ctx->poll_ev[head])); last_done = ctx->host_done[head]; head ^= 1; inflight--; }
                launched = 0;                                   // idle polling does not count against the drain guard
                continue;
            }
            if (!done && launched > max_launches)
                return fail(ctx, STK_PROCESSING_ERROR, "ECC queue did not drain (internal error)");
        }
        if (groups == 2) {                                       // the second group's remaining (empty) launches end before anything that follows
            HIP_TRY(hipEventRecord(ctx->gate_ev2, sg[1]));
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->gate_ev2, 0));
        }
        if (on_done) { const stk_status ds = (*on_done)(); if (ds) return ds; }
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        for (size_t i = 0; i + 1 < prof_used; i += 2) {
            ctx->timing.ecc_iter_ms += ev_ms(ctx->prof_ev[i], ctx->prof_ev[i + 1]);
            ctx->timing.ecc_iter_timed += 1;
        }
    }
    HIP_TRY(hipMemcpyAsync(res.data(), r, sizeof(EccFrameResult) * pl.n_templates, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(&ctx->host_done[2], &q->ring_fallbacks, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->timing.ecc_ring_fallbacks += ctx->host_done[2];
#ifdef STK_SOLVE_TIMING
    { EccQueue hq; (void)hipMemcpy(&hq, q, sizeof(hq), hipMemcpyDeviceToHost);
      fprintf(stderr, "SOLVE_DBG stage1 %lld ticket %lld sums %lld stats %lld lu %lld iph %lld tail %lld (x10ns)\n", hq.dbg[1] - hq.dbg[0], hq.dbg[2] - hq.dbg[1],
              hq.dbg[4] - hq.dbg[2], hq.dbg[5] - hq.dbg[4], hq.dbg[6] - hq.dbg[5], hq.dbg[7] - hq.dbg[6], hq.dbg[8] - hq.dbg[7]); }
#endif
    if (crit.n_iter < 1)
        for (auto& e : res) { for (int k = 0; k < 9; k++) e.warp[k] = (k % 4 == 0) ? 1.f : 0.f; e.iters = 0; e.status = 0; e.rho = -1; }
    for (auto& e : res) ctx->timing.ecc_slot_iterations += e.iters;
    return STK_OK;
}

static const char* ecc_status_message(int st) {
    switch (st) {
        case 1: return "findTransformECC: NaN encountered (StsNoConv)";
        case 2: return "findTransformECC: the algorithm stopped before its convergence; the correlation is going to be minimized (StsNoConv)";
        default: return "findTransformECC: frame was not processed";
    }
}

// fold frames into `sum` (device, tightly packed or strided) through their warps
stk_status warp_fold(stk_ctx* ctx, std::vector<WarpFrame>& wf, int depth, int w, int h, int cn,
                            size_t src_row_bytes, double alpha, int border_mode, const double* border_value,
                            int is_affine, float* acc, size_t acc_stride_floats, int accumulate, int dw, int dh) {
    if (wf.empty()) return STK_OK;
    // per frame, once: the flags the fast kernels branch on (common.h: warp_frame_flags; the rectangle is the DESTINATION's)
    for (WarpFrame& f : wf) f.flags = warp_frame_flags(f.src, f.M, src_row_bytes, dw > 0 ? dw : w, dh > 0 ? dh : h, is_affine);
    HIP_TRY(ctx->warpframes.reserve(sizeof(WarpFrame) * wf.size()));
    HIP_TRY(hipMemcpyAsync(ctx->warpframes.p, wf.data(), sizeof(WarpFrame) * wf.size(), hipMemcpyHostToDevice, ctx->stream));
    stk_status st = warp_fold_enqueue(ctx, (int)wf.size(), depth, w, h, cn, src_row_bytes, alpha, border_mode, border_value, is_affine,
                                      acc, acc_stride_floats, accumulate, 0, dw, dh);
    if (st) return st;
    // the host vector may die before the copy above ran if the caller does not synchronise
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STK_OK;
}

// the fold launch itself over entries [first_frame, first_frame + n_frames) of ctx->warpframes (device memory); asynchronous
stk_status warp_fold_enqueue(stk_ctx* ctx, int n_frames, int depth, int w, int h, int cn, size_t src_row_bytes, double alpha,
                             int border_mode, const double* border_value, int is_affine, float* acc, size_t acc_stride_floats,
                             int accumulate, int first_frame, int dw, int dh) {
    WarpArgs a{};
    a.frames = ctx->warpframes.as<WarpFrame>() + first_frame;
    a.n_frames = n_frames;
    a.sw = w; a.sh = h; a.cn = cn;
    a.src_stride = src_row_bytes / (depth / 8);
    a.alpha = (float)alpha;
    a.border_mode = border_mode;
    for (int c = 0; c < 4; c++) a.bv[c] = border_value ? (float)border_value[c] : 0.f;
    a.acc = acc; a.dw = dw > 0 ? dw : w; a.dh = dh > 0 ? dh : h; a.acc_stride = acc_stride_floats;
    a.accumulate = accumulate; a.is_affine = is_affine; a.subpixel_bits = ctx->opt_subpixel_bits; a.tune = ctx->opt_warp_tune;
    HIP_TRY(launch_warp_accumulate(a, depth, ctx->stream));
    ctx->timing.warp_launches += 1;
    ctx->timing.warp_frames += (int64_t)n_frames;
    return STK_OK;
}

void make_warp_frame(WarpFrame& wf, const void* src, const double* M, int is_affine) { warp_frame_make(wf, src, M, is_affine); }

stk_status image_check(stk_ctx* ctx, const stk_image_f32* im, int w, int h, int c) {
    if (!im || !im->data) return fail(ctx, STK_INVALID_PARAMS, "null output image");
    if (im->width != w || im->height != h || im->channels != c) return fail(ctx, STK_INVALID_PARAMS, "output image geometry mismatch");
    if (im->row_stride_bytes && im->row_stride_bytes % 4) return fail(ctx, STK_INVALID_PARAMS, "output stride must be a multiple of 4");
    return STK_OK;
}
size_t image_stride_floats(const stk_image_f32* im) {
    return im->row_stride_bytes ? im->row_stride_bytes / 4 : (size_t)im->width * im->channels;
}

void timing_begin(stk_ctx* ctx) { std::memset(&ctx->timing, 0, sizeof(ctx->timing)); }

// ecc_match on a shard. `seeds` (n x 9 f32, row-major, h22 == 1; entry 0 unused) replaces the identity as the initial
// warp of every moving frame; `alpha` is the convertTo scale of the fold; `allow16` admits 16-bit frames (ECC then runs on
// float(grey16)) — both only for the hybrid extension (stk_hybrid_match): the reference's ecc_match is (nullptr, 1/255, false).
stk_status ecc_shard_impl(stk_ctx* ctx, const stk_frames* frames, const stk_ecc_params* params, float scale_down_width,
                          int32_t add_reference, stk_image_f32* sum, int32_t* n_added, stk_frame_stats* stats,
                          const float* seeds, double alpha, bool allow16) {
    stk_status st = check_frames(ctx, frames, true);
    if (st) return st;
    (void)hipSetDevice(ctx->device);
    EccCriteria crit{};
    if ((st = ecc_validate(ctx, params, crit))) return st;
    if (frames->depth == 16 && !allow16)  // findTransformECC accepts CV_8UC1 / CV_32FC1 only
        return fail(ctx, STK_BACKEND_ERROR, "findTransformECC: 16-bit images are not supported (8UC1 or 32FC1 only)");
    const int w = frames->width, h = frames->height, n = frames->n;
    // ecc_match_scaling_down (lib.rs:849-1028): ECC on INTER_AREA-shrunk greys, then the warp is rescaled
    const bool scaled = scale_down_width > 0;
    int ew = w, eh = h;
    if (scaled) {
        if (scale_down_width >= (float)w)   // lib.rs:876-881
            return fail(ctx, STK_INVALID_PARAMS, "scale_down_to was larger (or equal) to the full image width: full_size:" +
                                                  std::to_string(w) + ", scale_down_to:" + std::to_string(scale_down_width));
        if (scale_down_width <= 10.0f)      // lib.rs:883-888
            return fail(ctx, STK_INVALID_PARAMS, "scale_down_to was too small scale_down_to:" + std::to_string(scale_down_width));
        if (!scaled_size(w, h, scale_down_width, ew, eh)) return fail(ctx, STK_INVALID_PARAMS, "scale_down_width gives an empty image");
    }
    const int cn = frames->channels;                    // 3 (BGR) or 4 (BGRA: the output is CV_32FC4 like the reference's)
    if ((st = image_check(ctx, sum, w, h, cn))) return st;
    if (sum->location != STK_DEVICE) return fail(ctx, STK_INVALID_PARAMS, "shard sum must be device memory");
    timing_begin(ctx);

    // Host-fed stacks cross PCIe in batches on the copy stream while the frames that have arrived are already being
    // prepared (prep stream) and aligned (the ECC queue hands a slot only templates below its `ready` count).
    const size_t rb = frame_row_bytes(frames), fb = rb * (size_t)h;
    const bool host_fed = frames->location == STK_HOST;
    std::vector<const void*> dev(n);
    // whatever path leaves this function, nothing of this call may still be queued on the helper streams: a late
    // ecc_set_ready or template launch would land in the next call's queue (destroyed after `up`, i.e. after its thread joined)
    struct HelperStreamsIdle {
        stk_ctx* c;
        ~HelperStreamsIdle() { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamSynchronize(c->prep_stream); }
    } helper_streams_idle{ctx};
    AsyncUpload up;
    if (host_fed) {
        HIP_TRY(ctx->frames.reserve(fb * (size_t)n));
        for (int i = 0; i < n; i++) dev[i] = ctx->frames.as<uint8_t>() + fb * (size_t)i;
        if ((st = up.start(ctx, frames, ctx->frames.p, fb, ctx->opt_upload_batch))) return st;
    } else {
        for (int i = 0; i < n; i++) dev[i] = frames->data[i];
    }
    auto bail = [&](stk_status e) { (void)up.finish(nullptr); return e; };      // never leave the helper thread behind
    EccPlan pl{};
    if ((st = ecc_plan(ctx, ew, eh, n - 1, params->motion_type, pl))) return bail(st);
    // (scaled: the grey image and its INTER_AREA reduction keep the frames' depth, 8-bit or f32, as cvtColor and resize do)
    const int gdepth = frames->depth;
    const size_t gel = (size_t)gdepth / 8;
    if (scaled) { hipError_t he = ctx->scratch.reserve(((size_t)w * h + (size_t)ew * eh) * gel + 512); if (he != hipSuccess) return bail(fail(ctx, STK_HIP_ERROR, "scratch allocation failed")); }
    uint8_t* gfull = ctx->scratch.as<uint8_t>();
    uint8_t* gsmall = scaled ? gfull + (((size_t)w * h * gel + 255) & ~(size_t)255) : nullptr;
    // one moving frame's template on stream `s`: grey (-> scale_image) -> blur
    auto prepare_template = [&](int i, hipStream_t s) -> stk_status {
        float* t = ctx->templates.as<float>() + pl.templ_plane_stride * (size_t)(i - 1);
        if (!scaled) { HIP_TRY(launch_grey_blur(dev[i], frames->depth, cn, w, h, rb, params->gauss_filt_size, t, pl.templ_row_stride, s)); return STK_OK; }
        HIP_TRY(launch_grey(dev[i], gdepth, w, h, rb, gfull, s, 1, 0, 0, cn));
        HIP_TRY(launch_resize_area(gfull, gdepth, w, h, gsmall, ew, eh, s));
        HIP_TRY(launch_grey_blur(gsmall, gdepth, 1, ew, eh, (size_t)ew * gel, params->gauss_filt_size, t, pl.templ_row_stride, s));
        return STK_OK;
    };
    // templates of frames [first, first + count): one streaming launch for the whole run when the frames are evenly spaced
    // in memory (a tensor, or the engine's own upload buffer) and the kernel applies, else frame by frame
    auto prepare_templates = [&](int first, int count, hipStream_t s) -> stk_status {
        if (count <= 0) return STK_OK;
        bool even = !scaled && count >= 2 && ctx->opt_prep_stream;
        const ptrdiff_t step = count >= 2 ? (const char*)dev[first + 1] - (const char*)dev[first] : 0;
        for (int k = 1; even && k + 1 < count; k++) even = ((const char*)dev[first + k + 1] - (const char*)dev[first + k]) == step;
        if (even && step > 0 && cn == 3) {                          // (the streaming kernel reads 3-channel pixels; BGRA goes frame by frame)
            const hipError_t e = launch_grey_blur_batch(nullptr, dev[first], (size_t)step, count, frames->depth, w, h, rb, params->gauss_filt_size,
                                                        ctx->templates.as<float>() + pl.templ_plane_stride * (size_t)(first - 1), pl.templ_row_stride,
                                                        pl.templ_plane_stride, s);
            if (e == hipSuccess) return STK_OK;
            if (e != hipErrorNotSupported) return fail(ctx, STK_HIP_ERROR, std::string("grey_blur batch: ") + hipGetErrorString(e));
        }
        for (int k = 0; k < count; k++) { const stk_status ps = prepare_template(first + k, s); if (ps) return ps; }
        return STK_OK;
    };
    auto prepare_reference = [&](hipStream_t s) -> stk_status {
        if (!scaled) return ecc_prepare_reference(ctx, pl, dev[0], frames->depth, cn, rb, params->gauss_filt_size);
        HIP_TRY(launch_grey(dev[0], gdepth, w, h, rb, gfull, s, 1, 0, 0, cn));
        HIP_TRY(launch_resize_area(gfull, gdepth, w, h, gsmall, ew, eh, s));
        return ecc_prepare_reference(ctx, pl, gsmall, gdepth, 1, (size_t)ew * gel, params->gauss_filt_size);
    };

    HIP_TRY(hipEventRecord(ctx->ev[0], ctx->stream));
    if (host_fed && (st = up.wait_batch(0, ctx->stream))) return bail(st);
    if ((st = prepare_reference(ctx->stream))) return bail(st);
    // (with `overlap_prep` the templates are prepared while the first frames already iterate: prep_ms is then the
    // reference's share only and the templates' time is inside align_ms)
    // (shards of at most 2 x slots frames prepare all templates first: forcing the overlap there — the first launches then run
    // on the frames prepared so far — was measured on a 32-frame 4K shard, round 3: 7.84-7.92 ms per step against 7.72-7.81:
    // the late starters lengthen the tail by what the hidden preparation saves)
    const bool overlap_prep = !host_fed && !scaled && ctx->opt_prep_overlap && n - 1 > 2 * pl.n_slots;
    if (!host_fed && !overlap_prep && (st = prepare_templates(1, n - 1, ctx->stream))) return st;
    HIP_TRY(hipEventRecord(ctx->ev[1], ctx->stream));
    const float* seeds_dev = nullptr;
    if (seeds && n > 1) {
        HIP_TRY(ctx->init_warps.reserve(sizeof(float) * 9 * (size_t)(n - 1)));
        HIP_TRY(hipMemcpyAsync(ctx->init_warps.p, seeds + 9, sizeof(float) * 9 * (size_t)(n - 1), hipMemcpyHostToDevice, ctx->stream));
        seeds_dev = ctx->init_warps.as<float>();
    }
    std::vector<EccFrameResult> res;
    // The fold follows the alignment in stream order, its frame table built on the device from the ECC results: the moment
    // the host learns that the queue has drained it enqueues table + fold behind the launches still in flight, instead of
    // waiting for the results, inverting 3x3 matrices and sending them back (118 us of idle GPU per 32-frame 4K shard).
    // The results still come to the host afterwards: statistics, and the reference's `?` on a failed frame.
    const int is_affine = params->motion_type != STK_MOTION_HOMOGRAPHY;
    bool folded_on_device = false;
    const EccDone fold_now = [&]() -> stk_status {
        HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));
        const int n_out = (n - 1) + (add_reference ? 1 : 0);
        if (n_out > 0) {
            HIP_TRY(ctx->warpframes.reserve(sizeof(WarpFrame) * (size_t)n_out));
            HIP_TRY(launch_warp_frames_from_ecc(ctx->results.as<EccFrameResult>(), ctx->frameptrs.as<const void*>(), n - 1, add_reference ? 1 : 0,
                                                is_affine, w, h, rb, ctx->warpframes.as<WarpFrame>(), ctx->stream));
            const stk_status fs = warp_fold_enqueue(ctx, n_out, frames->depth, w, h, cn, rb, alpha, STK_BORDER_CONSTANT, nullptr, is_affine,
                                                    sum->data, image_stride_floats(sum), 0);
            if (fs) return fs;
        } else {
            HIP_TRY(hipMemsetAsync(sum->data, 0, image_stride_floats(sum) * h * sizeof(float), ctx->stream));
        }
        HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
        folded_on_device = true;
        return STK_OK;
    };
    // (the scale-down variant rescales the warps on the host first, and a run without iterations has no device results)
    const bool device_fold = !scaled && !(crit.n_iter < 1) && std::fabs(-1.0 - (-crit.eps)) >= crit.eps && n > 1;
    const EccDone* on_done = device_fold ? &fold_now : nullptr;
    if (device_fold) {
        HIP_TRY(ctx->frameptrs.reserve(sizeof(void*) * (size_t)n));
        HIP_TRY(hipMemcpyAsync(ctx->frameptrs.p, dev.data(), sizeof(void*) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    }
    if (host_fed) {
        // the scaled path shares one grey scratch between the streams, so its templates are prepared on the compute
        // stream itself (still batch by batch as they arrive); the full-size path uses the prep stream
        hipStream_t ps = scaled ? ctx->stream : ctx->prep_stream;
        int next_batch = 1, enq = 0;
        EccQueue* q = ctx->queue.as<EccQueue>();
        const EccFeed feed = [&](bool block, int* enqueued) -> stk_status {
            while (next_batch < up.batches() && (block || up.recorded() > next_batch)) {
                stk_status fs = up.wait_batch(next_batch, ps);
                if (fs) return fs;
                if ((fs = prepare_templates(up.batch_first(next_batch), up.batch_count(next_batch), ps))) return fs;
                enq = up.batch_first(next_batch) + up.batch_count(next_batch) - 1;        // templates 0 .. enq-1 exist
                HIP_TRY(launch_ecc_set_ready(q, enq, ps));
                next_batch++;
                block = false;
            }
            *enqueued = enq;
            return STK_OK;
        };
        st = ecc_run(ctx, pl, crit, seeds_dev, res, &feed, on_done);
        double h2d = 0;
        const stk_status fin = up.finish(&h2d);
        if (st) return st;
        if (fin) return fin;
        ctx->timing.h2d_ms = h2d; ctx->timing.h2d_bytes = (int64_t)(fb * (size_t)n);
        HIP_TRY(hipStreamSynchronize(ctx->prep_stream));
    } else if (overlap_prep) {
        // device-resident stack: the templates are prepared on the prep stream, 16 frames at a time, each run raising the
        // queue's `ready` mark — the HBM-bound preparation runs under the VALU-bound iteration of the frames before it
        EccQueue* q = ctx->queue.as<EccQueue>();
        bool enqueued_all = false;
        const EccFeed feed = [&](bool, int* enqueued) -> stk_status {
            if (!enqueued_all) {
                for (int first = 1; first < n; first += 16) {
                    const int cnt = std::min(16, n - first);
                    const stk_status fs = prepare_templates(first, cnt, ctx->prep_stream);
                    if (fs) return fs;
                    HIP_TRY(launch_ecc_set_ready(q, first + cnt - 1, ctx->prep_stream));
                }
                enqueued_all = true;
            }
            *enqueued = n - 1;                                  // nothing further depends on the host
            return STK_OK;
        };
        st = ecc_run(ctx, pl, crit, seeds_dev, res, &feed, on_done);
        HIP_TRY(hipStreamSynchronize(ctx->prep_stream));
        if (st) return st;
    } else if ((st = ecc_run(ctx, pl, crit, seeds_dev, res, nullptr, on_done))) return st;
    if (!folded_on_device) HIP_TRY(hipEventRecord(ctx->ev[2], ctx->stream));

    int first_err = -1;
    if (stats) {
        std::memset(stats, 0, sizeof(stk_frame_stats) * n);
        stats[0].warp[0] = stats[0].warp[4] = stats[0].warp[8] = 1;
        stats[0].rho = 1;
    }
    for (int i = 1; i < n; i++) {
        EccFrameResult& e = res[i - 1];
        if (scaled && !e.status) {
            if (is_affine) {                               // lib.rs:941-951: only the translation column
                e.warp[2] *= (float)w / (float)ew;
                e.warp[5] *= (float)h / (float)eh;
            } else {                                       // adjust_homography_for_scale_f32, utils.rs:229-239
                const double sx = (double)w / (double)ew, sy = (double)h / (double)eh;
                e.warp[2] *= (float)sx; e.warp[5] *= (float)sy; e.warp[6] /= (float)sx; e.warp[7] /= (float)sy;
            }
        }
        if (stats) {
            stats[i].status = e.status ? 2 : 0; stats[i].iterations = e.iters; stats[i].rho = e.rho;
            for (int k = 0; k < 9; k++) stats[i].warp[k] = e.warp[k];
            if (is_affine) { stats[i].warp[6] = 0; stats[i].warp[7] = 0; stats[i].warp[8] = 1; }
        }
        if (e.status && first_err < 0) first_err = i;
    }
    if (first_err >= 0 && folded_on_device) (void)hipStreamSynchronize(ctx->stream);     // the fold behind the queue still writes `sum`
    if (first_err >= 0)  // `?` at lib.rs:777: any OpenCV error aborts the whole stack
        return fail(ctx, STK_BACKEND_ERROR, std::string(ecc_status_message(res[first_err - 1].status)) + " [frame " + std::to_string(first_err) + "]");

    if (folded_on_device) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->timing.prep_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
        ctx->timing.align_ms = ev_ms(ctx->ev[1], ctx->ev[2]);
        ctx->timing.warp_ms = ev_ms(ctx->ev[2], ctx->ev[3]);
        if (n_added) *n_added = (int32_t)((n - 1) + (add_reference ? 1 : 0));
        return STK_OK;
    }
    std::vector<WarpFrame> wf;
    wf.reserve(n);
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (add_reference) { wf.emplace_back(); make_warp_frame(wf.back(), dev[0], I3, is_affine); }
    for (int i = 1; i < n; i++) {
        double M[9];
        for (int k = 0; k < 9; k++) M[k] = res[i - 1].warp[k];
        if (is_affine) { M[6] = 0; M[7] = 0; M[8] = 1; }
        wf.emplace_back();
        make_warp_frame(wf.back(), dev[i], M, is_affine);
    }
    if (wf.empty()) HIP_TRY(hipMemsetAsync(sum->data, 0, image_stride_floats(sum) * h * sizeof(float), ctx->stream));
    if ((st = warp_fold(ctx, wf, frames->depth, w, h, cn, rb, alpha, STK_BORDER_CONSTANT, nullptr, is_affine,
                        sum->data, image_stride_floats(sum), 0))) return st;
    HIP_TRY(hipEventRecord(ctx->ev[3], ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->timing.prep_ms = ev_ms(ctx->ev[0], ctx->ev[1]);
    ctx->timing.align_ms = ev_ms(ctx->ev[1], ctx->ev[2]);
    ctx->timing.warp_ms = ev_ms(ctx->ev[2], ctx->ev[3]);
    if (n_added) *n_added = (int32_t)wf.size();
    return STK_OK;
}

extern "C" {

stk_status stk_ecc_match_shard(stk_ctx* ctx, const stk_frames* frames, const stk_ecc_params* params,
                               float scale_down_width, int32_t add_reference, stk_image_f32* sum,
                               int32_t* n_added, stk_frame_stats* stats) {
    return ecc_shard_impl(ctx, frames, params, scale_down_width, add_reference, sum, n_added, stats, nullptr, 1.0 / 255.0, false);
}

stk_status stk_finalize_mean(stk_ctx* ctx, const stk_image_f32* sum, int64_t n_frames, stk_image_f32* out) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (!sum || !out || !sum->data || !out->data) return fail(ctx, STK_INVALID_PARAMS, "null image");
    if (n_frames <= 0) return fail(ctx, STK_INVALID_PARAMS, "All images discarded");   // lib.rs:324
    if (sum->location != STK_DEVICE) return fail(ctx, STK_INVALID_PARAMS, "sum must be device memory");
    if (sum->row_stride_bytes || out->row_stride_bytes) return fail(ctx, STK_INVALID_PARAMS, "finalize expects tightly packed images");
    (void)hipSetDevice(ctx->device);
    const size_t n = (size_t)sum->width * sum->height * sum->channels;
    const float sc = (float)(1.0 / (double)n_frames);   // MatExpr A / s == A.convertTo(-1, 1/s)
    HIP_TRY(hipEventRecord(ctx->ev[4], ctx->stream));
    if (out->location == STK_DEVICE) {
        HIP_TRY(launch_scale(sum->data, out->data, n, sc, ctx->stream));
    } else {
        HIP_TRY(ctx->scratch.reserve(n * sizeof(float)));
        HIP_TRY(launch_scale(sum->data, ctx->scratch.as<float>(), n, sc, ctx->stream));
        HIP_TRY(hipMemcpyAsync(out->data, ctx->scratch.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipEventRecord(ctx->ev[5], ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->timing.finalize_ms = ev_ms(ctx->ev[4], ctx->ev[5]);
    return STK_OK;
}

stk_status stk_ecc_match(stk_ctx* ctx, const stk_frames* frames, const stk_ecc_params* params,
                         float scale_down_width, stk_image_f32* out, stk_frame_stats* stats) {
    if (ctx && ctx->multi) return multi_match(ctx, 0, frames, nullptr, params, scale_down_width, out, nullptr, stats);
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
    int32_t added = 0;
    if ((st = stk_ecc_match_shard(ctx, frames, params, scale_down_width, 1, &sum, &added, stats))) return st;
    const stk_timing keep = ctx->timing;
    st = stk_finalize_mean(ctx, &sum, frames->n, out);     // lib.rs:836-839: divide by files_vec.len()
    const double fin = ctx->timing.finalize_ms;
    ctx->timing = keep; ctx->timing.finalize_ms = fin;
    return st;
}

// ---- stage-level entry points ------------------------------------------------------------------
stk_status stk_grey(stk_ctx* ctx, const stk_frames* f, void* out) {
    stk_status st = check_frames(ctx, f, true);
    if (st) return st;
    if (!out) return fail(ctx, STK_INVALID_PARAMS, "null output");
    (void)hipSetDevice(ctx->device);
    std::vector<const void*> dev;
    if ((st = resolve_frames(ctx, f, dev))) return st;
    const size_t ob = (size_t)f->width * f->height * (f->depth / 8);
    void* d = out;
    if (f->location == STK_HOST) { HIP_TRY(ctx->scratch.reserve(ob)); d = ctx->scratch.p; }
    HIP_TRY(launch_grey(dev[0], f->depth, f->width, f->height, frame_row_bytes(f), d, ctx->stream, 1, 0, 0, f->channels));
    if (f->location == STK_HOST) HIP_TRY(hipMemcpyAsync(out, d, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STK_OK;
}

stk_status stk_convert_f32(stk_ctx* ctx, const stk_frames* f, double alpha, float* out) {
    stk_status st = check_frames(ctx, f, false);
    if (st) return st;
    if (!out) return fail(ctx, STK_INVALID_PARAMS, "null output");
    if (f->row_stride_bytes && f->row_stride_bytes != (size_t)f->width * f->channels * (f->depth / 8))
        return fail(ctx, STK_INVALID_PARAMS, "convert expects tightly packed frames");
    (void)hipSetDevice(ctx->device);
    std::vector<const void*> dev;
    if ((st = resolve_frames(ctx, f, dev))) return st;
    const size_t n = (size_t)f->width * f->height * f->channels;
    float* d = out;
    if (f->location == STK_HOST) { HIP_TRY(ctx->scratch.reserve(n * 4)); d = ctx->scratch.as<float>(); }
    HIP_TRY(launch_convert_f32(dev[0], f->depth, n, (float)alpha, d, ctx->stream));
    if (f->location == STK_HOST) HIP_TRY(hipMemcpyAsync(out, d, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STK_OK;
}

stk_status stk_gaussian_blur_f32(stk_ctx* ctx, const void* grey, int32_t depth, int32_t width, int32_t height,
                                 int32_t location, int32_t ksize, float* out) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (!grey || !out || width <= 0 || height <= 0) return fail(ctx, STK_INVALID_PARAMS, "bad arguments");
    if (depth != 8 && depth != 32) return fail(ctx, STK_INVALID_PARAMS, "blur input must be u8 or f32");
    if (ksize <= 0 || ksize % 2 == 0) return fail(ctx, STK_BACKEND_ERROR, "GaussianBlur: kernel size must be odd and positive");
    if (ksize > 63) return fail(ctx, STK_NOT_IMPLEMENTED, "ksize > 63 is not supported");
    (void)hipSetDevice(ctx->device);
    const size_t ib = (size_t)width * height * (depth / 8), ob = (size_t)width * height * 4;
    const void* src = grey; float* dst = out;
    if (location == STK_HOST) {
        HIP_TRY(ctx->frames.reserve(ib)); HIP_TRY(ctx->scratch.reserve(ob));
        HIP_TRY(hipMemcpyAsync(ctx->frames.p, grey, ib, hipMemcpyHostToDevice, ctx->stream));
        src = ctx->frames.p; dst = ctx->scratch.as<float>();
    }
    HIP_TRY(launch_grey_blur(src, depth, 1, width, height, (size_t)width * (depth / 8), ksize, dst, width, ctx->stream));
    if (location == STK_HOST) HIP_TRY(hipMemcpyAsync(out, dst, ob, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STK_OK;
}

stk_status stk_grey_blur_f32(stk_ctx* ctx, const stk_frames* f, int32_t ksize, float* out) {
    stk_status st = check_frames(ctx, f, true);
    if (st) return st;
    if (!out) return fail(ctx, STK_INVALID_PARAMS, "null output");
    // (16-bit frames: only the hybrid extension runs ECC on them, on float(grey16); the stage is exposed for its tests)
    if (ksize <= 0 || ksize % 2 == 0) return fail(ctx, STK_BACKEND_ERROR, "GaussianBlur: kernel size must be odd and positive");
    if (ksize > 63) return fail(ctx, STK_NOT_IMPLEMENTED, "ksize > 63 is not supported");
    (void)hipSetDevice(ctx->device);
    std::vector<const void*> dev;
    if ((st = resolve_frames(ctx, f, dev))) return st;
    const int stride = (f->width + 3) & ~3;                   // the engine's template row stride
    const size_t ob = (size_t)stride * f->height * 4;
    HIP_TRY(ctx->blur_tmp.reserve(ob));
    HIP_TRY(launch_grey_blur(dev[0], f->depth, f->channels, f->width, f->height, frame_row_bytes(f), ksize,
                             ctx->blur_tmp.as<float>(), stride, ctx->stream));
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)f->width * 4, ctx->blur_tmp.p, (size_t)stride * 4, (size_t)f->width * 4, f->height,
                             f->location == STK_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STK_OK;
}

stk_status stk_sharpness(stk_ctx* ctx, const void* grey, int32_t depth, int32_t width, int32_t height, int32_t location,
                         int32_t metric, int32_t ksize, double* out) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (!grey || !out || width <= 0 || height <= 0) return fail(ctx, STK_INVALID_PARAMS, "bad arguments");
    if (depth != 8 && depth != 32) return fail(ctx, STK_INVALID_PARAMS, "sharpness input must be 8-bit or f32, single channel");
    if (metric < STK_SHARPNESS_LAPM || metric > STK_SHARPNESS_GLVN) return fail(ctx, STK_INVALID_PARAMS, "unknown sharpness metric");
    if (metric == STK_SHARPNESS_TENG && ksize != 1 && ksize != 3 && ksize != 5 && ksize != 7)
        return fail(ctx, STK_INVALID_PARAMS, "Kernel size must be 1, 3, 5, or 7");                    // lib.rs:1105-1109
    (void)hipSetDevice(ctx->device);
    const size_t ib = (size_t)width * height * (depth / 8);
    const void* src = grey;
    if (location == STK_HOST) {
        HIP_TRY(ctx->frames.reserve(ib));
        HIP_TRY(hipMemcpyAsync(ctx->frames.p, grey, ib, hipMemcpyHostToDevice, ctx->stream));
        src = ctx->frames.p;
    }
    const int nb = (int)std::max<size_t>(1, std::min<size_t>(1024, ((size_t)width * height + 255) / 256));
    HIP_TRY(ctx->scratch.reserve((size_t)nb * 16));
    HIP_TRY(launch_sharpness(src, depth, width, height, metric, ksize, ctx->scratch.p, nb, ctx->stream));
    std::vector<unsigned char> hostp((size_t)nb * 16);
    HIP_TRY(hipMemcpyAsync(hostp.data(), ctx->scratch.p, hostp.size(), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // partials added in index order; 8-bit input: exact integers (LAPM carries a factor 4)
    double s = 0, sq = 0;
    if (depth == 8) {
        const long long* p = reinterpret_cast<const long long*>(hostp.data());
        long long is = 0, isq = 0;
        for (int b = 0; b < nb; b++) { is += p[2 * b]; isq += p[2 * b + 1]; }
        s = (double)is; sq = (double)isq;
        if (metric == STK_SHARPNESS_LAPM) s *= 0.25;
    } else {
        const double* p = reinterpret_cast<const double*>(hostp.data());
        for (int b = 0; b < nb; b++) { s += p[2 * b]; sq += p[2 * b + 1]; }
    }
    const double scale = 1. / ((double)width * height);          // cv::mean / meanStdDev multiply by the reciprocal
    if (metric == STK_SHARPNESS_LAPM || metric == STK_SHARPNESS_TENG) *out = s * scale;
    else {
        const double mean = s * scale;
        const double sigma = std::sqrt(std::max(sq * scale - mean * mean, 0.));
        *out = metric == STK_SHARPNESS_LAPV ? sigma * sigma : (sigma * sigma) / std::max(mean, DBL_EPSILON);
    }
    return STK_OK;
}

stk_status stk_find_transform_ecc(stk_ctx* ctx, const void* templ, const void* input, int32_t depth, int32_t width,
                                  int32_t height, int32_t location, const stk_ecc_params* params, float* warp,
                                  double* rho, int32_t* iterations) {
    if (!ctx) return STK_INVALID_PARAMS;
    if (!templ || !input || !warp || width <= 0 || height <= 0) return fail(ctx, STK_INVALID_PARAMS, "bad arguments");
    if (depth != 8 && depth != 32) return fail(ctx, STK_BACKEND_ERROR, "findTransformECC: images must be 8UC1 or 32FC1");
    EccCriteria crit{};
    stk_status st = ecc_validate(ctx, params, crit);
    if (st) return st;
    (void)hipSetDevice(ctx->device);
    timing_begin(ctx);
    const size_t ib = (size_t)width * height * (depth / 8);
    const void* t = templ; const void* in = input;
    if (location == STK_HOST) {
        HIP_TRY(ctx->frames.reserve(2 * ib));
        HIP_TRY(hipMemcpyAsync(ctx->frames.p, templ, ib, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipMemcpyAsync(ctx->frames.as<uint8_t>() + ib, input, ib, hipMemcpyHostToDevice, ctx->stream));
        t = ctx->frames.p; in = ctx->frames.as<uint8_t>() + ib;
    }
    EccPlan pl{};
    // variant 3 assumes m22 == 1 (true for every warp findTransformECC itself produces); a caller-supplied
    // initial warp with another m22 takes the general kernel
    struct VariantGuard {                                   // restores the option on every exit path
        stk_ctx* c; int saved;
        ~VariantGuard() { c->opt_ecc_variant = saved; }
    } guard{ctx, ctx->opt_ecc_variant};
    if (guard.saved != 0 && params->motion_type == STK_MOTION_HOMOGRAPHY && warp[8] != 1.0f) ctx->opt_ecc_variant = 0;
    st = ecc_plan(ctx, width, height, 1, params->motion_type, pl);
    if (st) return st;
    const size_t rb = (size_t)width * (depth / 8);
    if ((st = ecc_prepare_reference(ctx, pl, in, depth, 1, rb, params->gauss_filt_size))) return st;
    HIP_TRY(launch_grey_blur(t, depth, 1, width, height, rb, params->gauss_filt_size, ctx->templates.as<float>(), pl.templ_row_stride, ctx->stream));
    float w9[9];
    for (int k = 0; k < 9; k++) w9[k] = warp[k];
    if (params->motion_type != STK_MOTION_HOMOGRAPHY) { w9[6] = 0; w9[7] = 0; w9[8] = 1; }
    HIP_TRY(ctx->init_warps.reserve(sizeof(w9)));
    HIP_TRY(hipMemcpyAsync(ctx->init_warps.p, w9, sizeof(w9), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<EccFrameResult> res;
    st = ecc_run(ctx, pl, crit, ctx->init_warps.as<float>(), res);
    if (st) return st;
    if (crit.n_iter >= 1) for (int k = 0; k < 9; k++) warp[k] = res[0].warp[k];
    if (rho) *rho = res[0].rho;
    if (iterations) *iterations = res[0].iters;
    if (res[0].status) return fail(ctx, STK_BACKEND_ERROR, ecc_status_message(res[0].status));
    return STK_OK;
}

stk_status stk_warp_accumulate(stk_ctx* ctx, const stk_frames* f, const double* M, int32_t is_affine, int32_t border_mode,
                               const double* border_value, double alpha, int32_t accumulate, stk_image_f32* acc) {
    stk_status st = check_frames(ctx, f, false);
    if (st) return st;
    if (!M) return fail(ctx, STK_INVALID_PARAMS, "null matrix");
    if (border_mode < 0 || border_mode > 4)
        return fail(ctx, border_mode == STK_BORDER_TRANSPARENT ? STK_NOT_IMPLEMENTED : STK_INVALID_PARAMS,
                    "border mode not supported (BORDER_TRANSPARENT leaves the reference's output uninitialised)");
    if ((st = image_check(ctx, acc, f->width, f->height, f->channels))) return st;
    (void)hipSetDevice(ctx->device);
    std::vector<const void*> dev;
    if ((st = resolve_frames(ctx, f, dev))) return st;
    const size_t nel = (size_t)f->width * f->height * f->channels;
    float* d = acc->data; size_t stride = image_stride_floats(acc);
    if (acc->location == STK_HOST) {
        if (acc->row_stride_bytes) return fail(ctx, STK_INVALID_PARAMS, "host accumulator must be tightly packed");
        HIP_TRY(ctx->acc.reserve(nel * 4));
        d = ctx->acc.as<float>();
        if (accumulate) HIP_TRY(hipMemcpyAsync(d, acc->data, nel * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    std::vector<WarpFrame> wf(1);
    make_warp_frame(wf[0], dev[0], M, is_affine);
    if ((st = warp_fold(ctx, wf, f->depth, f->width, f->height, f->channels, frame_row_bytes(f), alpha, border_mode,
                        border_value, is_affine, d, stride, accumulate))) return st;
    if (acc->location == STK_HOST) HIP_TRY(hipMemcpyAsync(acc->data, d, nel * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return STK_OK;
}

}  // extern "C"
