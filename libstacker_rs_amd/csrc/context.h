// context.h — the engine's per-GPU context (stk_ctx) and host helpers shared by stacker.cpp and keypoint.cpp.
#pragma once
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "common.h"
#include "homography.h"
#include "keypoint.h"

// ---------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + (bytes >> 3);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T* as() const { return (T*)p; }
};

namespace stk { struct MultiState; }
void multi_destroy(stk_ctx* ctx);
stk_status multi_match(stk_ctx* ctx, int kind, const stk_frames* frames, const stk_keypoint_params* kp, const stk_ecc_params* ep,
                       float scale_down_width, stk_image_f32* out, int32_t* dropped_out, stk_frame_stats* stats);
stk_status set_option_one(stk_ctx* ctx, const char* name, int64_t value);
int multi_member_count(const stk_ctx* ctx);
stk_ctx* multi_member(const stk_ctx* ctx, int i);

// Frames that are still being produced (decoded) while the engine already runs: the uploader asks the gate before it
// copies a frame. `wait` blocks until the frame at `ptr` is complete and returns false if it never will be.
struct FrameGate { std::function<bool(const void* ptr)> wait; };

constexpr int STK_MAX_KP_LANES = 8;

struct stk_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t copy_stream = nullptr;    // host -> HBM copies of host-fed stacks (upload.cpp)
    hipStream_t prep_stream = nullptr;    // grey + blur of frames that arrive while the ECC queue is already running
    hipStream_t tail_stream = nullptr;    // highest priority: the keypoint lanes' descriptor / 2-NN / homography launches (keypoint.cpp)
    hipStream_t ecc_stream2 = nullptr;    // second launch sequence of the ECC queue (option ecc_groups = 2)
    std::vector<hipEvent_t> upload_events;
    hipEvent_t gate_ev = nullptr, gate_ev2 = nullptr;
    const FrameGate* frame_gate = nullptr;   // set by the path-based entry points for the duration of one call
    int opt_warp_tune = 0;
    int opt_prep_overlap = 1;             // ecc_match on a device-resident stack: templates prepared on the prep stream while the first frames iterate
    int opt_prep_stream = 1;              // 1: templates of a run of frames by the streaming grey+blur kernel in one launch; 0: tiled kernel, frame by frame
    int opt_upload_batch = 8;             // frames per host -> HBM batch
    std::string err;
    int opt_ecc_slots = 0;        // 0 = auto
    int opt_subpixel_bits = 0;
    int opt_profile = 1;
    int opt_ecc_chunk = 0;        // (iterate, solve) pairs between two polls of the completion counter; 0 = by frame size (stacker.cpp: ecc_run)
    int opt_profile_stride = 1;   // profile = 2: bracket every n-th ECC pixel pass with an event pair
    bool opt_orb_resize_tables = true; // ORB pyramid steps by the table-driven kernel (false: tables computed per tile, round 2's kernel; same bits)
    bool opt_kp_tail_priority = true; // keypoint lanes: descriptor / 2-NN / homography launches on the highest-priority stream
    bool opt_orb_device_cull = true; // ORB: Harris cull and ordering of the short lists on the device (false: on the host pool); same keypoints
    bool opt_orb_patch_blur = true;  // ORB: the descriptor kernel blurs the window it reads (false: blur every level whole, then sample)
    int opt_kp_lanes = 3;         // keypoint path on device-resident stacks: the stack is cut into this many runs of frames that go through the pipeline side by side (helper contexts), 1 = one pipeline
    int opt_kp_workers = 12;      // host threads for the per-frame host steps of the keypoint path (Harris cull, RANSAC)
    int opt_ecc_blocks = 0;       // total workgroups of one ECC iteration launch; 0 = 288 per frame in flight (see ecc_plan)
    int opt_ecc_ring = 1;         // column-walking ECC pass: frame-0 rows through the per-wave LDS ring (0: always gather from global memory)
    int opt_ecc_groups = 0;       // ECC slots in this many groups with their own launch sequences on two streams (stacker.cpp: ecc_run); 0 = by frame size
    int opt_ecc_ring_lookahead = 5;   // debug: frame-0 rows the ring keeps ahead (5 = production; less makes the run-time check fire and the strip fall back)
    int opt_ecc_variant = 3;      // ECC iteration kernel: 3 = production (column-walking homography pass / pipelined affine family), 0 = direct cross-check
    stk_timing timing{};
    hipEvent_t ev[8] = {};
    hipEvent_t poll_ev[2] = {};
    int* host_done = nullptr;     // pinned, 16 ints: [0], [1] completion counters of the two chunks in flight, [2] ring fall-back count
    const void* ref_zeroed_ptr = nullptr;   // the frame-0 planes' zero border exists for this buffer and geometry (ecc_prepare_reference)
    int ref_zeroed_w = 0, ref_zeroed_h = 0;
    std::vector<hipEvent_t> prof_ev;   // event pairs for per-launch timing (option profile = 2)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> fold_ev;   // event pairs around the keypoint path's fold launches (grow-only pool)
    // page-locked host block the path-based entry points decode a stack into (imread.cpp: match_files); grow-only, like the
    // device workspaces: locking 6.4 GB of pages for a 256-frame 4K stack costs more than decoding into them
    unsigned char* files_block = nullptr; size_t files_block_cap = 0; bool files_block_pinned = false;
    // workspace
    DevBuf frames, ref, blur_tmp, templates, slots, queue, results, partials, warpframes, acc, scratch, init_warps, frameptrs;
    stk::KeypointWorkspace* kp = nullptr;
    stk::geom::HgWorkspace* hg = nullptr;   // findHomography batch workspace (homography.cpp)
    stk::HostPool* host_pool = nullptr;
    stk::HostPool* shared_pool = nullptr;  // not owned: the pool every member of a multi-device context shares (multi.cpp); overrides host_pool
    stk_ctx* lanes[STK_MAX_KP_LANES - 1] = {};   // hidden helper contexts of the same device: lanes 1.. of keypoint_align_impl (keypoint.cpp)

    stk::MultiState* multi = nullptr;      // non-null: this context spans several devices (multi.cpp); it is member 0 itself   // persistent host threads of the keypoint path (keypoint.cpp)
    std::mutex err_mutex;
};

inline stk_status fail(stk_ctx* ctx, stk_status st, const std::string& msg) {
    if (!ctx) return st;
    std::lock_guard<std::mutex> lock(ctx->err_mutex);   // keypoint workers may fail concurrently
    ctx->err = msg;
    return st;
}
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, STK_HIP_ERROR, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)


using stk::WarpFrame;
size_t frame_row_bytes(const stk_frames* f);
stk_status resolve_frames(stk_ctx* ctx, const stk_frames* f, std::vector<const void*>& dev);
stk_status check_frames(stk_ctx* ctx, const stk_frames* f, bool need_bgr);
// (w, h): the SOURCE frames' size; (dw, dh): the accumulator's, 0 = the same (a stack of one geometry)
stk_status warp_fold(stk_ctx* ctx, std::vector<WarpFrame>& wf, int depth, int w, int h, int cn,
                     size_t src_row_bytes, double alpha, int border_mode, const double* border_value,
                     int is_affine, float* acc, size_t acc_stride_floats, int accumulate, int dw = 0, int dh = 0);
stk_status warp_fold_enqueue(stk_ctx* ctx, int n_frames, int depth, int w, int h, int cn, size_t src_row_bytes, double alpha,
                             int border_mode, const double* border_value, int is_affine, float* acc, size_t acc_stride_floats,
                             int accumulate, int first_frame = 0, int dw = 0, int dh = 0);
void make_warp_frame(WarpFrame& wf, const void* src, const double* M, int is_affine);
stk_status image_check(stk_ctx* ctx, const stk_image_f32* im, int w, int h, int c);
size_t image_stride_floats(const stk_image_f32* im);
void timing_begin(stk_ctx* ctx);
float ev_ms(hipEvent_t a, hipEvent_t b);

// shared internals of the entry points (stacker.cpp / keypoint.cpp / hybrid.cpp)
stk_status ecc_shard_impl(stk_ctx* ctx, const stk_frames* frames, const stk_ecc_params* params, float scale_down_width,
                          int32_t add_reference, stk_image_f32* sum, int32_t* n_added, stk_frame_stats* stats,
                          const float* seeds, double alpha, bool allow16);
// called with consecutive frame ranges [lo, hi) in increasing order, each once, as soon as the results of those frames (and
// of every frame before them) are final — possibly from a helper lane's thread, never from two threads at once
using KpFramesFinal = std::function<stk_status(int lo, int hi)>;
struct KpAlign { bool ok = false; double H[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; int n_keypoints = 0, n_matches = 0, n_inliers = 0; };
// keypoint_match's alignment half: ORB + 2-NN + ratio / sort / truncate + homography per moving frame (no fold).
// reduce16: 16-bit frames are matched on their 8-bit reduction (g + 128) / 257 (hybrid extension only).
stk_status keypoint_align_impl(stk_ctx* ctx, const stk_frames* frames, const stk_keypoint_params* params, float scale_down_width,
                               bool reduce16, std::vector<KpAlign>& out, int* n_ref_keypoints, std::vector<const void*>& dev,
                               const KpFramesFinal* on_final = nullptr);
