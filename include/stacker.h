/*
 * stacker.h — C ABI of the MI355X-native align-and-stack engine.
 *
 * This is the drop-in boundary for the hot path of eadf/libstacker.rs:
 * everything a Rust `libstacker`-compatible shim needs to bind (see
 * INTEGRATION.md for the `extern "C"` block a maintainer would add).
 * Plain pointers and sizes only; no C++ / torch types cross this line.
 *
 * Conventions
 *  - Images are interleaved, row-major, BGR channel order (OpenCV `Mat`
 *    layout, which is what the reference hands around: utils.rs:128-144).
 *  - `location` says whether a pointer is host (0) or device/HBM (1) memory.
 *  - Every entry point returns an `stk_status`; the message of the last
 *    failure is available from stk_last_error(). Nothing throws or aborts.
 *  - A `stk_ctx` owns one GPU, one HIP stream and the HBM workspace — or, made
 *    with stk_create_multi, several GPUs of the node (one host thread, stream
 *    and workspace per GPU, RCCL communicators for the accumulator reduce);
 *    one call at a time per context, any number of contexts per process.
 *
 * Reference citations are `file:line` under the reference checkout
 * (`src/lib.rs`, `src/utils.rs`).
 */
#ifndef STACKER_H
#define STACKER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: mirror StackerError (lib.rs:27-45) ------------------ */
typedef enum {
    STK_OK = 0,
    STK_NOT_ENOUGH_FILES = 1, /* StackerError::NotEnoughFiles  lib.rs:31,155,725 */
    STK_INVALID_PARAMS = 2,   /* StackerError::InvalidParams   lib.rs:41,324,377,876,883 */
    STK_PROCESSING_ERROR = 3, /* StackerError::ProcessingError lib.rs:43,347,841 */
    STK_BACKEND_ERROR = 4,    /* StackerError::OpenCvError     lib.rs:30 (ECC no-conv/NaN, bad types) */
    STK_IO_ERROR = 5,         /* StackerError::IoError         lib.rs:36 */
    STK_HIP_ERROR = 6,        /* no reference analogue: HIP runtime failure */
    STK_NOT_IMPLEMENTED = 7   /* StackerError::NotImplemented  lib.rs:33 */
} stk_status;

/* ---- constants the reference passes through from OpenCV ---------------- */
enum { STK_MOTION_TRANSLATION = 0, STK_MOTION_EUCLIDEAN = 1, /* MotionType lib.rs:603-609 */
       STK_MOTION_AFFINE = 2, STK_MOTION_HOMOGRAPHY = 3 };
enum { STK_METHOD_LEAST_SQUARES = 0, STK_METHOD_LMEDS = 4, /* KeyPointMatchParameters::method lib.rs:51 */
       STK_METHOD_RANSAC = 8, STK_METHOD_RHO = 16 };
/* (RHO and OpenCV's USAC numbers 32 .. 38: STK_NOT_IMPLEMENTED. Any other value: findHomography throws, stk_find_homography
 * reports STK_BACKEND_ERROR, and the whole-stack calls skip every moving frame as lib.rs:275 does.) */
enum { STK_BORDER_CONSTANT = 0, STK_BORDER_REPLICATE = 1, STK_BORDER_REFLECT = 2,
       STK_BORDER_WRAP = 3, STK_BORDER_REFLECT_101 = 4, STK_BORDER_TRANSPARENT = 5 };
enum { STK_HOST = 0, STK_DEVICE = 1 };
enum { STK_DEPTH_U8 = 8, STK_DEPTH_U16 = 16, STK_DEPTH_F32 = 32 };

/* ---- parameter mirrors -------------------------------------------------- */
/* KeyPointMatchParameters, lib.rs:48-73; defaults utils.rs:250-261. */
typedef struct {
    int32_t method;                 /* STK_METHOD_* */
    double  ransac_reproj_threshold;
    float   match_keep_ratio;
    float   match_ratio;
    int32_t border_mode;            /* STK_BORDER_* */
    double  border_value[4];        /* core::Scalar */
} stk_keypoint_params;

/* EccMatchParameters, lib.rs:611-623; Option<T> -> has_* flag + value
 * (TermCriteria mapping utils.rs:159-170). */
typedef struct {
    int32_t motion_type;            /* STK_MOTION_* */
    int32_t has_max_count;
    int32_t max_count;
    int32_t has_epsilon;
    double  epsilon;
    int32_t gauss_filt_size;        /* odd, 1 .. 63 (larger: STK_NOT_IMPLEMENTED) */
} stk_ecc_params;

/* A stack of decoded frames (what read_grey_and_f32's imread produced,
 * utils.rs:132): data[0] is the reference frame. All frames share geometry. */
typedef struct {
    const void* const* data;        /* n pointers */
    int32_t n;
    int32_t width, height;
    int32_t channels;               /* 3 (BGR) or 4 (BGRA, what imread(IMREAD_UNCHANGED) makes of a PNG with alpha: grey comes
                                       from B, G, R — cvtColor ignores the fourth channel, utils.rs:136-142 — and the stacked image
                                       has four channels too, the fourth being the aligned, averaged alpha / 255, exactly what the
                                       reference's convertTo / warp / add do to it); 1 is accepted by stage-level calls only */
    int32_t depth;                  /* STK_DEPTH_* */
    int32_t location;               /* STK_HOST | STK_DEVICE */
    size_t  row_stride_bytes;       /* 0 = tightly packed */
} stk_frames;

/* Geometry of ONE frame of a stack whose frames differ in size (stk_keypoint_match_mixed). */
typedef struct {
    int32_t width, height;
    size_t  row_stride_bytes;       /* 0 = tightly packed */
} stk_frame_geometry;

/* Caller-allocated f32 image (the returned CV_32FC3 Mat, lib.rs:98,656; CV_32FC4 — channels = 4 — for BGRA stacks). */
typedef struct {
    float*  data;
    int32_t width, height, channels;
    int32_t location;               /* STK_HOST | STK_DEVICE */
    size_t  row_stride_bytes;       /* 0 = tightly packed */
} stk_image_f32;

/* Optional per-frame report (additive; the reference discards these). */
typedef struct {
    int32_t status;                 /* 0 used, 1 dropped (keypoint path), 2 error */
    int32_t iterations;             /* ECC iterations executed */
    double  rho;                    /* final ECC correlation coefficient */
    int32_t n_keypoints;
    int32_t n_matches;              /* after ratio test + truncation */
    int32_t n_inliers;
    int32_t reserved;
    double  warp[9];                /* row-major 3x3 (2x3 padded with 0 0 1) */
} stk_frame_stats;

/* Device-side timing of the last whole-stack call, in milliseconds (HIP
 * events on the context's stream) plus launch counts, for roofline reports. */
typedef struct {
    double prep_ms;        /* grey + blur + gradients (ECC) or ORB (keypoint) */
    double align_ms;       /* ECC iterations or match+RANSAC */
    double warp_ms;        /* warp + accumulate */
    double finalize_ms;
    int64_t ecc_iter_launches;      /* launches of the fused ECC iteration kernel */
    int64_t ecc_slot_iterations;    /* sum over launches of active slots (frame-iterations) */
    int64_t warp_launches;
    int64_t warp_frames;
    /* option "profile" = 2 brackets every ECC iteration launch with its own HIP event pair: */
    double  ecc_iter_ms;            /* sum of those launch durations */
    int64_t ecc_iter_timed;         /* number of launches measured */
    /* host-fed stacks (frames->location == STK_HOST): the copy stream's wall time from the first copy's start to the
     * last copy's end, and the bytes moved; 0 for device-resident stacks */
    double  h2d_ms;
    int64_t h2d_bytes;
    /* keypoint path: time and launches of the dominant ORB kernel (FAST-9/16 + NMS, all pyramid levels), pixels scanned */
    double  fast_ms;
    int64_t fast_launches;
    int64_t fast_pixels;
    /* ECC iteration pass: column strips that started on the per-wave LDS ring, failed its run-time bounds check and were
     * redone by the gather loop (same bits). 0 on every BASELINE stack; non-zero only costs time. */
    int64_t ecc_ring_fallbacks;
} stk_timing;

typedef struct stk_ctx stk_ctx;

/* ---- context ------------------------------------------------------------ */
stk_status  stk_create(int32_t device_id, stk_ctx** out);
/* One context over n_devices GPUs of this node: what a single-process caller (the Rust drop-in) uses to stack on all of
 * them. stk_keypoint_match / stk_ecc_match / stk_hybrid_match (and the *_files forms) then cut the moving frames 1..n-1
 * into contiguous ranges, one per device (stk_shard_moving_frames; replaces the Rayon fold, lib.rs:188-320, 746-818),
 * run one host thread per device, sum the f32 accumulators and the {added, dropped} counters on device_ids[0] with
 * RCCL ncclReduce over xGMI (replaces try_reduce, lib.rs:321-335, 819-833) and scale there (lib.rs:339-345, 836-839).
 * Frames may be host memory or device memory on any of the devices (frames on another device are copied over once).
 * Per-frame results are bit-identical to the single-device run (a frame's summation partition depends on the frame size
 * only); the image differs by the order of the f32 adds only. n_devices == 1 returns a plain context. Stage-level and
 * *_shard entry points on such a context run on device_ids[0]. RCCL (librccl.so.1) is loaded at this call. */
stk_status  stk_create_multi(int32_t n_devices, const int32_t* device_ids, stk_ctx** out);
/* The cut itself: rank `rank` of `world_size` aligns frames [first, first + count) of an n_frames stack (frame 0 is the
 * reference and belongs to nobody's range; rank 0 folds it in). Sizes differ by at most one, earlier ranks are larger. */
stk_status  stk_shard_moving_frames(int32_t n_frames, int32_t world_size, int32_t rank, int32_t* first, int32_t* count);
/* Loads RCCL, sum-reduces `count` floats over the context's devices (a 1-rank communicator on a plain context) and
 * verifies the result on the root: the part of the multi-device path that a one-GPU machine can still execute. */
stk_status  stk_rccl_selftest(stk_ctx* ctx, int64_t count);
void        stk_destroy(stk_ctx* ctx);
const char* stk_last_error(const stk_ctx* ctx);    /* valid until the next call on ctx */
/* Run on a caller-owned hipStream_t (e.g. torch's current stream); NULL
 * restores the context's own stream. */
stk_status  stk_set_stream(stk_ctx* ctx, void* hip_stream);
stk_status  stk_get_timing(const stk_ctx* ctx, stk_timing* out);
/* Pinned (page-locked) host memory for frames: stacks handed over in such buffers cross PCIe by DMA at link rate and
 * overlap with the alignment of the frames that have already arrived (a decoder — the Rust shim's imread — writes into
 * them directly). Pageable frames work too (the HIP runtime locks large sources on the fly: 25 MB frames measured the same
 * 55 GB/s; small or fragmented buffers go through its staging path). */
stk_status  stk_host_alloc(size_t bytes, void** out);
void        stk_host_free(void* p);
/* Tuning knobs. None changes a frame's warp or the stacked image except where noted:
 *   "ecc_slots"          frames iterated concurrently by one ECC launch (0 = auto: the whole stack in one round up to 128 frames,
 *                        beyond that at least three equal rounds of at most 96; rounds of at most 64 for frames up to 1080p;
 *                        1..256); changes no result
 *   "ecc_blocks"         workgroups per ECC launch, all frames in flight together (0 = auto: per frame (column strips x rows) /
 *                        (4 x 112), at most 288 — a function of the frame size only: 288 at 4K, 72 at 1080p); a non-zero
 *                        value changes the f32 summation partition, i.e. results at round-off level (within the stated
 *                        ECC tolerance)
 *   "ecc_variant"        ECC pixel-pass kernel: 3 production (default), 0 the direct cross-check version
 *   "ecc_ring"           homography pass: 1 (default) frame-0 rows go through a per-wave LDS ring where a strip allows it,
 *                        0 every tap is gathered from global memory; the results are bit-identical
 *   "ecc_ring_lookahead" debug: frame-0 rows the ring keeps ahead of the row being fetched (5; 1..4 make its run-time check
 *                        fire, the strips then fall back to the gather loop: stk_timing.ecc_ring_fallbacks); same bits
 *   "ecc_groups"         0 (default): by frame size; 2: the slots form two groups with their own (iterate, solve) launch sequences on
 *                        two streams, one group's solve and launch boundaries running under the other's iteration pass (pays for
 *                        device-resident stacks of frames up to 1080p with >= 32 slots); 1: one sequence. Per-frame results do not depend on it
 *   "ecc_chunk"          (iterate, solve) pairs enqueued between two polls of the completion counter (0 = default: 2 for frames larger
 *                        than 1080p, 4 otherwise)
 *   "kp_lanes"           3 (default; 1..8): device-resident keypoint stacks of >= 16 frames are cut into this many runs of frames (at
 *                        least 8 each) that go through the pipeline side by side, the later ones on hidden helper contexts of
 *                        the same device, so that one run's kernels fill the other runs' host steps; the fold follows run by
 *                        run in stack order. 1: one pipeline. Per-frame results and the stacked image do not depend on it
 *   "orb_resize_tables"  1 (default): ORB's pyramid steps read their bilinear coefficient tables from memory (computed once per
 *                        geometry); 0: every tile computes its own. Same bits either way
 *   "kp_tail_priority"   1 (default): on device-resident stacks that run in several lanes, a lane's descriptor / 2-NN / homography launches
 *                        go to a highest-priority stream (they queue behind the other lanes' large launches otherwise); 0: one stream
 *   "orb_device_cull"    1 (default): ORB's Harris cull (retainBest) and ordering run on the device; 0: on the host pool. Same keypoints
 *   "orb_patch_blur"     1 (default): ORB's 7x7 blur is computed by the descriptor kernel, for the 45 x 40 window around each kept
 *                        keypoint only; 0: every pyramid level is blurred whole first. Same bits either way
 *   "kp_workers"         host threads for the per-frame host steps of the keypoint path (Harris cull, RANSAC)
 *   "warp_subpixel_bits" 0 = exact f32 coordinates (OpenCV >= 4.11 kernels); 5 = classic 1/32-px quantised table
 *                        (changes results: it selects the other OpenCV behaviour)
 *   "profile"            0 off, 1 per-stage events (stk_get_timing), 2 + event pairs around ECC launches
 *   "profile_stride"     with profile = 2: bracket every n-th ECC launch only
 *   "prep_stream"        1 (default): ECC templates of a run of frames by the streaming grey + blur kernel, one launch per
 *                        run; 0: the LDS-tiled kernel, frame by frame. Same bits either way
 *   "prep_overlap"       1 (default): on a device-resident stack of more than 2 x "ecc_slots" frames the ECC templates are
 *                        prepared on a second stream while the first frames already iterate; 0: all templates first.
 *                        Same bits either way (stk_timing.prep_ms then covers the reference frame only)
 *   "upload_batch"       host-fed stacks: frames per host -> HBM batch (default 8); a batch is the unit the ECC queue
 *                        and the batched ORB wait for */
stk_status  stk_set_option(stk_ctx* ctx, const char* name, int64_t value);
const char* stk_version(void);

/* ---- whole-stack entry points: replace lib.rs:129-144 and lib.rs:702-717 -- */
/* keypoint_match(files, params, scale_down_width) -> (dropped, Mat).
 * scale_down_width <= 0 means None. */
stk_status stk_keypoint_match(stk_ctx* ctx, const stk_frames* frames,
                              const stk_keypoint_params* params, float scale_down_width,
                              stk_image_f32* out, int32_t* dropped,
                              stk_frame_stats* stats_or_null);
/* keypoint_match on frames of DIFFERING size, as the reference handles them: every frame is read on its own, ORB runs at
 * the frame's own size, and warp_perspective's dsize is the FIRST frame's (lib.rs:166, 200-204, 290-299) — `out` has
 * geometry[0]'s size. geometry: n entries (frames->width / height / row_stride_bytes are ignored), or NULL = the stack of
 * one geometry `frames` describes; a stack whose entries are all equal takes the batched pipeline of stk_keypoint_match,
 * any other goes frame by frame through the same stages (same per-frame results). 8-bit BGR(A) frames. scale_down_width > 0 is
 * keypoint_match_scale_down (lib.rs:355-601) on such a stack: validated against the FIRST frame's width, every grey shrunk
 * (or enlarged) by scale_image to ITS OWN smaller dimension = scale_down_width, the homography rescaled by that frame's own ratios.
 * (ecc_match has no such form: on frames of differing size the reference fails in cv::add, lib.rs:809 —
 * stk_ecc_match_files reports that as STK_BACKEND_ERROR.) */
stk_status stk_keypoint_match_mixed(stk_ctx* ctx, const stk_frames* frames, const stk_frame_geometry* geometry,
                                    const stk_keypoint_params* params, float scale_down_width, stk_image_f32* out,
                                    int32_t* dropped, stk_frame_stats* stats_or_null);
/* ecc_match(files, params, scale_down_width) -> Mat. */
stk_status stk_ecc_match(stk_ctx* ctx, const stk_frames* frames,
                         const stk_ecc_params* params, float scale_down_width,
                         stk_image_f32* out, stk_frame_stats* stats_or_null);

/* ---- shard-level entry points (one process per GPU; frames[0] is always the
 * reference frame, the remaining entries are this rank's slice of 1..N-1).
 * They produce the UN-normalised f32 sum (the Rayon fold accumulator,
 * lib.rs:306-316 / 807-814) in `sum` (device memory) and the number of frames
 * added; the caller reduces sums/counts across ranks (RCCL) and then calls
 * stk_finalize_mean once on the root. add_reference != 0 adds frame 0 itself
 * (lib.rs:194-196, 752-754) — exactly one rank does that.
 * On any status other than STK_OK the contents of `sum` are UNDEFINED (the fold is enqueued on the device behind the
 * alignment and may already have overwritten it when a frame's failure reaches the host; the reference's `?` yields no
 * image at all): do not reduce or reuse it. */
stk_status stk_ecc_match_shard(stk_ctx* ctx, const stk_frames* frames,
                               const stk_ecc_params* params, float scale_down_width,
                               int32_t add_reference, stk_image_f32* sum,
                               int32_t* n_added, stk_frame_stats* stats_or_null);
stk_status stk_keypoint_match_shard(stk_ctx* ctx, const stk_frames* frames,
                                    const stk_keypoint_params* params, float scale_down_width,
                                    int32_t add_reference, stk_image_f32* sum,
                                    int32_t* n_added, int32_t* n_dropped,
                                    stk_frame_stats* stats_or_null);
/* img / (n as f64)  ==  img * (float)(1.0/n)  (lib.rs:339-345, 836-839). In place if out==sum. */
stk_status stk_finalize_mean(stk_ctx* ctx, const stk_image_f32* sum, int64_t n_frames,
                             stk_image_f32* out);

/* ---- stage-level entry points (parity tests bind these) ------------------ */
/* cvt_color(BGR2GRAY) on the integer image, utils.rs:136-142. out: w*h of the input depth
 * (u8 / u16 / f32), tightly packed, same location as the frame. */
stk_status stk_grey(stk_ctx* ctx, const stk_frames* frame /* n==1 */, void* out);
/* Mat::convert_to(CV_32F, alpha) utils.rs:133 (alpha = 1/255 there). */
stk_status stk_convert_f32(stk_ctx* ctx, const stk_frames* frame /* n==1 */, double alpha,
                           float* out);
/* ---- BASELINE configs[4]: ORB-seeded ECC on 8- or 16-bit stacks — an EXTENSION beyond the reference --------------
 * (both reference paths reject 16-bit input). Definition: ORB + RANSAC homography on the 8-bit grey
 * ((grey16 + 128) / 257 for 16-bit frames), H / h22 cast to f32 is the initial warp of findTransformECC (Homography)
 * on float(grey), fold with alpha = 1/65535 (16-bit) or 1/255 (8-bit). A frame without a homography starts from the
 * identity; nothing is dropped. stats carry the ECC result plus the keypoint / match / inlier counts. */
stk_status stk_hybrid_match(stk_ctx* ctx, const stk_frames* frames, const stk_keypoint_params* kp_params,
                            const stk_ecc_params* ecc_params, stk_image_f32* out, stk_frame_stats* stats);
stk_status stk_hybrid_match_shard(stk_ctx* ctx, const stk_frames* frames, const stk_keypoint_params* kp_params,
                                  const stk_ecc_params* ecc_params, int32_t add_reference, stk_image_f32* sum,
                                  int32_t* n_added, stk_frame_stats* stats);

/* ---- file front-end ---------------------------------------------------------
 * imgcodecs::imread(path, IMREAD_UNCHANGED) (utils.rs:110-117, 132) for binary PNM (P5 / P6, 8 or 16 bit), grey / YCbCr / CMYK
 * JPEG (libjpeg-turbo's libjpeg.so.8, OpenCV's decoder family at its default settings), PNG (libpng16.so.16: 8- and 16-bit
 * grey / RGB, palette -> BGR, 1/2/4-bit grey -> 8 bit; anything with alpha — RGBA, grey + alpha, a tRNS chunk — comes out as
 * FOUR channels B G R A, as OpenCV's decoder delivers it under IMREAD_UNCHANGED) and stripped 8/16-bit grey / RGB TIFF
 * (libtiff.so.5 / .6), the libraries loaded at run time: BGR(A) or grey rows, tightly packed, into `data` (capacity_bytes);
 * data == NULL only reports the geometry. ctx may be NULL. A file that is unreadable or not an image: STK_BACKEND_ERROR (the
 * reference's empty Mat + cvtColor). TIFF may be stripped or tiled, RGBA TIFF gives four channels; BMP (no library:
 * uncompressed 24-bit, 32-bit -> B G R A, 8-bit palette -> BGR or grey); still WebP (libwebp.so.7: BGR, or B G R A when the
 * bitstream has alpha). CMYK / YCCK JPEG comes out as B G R through
 * OpenCV's own conversion. Flavours no decoder here takes (planar TIFF, RLE / 1- / 4- / 16-bit BMP, animated WebP,
 * EXR / JPEG 2000 ...): STK_NOT_IMPLEMENTED — the caller decodes those itself and uses the frame-based entry points. */
stk_status stk_imread(stk_ctx* ctx, const char* path, void* data, size_t capacity_bytes, int32_t* width,
                      int32_t* height, int32_t* channels, int32_t* depth);
/* keypoint_match / ecc_match in the reference's own call shape: a list of file paths, first = reference frame
 * (lib.rs:129-137, 702-710). `out` is a host or device image as for the frame-based entry points. */
stk_status stk_keypoint_match_files(stk_ctx* ctx, const char* const* paths, int32_t n, const stk_keypoint_params* params,
                                    float scale_down_width, stk_image_f32* out, int32_t* dropped, stk_frame_stats* stats);
stk_status stk_ecc_match_files(stk_ctx* ctx, const char* const* paths, int32_t n, const stk_ecc_params* params,
                               float scale_down_width, stk_image_f32* out, stk_frame_stats* stats);
/* stk_hybrid_match on a list of paths (8-bit PNM / PNG / TIFF, 16-bit PNM / TIFF). */
stk_status stk_hybrid_match_files(stk_ctx* ctx, const char* const* paths, int32_t n, const stk_keypoint_params* kp_params,
                                  const stk_ecc_params* ecc_params, stk_image_f32* out, stk_frame_stats* stats);

/* sharpness_modified_laplacian / _variance_of_laplacian / _tenengrad(k_size) / _normalized_gray_level_variance
 * (lib.rs:1030-1166; the pre-filter of examples/main.rs:40-47) of a single-channel 8-bit or f32 image, tightly packed.
 * `ksize` is read by TENG only (1, 3, 5 or 7, else STK_INVALID_PARAMS like lib.rs:1105). */
enum { STK_SHARPNESS_LAPM = 0, STK_SHARPNESS_LAPV = 1, STK_SHARPNESS_TENG = 2, STK_SHARPNESS_GLVN = 3 };
stk_status stk_sharpness(stk_ctx* ctx, const void* grey, int32_t depth, int32_t width, int32_t height,
                         int32_t location, int32_t metric, int32_t ksize, double* out);
/* One frame's whole ECC preparation as ecc_match runs it per frame: cvt_color(BGR2GRAY) (utils.rs:136-142) followed
 * by findTransformECC's own GaussianBlur of the float image (lib.rs:769-777) in one fused pass. `out` is a tightly
 * packed width x height f32 plane in the frame's location. BGR frames, 8-bit or f32. */
stk_status stk_grey_blur_f32(stk_ctx* ctx, const stk_frames* frame /* n==1 */, int32_t ksize, float* out);
/* GaussianBlur(float(grey), g x g, sigma 0, REFLECT_101) as findTransformECC's setup does. */
stk_status stk_gaussian_blur_f32(stk_ctx* ctx, const void* grey, int32_t depth, int32_t width,
                                 int32_t height, int32_t location, int32_t ksize, float* out);
/* video::find_transform_ecc(template, input, warp, motion, criteria, no mask, gauss) lib.rs:769-777.
 * template/input: single-channel u8 (or f32) width x height, tightly packed.
 * warp: 9 floats row-major in/out (2x3 motions use the first 6). rho/iterations optional. */
stk_status stk_find_transform_ecc(stk_ctx* ctx, const void* templ, const void* input,
                                  int32_t depth, int32_t width, int32_t height, int32_t location,
                                  const stk_ecc_params* params, float* warp, double* rho,
                                  int32_t* iterations);
/* imgproc::warp_perspective / warp_affine (INTER_LINEAR, no WARP_INVERSE_MAP: M is inverted)
 * of convert(frame, 1/255)  fused with  acc += warped   (lib.rs:290-316, 780-814).
 * M: 9 doubles row-major (affine: last row 0 0 1). acc: device or host f32 w*h*c.
 * If accumulate == 0 the warped image overwrites acc. alpha is the convert scale. */
stk_status stk_warp_accumulate(stk_ctx* ctx, const stk_frames* frame /* n==1 */, const double* M,
                               int32_t is_affine, int32_t border_mode, const double* border_value,
                               double alpha, int32_t accumulate, stk_image_f32* acc);

/* scale_image (utils.rs:186-214) on an 8-bit grey image: aspect-preserving resize(INTER_AREA) so that the
 * SMALLER dimension becomes scale_down: new_w = (int)(w * f), new_h = (int)(h * f), f = scale_down / min(w, h). out must hold
 * new_w * new_h bytes — more than width * height when scale_down exceeds the smaller dimension (the image is then enlarged,
 * as the reference does for a landscape frame with height < scale_down < width: INTER_AREA's bilinear emulation). */
stk_status stk_scale_image_grey(stk_ctx* ctx, const uint8_t* grey, int32_t width, int32_t height, int32_t location,
                                float scale_down, uint8_t* out, int32_t* new_width, int32_t* new_height);
/* The same on a 32FC1 grey (the grey of a float stack, e.g. float TIFF: ecc_match_scaling_down shrinks whatever depth cvtColor
 * gave it, lib.rs:896, 921; 16-bit greys never get that far — findTransformECC and ORB reject them first). */
stk_status stk_scale_image_grey_f32(stk_ctx* ctx, const float* grey, int32_t width, int32_t height, int32_t location,
                                    float scale_down, float* out, int32_t* new_width, int32_t* new_height);

/* ORB::create_def + detect_and_compute, utils.rs:174-183. keypoints: rows of 7 floats
 * {x, y, size, angle, response, octave, class_id}; descriptors: rows of 32 bytes. */
stk_status stk_orb_detect_and_compute(stk_ctx* ctx, const uint8_t* grey, int32_t width,
                                      int32_t height, int32_t location, int32_t max_keypoints,
                                      float* keypoints, uint8_t* descriptors, int32_t* n_keypoints);
/* BFMatcher(NORM_HAMMING).knn_match(query, k=2) lib.rs:208-219. out rows {train0, dist0, train1, dist1}
 * (-1 where the train set has fewer than 2 rows). Host pointers. */
stk_status stk_bf_knn2_hamming(stk_ctx* ctx, const uint8_t* query, int32_t n_query,
                               const uint8_t* train, int32_t n_train, int32_t* out);
/* calib3d::find_homography(src_pts, dst_pts, mask, method, thr) lib.rs:267-276. Points are host
 * float pairs. H: 9 doubles; *found = 0 when OpenCV would return an empty Mat. */
stk_status stk_find_homography(stk_ctx* ctx, const float* src_pts, const float* dst_pts, int32_t n,
                               int32_t method, double ransac_reproj_threshold, double* H,
                               uint8_t* inlier_mask_or_null, int32_t* found);

#ifdef __cplusplus
}
#endif
#endif /* STACKER_H */
