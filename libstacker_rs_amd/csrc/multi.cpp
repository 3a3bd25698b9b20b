// multi.cpp — one stk_ctx over several GPUs of a node: what the Rust drop-in's single process calls.
//
// The reference folds frames on a Rayon pool and tree-reduces the per-thread accumulators (lib.rs:188-335, 746-833).
// Here the moving frames 1..n-1 are cut into contiguous ranges, one per device (the same cut as shard.py, so the
// per-frame results equal the one-process-per-GPU runs bit for bit); one host thread per device runs the ordinary
// shard-level entry point on that device's own context (own stream, own workspace; frame 0's planes / descriptors
// are recomputed locally instead of being broadcast); the one exchange of the path is an RCCL ncclReduce(sum) of
// the f32 accumulators over xGMI plus one of the {added, dropped} counters, issued as a single group call from the
// calling thread once every device has finished aligning; the root then scales by (float)(1 / (n - dropped)).
//
// RCCL is bound at run time (dlopen of librccl.so.1 when the first multi-device context is created): single-GPU users
// never load it, and inside a PyTorch process the already-loaded copy with the same SONAME is the one that is used.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <thread>

#include "context.h"
#include "host_pool.h"

using namespace stk;

namespace stk {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

struct MultiState {
    std::vector<stk_ctx*> members;        // members[0] is the owning context itself
    std::vector<int> devices;
    bool distinct = true;                 // all members on different devices (else: one-GPU rehearsal, local adds)
    RcclApi api;
    std::vector<ncclComm_t> comms;
    std::vector<DevBuf> sums;             // per member: f32 accumulator of its shard
    std::vector<DevBuf> counts;           // per member: int32 {added, dropped}
    std::vector<DevBuf> stage;            // per member: copies of frames that live on ANOTHER device
};

}  // namespace stk

int multi_member_count(const stk_ctx* ctx) { return ctx && ctx->multi ? (int)ctx->multi->members.size() : 1; }
stk_ctx* multi_member(const stk_ctx* ctx, int i) { return ctx->multi->members[i]; }

// contiguous ranges of the moving frames 1..n-1, sizes differing by at most one, earlier ranks take the remainder
static void shard_range(int n_frames, int world, int rank, int& first, int& count) {
    const int moving = n_frames - 1, base = moving / world, rem = moving % world;
    first = 1 + rank * base + std::min(rank, rem);
    count = base + (rank < rem ? 1 : 0);
}

static stk_status load_rccl(stk_ctx* ctx, RcclApi& a) {
    if (a.lib) return STK_OK;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (a.lib) break;
    }
    if (!a.lib) return fail(ctx, STK_HIP_ERROR, std::string("RCCL (librccl.so.1) could not be loaded: ") + dlerror());
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.lib, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.lib, "ncclCommDestroy");
    a.Reduce = (decltype(a.Reduce))dlsym(a.lib, "ncclReduce");
    a.GroupStart = (decltype(a.GroupStart))dlsym(a.lib, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.lib, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.lib, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.Reduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString)
        return fail(ctx, STK_HIP_ERROR, "RCCL: a required symbol is missing");
    return STK_OK;
}

#define NCCL_TRY(expr)                                                                                        \
    do {                                                                                                      \
        ncclResult_t r_ = (expr);                                                                             \
        if (r_ != ncclSuccess) return fail(ctx, STK_HIP_ERROR, std::string(#expr) + ": " + ms->api.GetErrorString(r_)); \
    } while (0)

void multi_destroy(stk_ctx* ctx) {
    MultiState* ms = ctx->multi;
    if (!ms) return;
    for (size_t i = 0; i < ms->comms.size(); i++) if (ms->comms[i]) (void)ms->api.CommDestroy(ms->comms[i]);
    for (size_t i = 0; i < ms->members.size(); i++) {
        (void)hipSetDevice(ms->devices[i]);
        // (the buffers are sized for all devices before the first member is created: a failed stk_create_multi arrives
        // here with fewer members than buffers, never the other way round)
        if (i < ms->sums.size()) { ms->sums[i].release(); ms->counts[i].release(); ms->stage[i].release(); }
        if (i > 0) stk_destroy(ms->members[i]);
    }
    (void)hipSetDevice(ctx->device);
    delete ms;
    ctx->multi = nullptr;
}

enum MultiKind { MULTI_ECC, MULTI_KEYPOINT, MULTI_HYBRID };

// The whole-stack call on all members. `out` may be host or device (on members[0]'s device) memory.
stk_status multi_match(stk_ctx* ctx, int kind, const stk_frames* frames, const stk_keypoint_params* kp, const stk_ecc_params* ep,
                       float scale_down_width, stk_image_f32* out, int32_t* dropped_out, stk_frame_stats* stats) {
    MultiState* ms = ctx->multi;
    stk_status st = check_frames(ctx, frames, true);
    if (st) return st;
    const int w = frames->width, h = frames->height, n = frames->n;
    const int cn = frames->channels == 4 ? 4 : 3;
    if ((st = image_check(ctx, out, w, h, cn))) return st;
    if (out->row_stride_bytes) return fail(ctx, STK_INVALID_PARAMS, "output must be tightly packed");
    const int world = (int)ms->members.size();
    const size_t nel = (size_t)w * h * cn;
    for (int r = 0; r < world; r++) {
        (void)hipSetDevice(ms->devices[r]);
        HIP_TRY(ms->sums[r].reserve(nel * sizeof(float)));
        HIP_TRY(ms->counts[r].reserve(2 * sizeof(int32_t)));
    }
    // ---- phase 1: every device aligns and folds its range (one host thread per device) ----
    std::vector<stk_status> status(world, STK_OK);
    std::vector<int32_t> added(world, 0), ndropped(world, 0);
    std::vector<std::vector<stk_frame_stats>> sub_stats(world);
    std::vector<std::thread> workers;
    auto body = [&](int r) {
        stk_ctx* c = ms->members[r];
        (void)hipSetDevice(ms->devices[r]);
        // frames that are still being decoded (path-based entry points): every member asks the caller's gate
        struct GateLoan {
            stk_ctx* member;
            GateLoan(stk_ctx* m, const FrameGate* g) : member(m) { if (member) member->frame_gate = g; }
            ~GateLoan() { if (member) member->frame_gate = nullptr; }
        } loan(r > 0 ? c : nullptr, ctx->frame_gate);
        int first, count;
        shard_range(n, world, r, first, count);
        std::vector<const void*> ptrs(1 + count);
        ptrs[0] = frames->data[0];
        for (int k = 0; k < count; k++) ptrs[1 + k] = frames->data[first + k];
        stk_frames sub = *frames;
        sub.data = ptrs.data(); sub.n = 1 + count;
        if (frames->location == STK_DEVICE) {
            // frames resident on another device of the node (e.g. frame 0 for every member but the first) are copied over
            // xGMI into this member's staging buffer once; frames already here are used in place
            const size_t fb = frame_row_bytes(frames) * (size_t)h;
            std::vector<int> remote;
            for (int k = 0; k < sub.n; k++) {
                hipPointerAttribute_t at{};
                if (hipPointerGetAttributes(&at, ptrs[k]) == hipSuccess && at.device != ms->devices[r]) remote.push_back(k);
            }
            if (!remote.empty()) {
                if (ms->stage[r].reserve(fb * remote.size()) != hipSuccess) { status[r] = fail(c, STK_HIP_ERROR, "staging allocation failed"); return; }
                for (size_t q = 0; q < remote.size(); q++) {
                    void* d = ms->stage[r].as<uint8_t>() + fb * q;
                    if (hipMemcpyAsync(d, ptrs[remote[q]], fb, hipMemcpyDefault, c->stream) != hipSuccess) { status[r] = fail(c, STK_HIP_ERROR, "peer copy failed"); return; }
                    ptrs[remote[q]] = d;
                }
            }
        }
        stk_image_f32 sum{ms->sums[r].as<float>(), w, h, cn, STK_DEVICE, 0};
        sub_stats[r].resize(sub.n);
        stk_frame_stats* sst = stats ? sub_stats[r].data() : nullptr;
        const int add_ref = r == 0;
        if (count == 0 && !add_ref) {                  // more devices than moving frames: contributes zeros
            if (hipMemsetAsync(sum.data, 0, nel * sizeof(float), c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
                status[r] = fail(c, STK_HIP_ERROR, "hipMemsetAsync failed");
            return;
        }
        if (kind == MULTI_ECC) status[r] = stk_ecc_match_shard(c, &sub, ep, scale_down_width, add_ref, &sum, &added[r], sst);
        else if (kind == MULTI_KEYPOINT) status[r] = stk_keypoint_match_shard(c, &sub, kp, scale_down_width, add_ref, &sum, &added[r], &ndropped[r], sst);
        else status[r] = stk_hybrid_match_shard(c, &sub, kp, ep, add_ref, &sum, &added[r], sst);
    };
    for (int r = 1; r < world; r++) workers.emplace_back(body, r);
    body(0);
    for (auto& t : workers) t.join();
    (void)hipSetDevice(ctx->device);
    for (int r = 0; r < world; r++)
        if (status[r]) {                               // first failing range decides, like `?` in the reference's fold
            if (r > 0) fail(ctx, status[r], std::string(stk_last_error(ms->members[r])) + " [device " + std::to_string(ms->devices[r]) + "]");
            return status[r];
        }
    if (stats) {
        for (int r = 0; r < world; r++) {
            int first, count;
            shard_range(n, world, r, first, count);
            if (r == 0) stats[0] = sub_stats[0][0];
            for (int k = 0; k < count; k++) stats[first + k] = sub_stats[r][1 + k];
        }
    }
    // ---- phase 2: the one exchange — accumulators and counters to the root ----
    int32_t tot_added = 0, tot_dropped = 0;
    if (ms->distinct) {
        for (int r = 0; r < world; r++) {
            (void)hipSetDevice(ms->devices[r]);
            const int32_t c2[2] = {added[r], ndropped[r]};
            HIP_TRY(hipMemcpyAsync(ms->counts[r].p, c2, sizeof(c2), hipMemcpyHostToDevice, ms->members[r]->stream));
            HIP_TRY(hipStreamSynchronize(ms->members[r]->stream));      // c2 is a stack temporary
        }
        NCCL_TRY(ms->api.GroupStart());
        ncclResult_t gr = ncclSuccess;                               // a failure inside the group must still close it
        for (int r = 0; r < world && gr == ncclSuccess; r++) {
            gr = ms->api.Reduce(ms->sums[r].p, ms->sums[r].p, nel, ncclFloat32, ncclSum, 0, ms->comms[r], ms->members[r]->stream);
            if (gr == ncclSuccess)
                gr = ms->api.Reduce(ms->counts[r].p, ms->counts[r].p, 2, ncclInt32, ncclSum, 0, ms->comms[r], ms->members[r]->stream);
        }
        const ncclResult_t ge = ms->api.GroupEnd();
        if (gr != ncclSuccess || ge != ncclSuccess)
            return fail(ctx, STK_HIP_ERROR, std::string("ncclReduce of the accumulators: ") + ms->api.GetErrorString(gr != ncclSuccess ? gr : ge));
        for (int r = world - 1; r >= 0; r--) {
            (void)hipSetDevice(ms->devices[r]);
            HIP_TRY(hipStreamSynchronize(ms->members[r]->stream));
        }
        int32_t c2[2] = {0, 0};
        HIP_TRY(hipMemcpy(c2, ms->counts[0].p, sizeof(c2), hipMemcpyDeviceToHost));
        tot_added = c2[0]; tot_dropped = c2[1];
    } else {
        // members share a device (one-GPU rehearsal of the sharded control flow): plain adds in rank order
        (void)hipSetDevice(ctx->device);
        for (int r = 0; r < world; r++) { tot_added += added[r]; tot_dropped += ndropped[r]; }
        for (int r = 1; r < world; r++) HIP_TRY(launch_add(ms->sums[0].as<float>(), ms->sums[r].as<float>(), nel, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    if (dropped_out) *dropped_out = tot_dropped;
    if (tot_added <= 0)   // lib.rs:324
        return fail(ctx, STK_INVALID_PARAMS, "All images discarded: try modifying KeyPointMatchParameters::match_distance_threshold");
    stk_image_f32 sum0{ms->sums[0].as<float>(), w, h, cn, STK_DEVICE, 0};
    // keypoint: img / (n - dropped) (lib.rs:342); ecc / hybrid: img / n (lib.rs:836-839)
    return stk_finalize_mean(ctx, &sum0, kind == MULTI_KEYPOINT ? (int64_t)n - tot_dropped : (int64_t)n, out);
}

extern "C" {

stk_status stk_shard_moving_frames(int32_t n_frames, int32_t world_size, int32_t rank, int32_t* first, int32_t* count) {
    if (!first || !count || n_frames <= 0 || world_size <= 0 || rank < 0 || rank >= world_size) return STK_INVALID_PARAMS;
    int f, c;
    shard_range(n_frames, world_size, rank, f, c);
    *first = f; *count = c;
    return STK_OK;
}

stk_status stk_create_multi(int32_t n_devices, const int32_t* device_ids, stk_ctx** out) {
    if (!out) return STK_INVALID_PARAMS;
    *out = nullptr;
    if (n_devices <= 0 || n_devices > 64 || !device_ids) return STK_INVALID_PARAMS;
    stk_ctx* ctx = nullptr;
    stk_status st = stk_create(device_ids[0], &ctx);
    if (st) return st;
    if (n_devices == 1) { *out = ctx; return STK_OK; }        // a one-device group IS the plain context: same code, same bits
    MultiState* ms = new MultiState();
    ctx->multi = ms;
    ms->members.push_back(ctx); ms->devices.push_back(device_ids[0]);
    ms->sums.resize(n_devices); ms->counts.resize(n_devices); ms->stage.resize(n_devices);
    for (int i = 1; i < n_devices; i++) {
        stk_ctx* c = nullptr;
        if ((st = stk_create(device_ids[i], &c))) { stk_destroy(ctx); return st; }      // e.g. a device id the node does not have
        ms->members.push_back(c); ms->devices.push_back(device_ids[i]);
        for (int k = 0; k < i; k++) if (device_ids[k] == device_ids[i]) ms->distinct = false;
    }
    if (ms->distinct) {
        if ((st = load_rccl(ctx, ms->api))) { stk_destroy(ctx); return st; }
        ms->comms.assign(n_devices, nullptr);
        ncclResult_t r = ms->api.CommInitAll(ms->comms.data(), n_devices, ms->devices.data());
        if (r != ncclSuccess) { stk_destroy(ctx); return STK_HIP_ERROR; }
    }
    // ONE host pool for the per-frame host steps of all members (Harris cull, match filter: keypoint.cpp), sized for the
    // host rather than for the number of GPUs: every member thread takes part in its own runs, the pool adds the rest
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    const int workers = std::max(1, std::min(hw, 12 * n_devices) - n_devices);
    ctx->host_pool = new HostPool(workers);
    for (stk_ctx* m : ms->members) m->shared_pool = ctx->host_pool;
    (void)hipSetDevice(ctx->device);
    *out = ctx;
    return STK_OK;
}

// RCCL self-test on whatever devices the context spans (also a plain context: a 1-rank communicator): loads the library,
// creates a communicator, sum-reduces `count` floats and an int pair to the root and checks the result. Used by the GPU
// tests on the one-GPU box, where the multi-device reduce itself cannot run.
stk_status stk_rccl_selftest(stk_ctx* ctx, int64_t count) {
    if (!ctx || count <= 0) return STK_INVALID_PARAMS;
    MultiState local;
    MultiState* ms = ctx->multi && ctx->multi->distinct ? ctx->multi : &local;
    const bool own = ms == &local;
    if (own) {
        ms->members.push_back(ctx); ms->devices.push_back(ctx->device);
        stk_status st = load_rccl(ctx, ms->api);
        if (st) return st;
        ms->comms.assign(1, nullptr);
        NCCL_TRY(ms->api.CommInitAll(ms->comms.data(), 1, ms->devices.data()));
    }
    const int world = (int)ms->members.size();
    std::vector<DevBuf> bufs(world);
    std::vector<float> host((size_t)count);
    stk_status result = STK_OK;
    auto cleanup = [&]() {
        for (int r = 0; r < world; r++) { (void)hipSetDevice(ms->devices[r]); bufs[r].release(); }
        if (own) (void)ms->api.CommDestroy(ms->comms[0]);
        (void)hipSetDevice(ctx->device);
    };
    for (int r = 0; r < world && !result; r++) {
        (void)hipSetDevice(ms->devices[r]);
        for (int64_t i = 0; i < count; i++) host[(size_t)i] = (float)((i % 251) + r);
        if (bufs[r].reserve((size_t)count * sizeof(float)) != hipSuccess ||
            hipMemcpy(bufs[r].p, host.data(), (size_t)count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            result = fail(ctx, STK_HIP_ERROR, "rccl selftest: allocation / upload failed");
    }
    if (!result) {
        ncclResult_t nr = ms->api.GroupStart();
        if (nr == ncclSuccess) {
            for (int r = 0; r < world && nr == ncclSuccess; r++)
                nr = ms->api.Reduce(bufs[r].p, bufs[r].p, (size_t)count, ncclFloat32, ncclSum, 0, ms->comms[r], ms->members[r]->stream);
            const ncclResult_t ge = ms->api.GroupEnd();              // closed on the error path too
            if (nr == ncclSuccess) nr = ge;
        }
        if (nr != ncclSuccess) result = fail(ctx, STK_HIP_ERROR, std::string("rccl selftest: ") + ms->api.GetErrorString(nr));
    }
    for (int r = 0; r < world && !result; r++) {
        (void)hipSetDevice(ms->devices[r]);
        if (hipStreamSynchronize(ms->members[r]->stream) != hipSuccess) result = fail(ctx, STK_HIP_ERROR, "rccl selftest: sync failed");
    }
    if (!result) {
        (void)hipSetDevice(ms->devices[0]);
        if (hipMemcpy(host.data(), bufs[0].p, (size_t)count * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
            result = fail(ctx, STK_HIP_ERROR, "rccl selftest: download failed");
        const float extra = (float)(world * (world - 1) / 2);
        for (int64_t i = 0; i < count && !result; i++)
            if (host[(size_t)i] != (float)(i % 251) * world + extra) result = fail(ctx, STK_PROCESSING_ERROR, "rccl selftest: wrong sum");
    }
    cleanup();
    return result;
}

}  // extern "C"
