// upload.cpp — see upload.h.
#include "upload.h"

#include "context.h"

namespace stk {

stk_status AsyncUpload::start(stk_ctx* ctx, const stk_frames* frames, void* dst_base, size_t frame_bytes, int batch) {
    ctx_ = ctx; n_frames_ = frames->n; batch_ = std::max(1, batch); frame_bytes_ = frame_bytes;
    first_.clear(); count_.clear();
    first_.push_back(0); count_.push_back(1);                          // the reference frame on its own: its planes come first
    for (int i = 1; i < n_frames_; i += batch_) { first_.push_back(i); count_.push_back(std::min(batch_, n_frames_ - i)); }
    // events are kept by the context and re-used by later calls
    while ((int)ctx->upload_events.size() < batches() + 1) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        ctx->upload_events.push_back(e);
    }
    t0_ = ctx->upload_events[0];
    events_.assign(ctx->upload_events.begin() + 1, ctx->upload_events.begin() + 1 + batches());
    recorded_ = 0; error_ = hipSuccess;
    std::vector<const void*> src(frames->data, frames->data + n_frames_);
    const int device = ctx->device;
    hipStream_t cs = ctx->copy_stream;
    const FrameGate* gate = ctx->frame_gate;
    thread_ = std::thread([this, src, dst_base, device, cs, gate]() {
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipEventRecord(t0_, cs);
        for (int b = 0; b < batches(); b++) {
            for (int k = 0; k < count_[b] && e == hipSuccess; k++) {
                const int i = first_[b] + k;
                if (gate && !gate->wait(src[i])) { e = hipErrorInvalidValue; break; }      // the producer (a decoder) failed
                e = hipMemcpyAsync((uint8_t*)dst_base + frame_bytes_ * (size_t)i, src[i], frame_bytes_, hipMemcpyHostToDevice, cs);
            }
            if (e == hipSuccess) e = hipEventRecord(events_[b], cs);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (e != hipSuccess) { error_ = e; recorded_ = batches(); }     // wake every waiter; they see the error
                else recorded_ = b + 1;
            }
            cv_.notify_all();
            if (e != hipSuccess) return;
        }
    });
    return STK_OK;
}

int AsyncUpload::recorded() {
    std::lock_guard<std::mutex> lk(m_);
    return recorded_;
}

stk_status AsyncUpload::wait_batch(int b, hipStream_t stream) {
    stk_ctx* ctx = ctx_;
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&]() { return recorded_ > b; });
        if (error_ != hipSuccess) return fail(ctx, STK_HIP_ERROR, std::string("host -> HBM copy failed: ") + hipGetErrorString(error_));
    }
    HIP_TRY(hipStreamWaitEvent(stream, events_[b], 0));
    return STK_OK;
}

stk_status AsyncUpload::finish(double* h2d_ms) {
    stk_ctx* ctx = ctx_;
    if (!active()) return STK_OK;
    join();
    n_frames_ = 0;
    if (error_ != hipSuccess) return fail(ctx, STK_HIP_ERROR, std::string("host -> HBM copy failed: ") + hipGetErrorString(error_));
    HIP_TRY(hipEventSynchronize(events_.back()));
    if (h2d_ms) { float ms = 0; (void)hipEventElapsedTime(&ms, t0_, events_.back()); *h2d_ms = ms; }
    return STK_OK;
}

}  // namespace stk
