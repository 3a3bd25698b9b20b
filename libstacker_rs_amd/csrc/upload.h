// upload.h — host -> HBM transfer of a frame stack in batches, overlapped with the compute that consumes it
// (the `read_grey_and_f32` hand-over of utils.rs:128-144 when the caller's frames live in host memory).
//
// A helper thread enqueues the copies on the context's copy stream (batch 0 = the reference frame alone, then `batch`
// frames each) and records one event per batch. Frames in pinned memory (stk_host_alloc, hipHostMalloc, hipHostRegister)
// go by DMA at PCIe rate; for pageable frames the HIP runtime locks or stages the source and may block the enqueuing
// thread — the reason the enqueuing happens off the caller's thread (25 MB malloc'ed frames measured the same 55 GB/s).
// Consumers make their stream wait for the batch they need.
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

struct stk_ctx;

namespace stk {

class AsyncUpload {
public:
    AsyncUpload() = default;
    ~AsyncUpload() { join(); }
    AsyncUpload(const AsyncUpload&) = delete;
    AsyncUpload& operator=(const AsyncUpload&) = delete;

    // Starts copying frames->data[i] (host) to dst_base + i * frame_bytes. Returns at once.
    stk_status start(stk_ctx* ctx, const stk_frames* frames, void* dst_base, size_t frame_bytes, int batch);
    bool active() const { return n_frames_ > 0; }
    int batches() const { return (int)first_.size(); }
    int batch_first(int b) const { return first_[b]; }
    int batch_count(int b) const { return count_[b]; }
    int batch_of_frame(int i) const { return i == 0 ? 0 : 1 + (i - 1) / batch_; }
    // batches whose copies have been enqueued and whose event is recorded (non-blocking)
    int recorded();
    // blocks until batch b is enqueued, then makes `stream` wait for its completion
    stk_status wait_batch(int b, hipStream_t stream);
    stk_status wait_frame(int i, hipStream_t stream) { return wait_batch(batch_of_frame(i), stream); }
    // joins the helper thread; reports its failure, and the wall time of the copy stream between the first copy's start
    // and the last copy's end (ms) for the host-fed bench figure
    stk_status finish(double* h2d_ms);
    size_t bytes() const { return frame_bytes_ * (size_t)n_frames_; }

private:
    void join() { if (thread_.joinable()) thread_.join(); }
    stk_ctx* ctx_ = nullptr;
    int n_frames_ = 0, batch_ = 1;
    size_t frame_bytes_ = 0;
    std::vector<int> first_, count_;
    std::vector<hipEvent_t> events_;
    hipEvent_t t0_ = nullptr;
    std::thread thread_;
    std::mutex m_;
    std::condition_variable cv_;
    int recorded_ = 0;
    hipError_t error_ = hipSuccess;
};

}  // namespace stk
