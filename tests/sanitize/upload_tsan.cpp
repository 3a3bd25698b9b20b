// upload_tsan.cpp — AsyncUpload (the helper thread that enqueues host -> HBM copies batch by batch while the caller's
// thread waits for batches and feeds the pipeline) under ThreadSanitizer, against host-only stand-ins of the HIP calls it
// makes: copies become memcpy, events carry a monotonically increasing stamp. Checks data, batch bookkeeping and that
// repeated start / wait / finish cycles on one context do not race.
#include "hip_stubs.h"

#include <atomic>
#include <cstdio>
#include <vector>

static std::atomic<long> g_stamp{0};
struct FakeEvent { std::atomic<long> at{-1}; };
extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(new FakeEvent()); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { reinterpret_cast<FakeEvent*>(e)->at.store(++g_stamp); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { return reinterpret_cast<FakeEvent*>(e)->at.load() >= 0 ? hipSuccess : hipErrorNotReady; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return reinterpret_cast<FakeEvent*>(e)->at.load() >= 0 ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(reinterpret_cast<FakeEvent*>(b)->at.load() - reinterpret_cast<FakeEvent*>(a)->at.load()); return hipSuccess; }
}
#include "../../libstacker_rs_amd/csrc/upload.cpp"

int main() {
    stk_ctx ctx;
    const int n = 23;
    const size_t fb = 4096;
    std::vector<std::vector<unsigned char>> host(n, std::vector<unsigned char>(fb));
    std::vector<const void*> ptrs(n);
    for (int i = 0; i < n; i++) { for (size_t k = 0; k < fb; k++) host[i][k] = (unsigned char)(i * 7 + k); ptrs[i] = host[i].data(); }
    stk_frames fr{};
    fr.data = ptrs.data(); fr.n = n; fr.width = 32; fr.height = 32; fr.channels = 4; fr.depth = 8; fr.location = STK_HOST;
    std::vector<unsigned char> dev(fb * n);
    for (int rep = 0; rep < 300; rep++) {
        for (int batch : {1, 4, 8, 64}) {
            std::fill(dev.begin(), dev.end(), 0);
            stk::AsyncUpload up;
            if (up.start(&ctx, &fr, dev.data(), fb, batch) != STK_OK) return 2;
            int frames_seen = 0;
            for (int b = 0; b < up.batches(); b++) {
                if (up.wait_batch(b, nullptr) != STK_OK) return 3;
                if (up.recorded() <= b) return 4;
                for (int k = 0; k < up.batch_count(b); k++) {                   // a batch that has been waited for is complete
                    const int i = up.batch_first(b) + k;
                    if (dev[fb * i] != host[i][0] || dev[fb * i + fb - 1] != host[i][fb - 1]) return 5;
                    frames_seen++;
                }
            }
            double ms = 0;
            if (up.finish(&ms) != STK_OK || frames_seen != n || ms <= 0) return 6;
        }
    }
    std::printf("ok\n");
    return 0;
}
