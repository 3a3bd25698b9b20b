// ransac_host_harness.cpp — the host half of findHomography (cv::RNG stream, sample admissibility, iteration-count
// update) under ASan + UBSan on random, duplicate-ridden and collinear point sets. The device half cannot run here.
#include "hip_stubs.h"
namespace stk { namespace geom {
hipError_t launch_hg_models(const struct HgPoint*, const struct HgFrame*, int, int, const struct HgSample*, int, float*, int*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_hg_refine(const struct HgPoint*, const struct HgJob*, int, struct HgResult*, uint8_t*, hipStream_t) { return hipErrorNotSupported; }
} }
#include "../../libstacker_rs_amd/csrc/homography.cpp"

#include <cstdio>
#include <random>

using namespace stk::geom;

int main() {
    std::mt19937 gen(5);
    std::uniform_real_distribution<float> U(0.f, 1920.f);
    unsigned long sig = 0;
    for (int trial = 0; trial < 300; trial++) {
        const int n = 4 + (int)(gen() % 400);
        std::vector<float> a(2 * n), b(2 * n);
        for (int i = 0; i < n; i++) { a[2 * i] = U(gen); a[2 * i + 1] = U(gen); b[2 * i] = a[2 * i] + 3.f; b[2 * i + 1] = a[2 * i + 1] * 1.01f; }
        if (trial % 5 == 1) for (int i = 0; i < n; i++) { a[2 * i + 1] = 2.f * a[2 * i]; b[2 * i + 1] = 2.f * b[2 * i]; }   // all collinear
        if (trial % 5 == 2) for (int i = 1; i < n; i += 2) { a[2 * i] = a[0]; a[2 * i + 1] = a[1]; }                        // duplicates
        HgProblem pr{a.data(), b.data(), n, nullptr};
        Track t;
        t.n = n;
        for (int k = 0; k < 64 && !t.stream_ended; k++) {
            HgSample s{};
            if (draw_sample(t, pr, s)) { for (int q = 0; q < 4; q++) { if (s.idx[q] < 0 || s.idx[q] >= n) return 2; sig = sig * 31 + (unsigned)s.idx[q]; } }
            else t.stream_ended = true;
        }
        for (int good = 0; good <= n; good += 1 + n / 7) sig = sig * 31 + (unsigned)iterations_needed(0.995, (double)(n - good) / n, 2000);
    }
    // first outputs of cv::RNG(-1): state = lo * 4164903690 + hi
    MwcStream r;
    const int first = r.below(1000), second = r.below(1000);
    std::printf("ok %lu %d %d\n", sig, first, second);
    return 0;
}
