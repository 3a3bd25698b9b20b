// hostpool_tsan.cpp — the keypoint path's HostPool under ThreadSanitizer: alternating run(n - 1) / run(n) thousands of
// times, the pattern keypoint_match produces (match step over n - 1 frames, next ORB step over n) and the one in which a
// worker still leaving run k could claim an index of run k + 1 before round 2 gave every run its own job object.
#include <cstdio>
#include <vector>

#include "../../libstacker_rs_amd/csrc/host_pool.h"

int main() {
    stk::HostPool pool(11);
    const int n = 9;
    long total = 0;
    for (int rep = 0; rep < 20000; rep++) {
        for (int m : {n - 1, n}) {
            std::vector<int> hits(m, 0);
            const std::function<void(int)> fn = [&](int i) { hits[i]++; };   // a second claim of an index would be a race AND a count of 2
            pool.run(m, fn);
            for (int i = 0; i < m; i++) { if (hits[i] != 1) { std::printf("index %d ran %d times in rep %d\n", i, hits[i], rep); return 1; } total++; }
        }
    }
    std::printf("ok %ld\n", total);
    return 0;
}
