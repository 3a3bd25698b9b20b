// hostpool_tsan.cpp — the keypoint path's HostPool under ThreadSanitizer: alternating run(n - 1) / run(n) thousands of
// times, the pattern keypoint_match produces (match step over n - 1 frames, next ORB step over n) and the one in which a
// worker still leaving run k could claim an index of run k + 1 before round 2 gave every run its own job object.
#include <atomic>
#include <cstdio>
#include <thread>
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
    // several callers at once on one pool: the member threads of a multi-device context (multi.cpp) share it
    {
        stk::HostPool shared(6);
        std::atomic<long> sum{0};
        std::atomic<int> bad{0};
        std::vector<std::thread> callers;
        for (int c = 0; c < 4; c++)
            callers.emplace_back([&, c]() {
                for (int rep = 0; rep < 3000; rep++) {
                    const int m = 3 + (rep + c) % 7;
                    std::vector<int> hits(m, 0);
                    const std::function<void(int)> fn = [&](int i) { hits[i]++; };
                    shared.run(m, fn);
                    for (int i = 0; i < m; i++) { if (hits[i] != 1) bad++; sum++; }
                }
            });
        for (auto& t : callers) t.join();
        if (bad.load()) { std::printf("%d indices ran a wrong number of times with concurrent callers\n", bad.load()); return 1; }
        total += sum.load();
    }
    std::printf("ok %ld\n", total);
    return 0;
}
