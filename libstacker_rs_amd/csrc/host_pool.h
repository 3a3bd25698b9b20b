// host_pool.h — persistent host worker pool of the keypoint path (pure C++, no HIP: also compiled by the CPU
// sanitizer tests, tests/test_cpu_sanitize.py).
#pragma once
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace stk {

// A small persistent pool for the per-frame host steps (spawning 12 threads twice per stack cost ~0.5 ms of a 6 ms stack).
// Several threads may call run() at the same time — the member threads of a multi-device context share ONE pool sized for
// the host, not for the number of GPUs (round 2 gave every member 12 threads of its own: 96 + 8 on an 8-GPU node with 32
// cores). Every run() owns a Job (claim counter, size, function, completion count) on a list of active jobs; a worker
// takes the next index of the first job that still has one. A worker reaches a job only through a shared_ptr taken under
// the lock: one that is still leaving a finished run can only ever touch that run's (exhausted) counter, never a later run's.
class HostPool {
public:
    explicit HostPool(int n) {
        for (int i = 0; i < n; i++) workers_.emplace_back([this]() { loop(); });
    }
    ~HostPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    int size() const { return (int)workers_.size(); }
    // run fn(i) for i in [0, n); the calling thread takes part; returns when all are done. Thread-safe.
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        auto job = std::make_shared<Job>(n, &fn);
        {
            std::lock_guard<std::mutex> lk(m_);
            jobs_.push_back(job);
        }
        cv_.notify_all();
        work(*job);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&]() { return job->finished == job->n; });
        jobs_.erase(std::find(jobs_.begin(), jobs_.end(), job));
    }

private:
    struct Job {
        Job(int n_, const std::function<void(int)>* fn_) : n(n_), fn(fn_) {}
        const int n;
        const std::function<void(int)>* const fn;   // valid until `finished == n` (run() does not return before)
        std::atomic<int> next{0};
        int finished = 0;                           // guarded by m_
    };
    void work(Job& job) {
        for (;;) {
            const int i = job.next.fetch_add(1);
            if (i >= job.n) break;
            (*job.fn)(i);
            std::lock_guard<std::mutex> lk(m_);
            if (++job.finished == job.n) done_.notify_all();
        }
    }
    // a job with unclaimed indices, or null (caller holds m_)
    std::shared_ptr<Job> pending() const {
        for (const auto& j : jobs_)
            if (j->next.load(std::memory_order_relaxed) < j->n) return j;
        return nullptr;
    }
    void loop() {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return stop_ || (job = pending()) != nullptr; });
                if (stop_) return;
            }
            work(*job);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::vector<std::shared_ptr<Job>> jobs_;        // active runs (guarded by m_)
    bool stop_ = false;
};

}  // namespace stk
