// host_pool.h — persistent host worker pool of the keypoint path (pure C++, no HIP: also compiled by the CPU
// sanitizer tests, tests/test_cpu_sanitize.py).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace stk {

// A small persistent pool for the per-frame host steps (spawning 12 threads twice per stack cost ~0.5 ms of a 6 ms stack).
// Every run() owns a Job (claim counter, size, function, completion count) that the workers reach through a
// shared_ptr snapshot taken under the lock: a worker that is still leaving the previous run can only ever touch that
// run's (exhausted) counter, never the next run's.
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
    // run fn(i) for i in [0, n); the calling thread takes part; returns when all are done
    void run(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        auto job = std::make_shared<Job>(n, &fn);
        {
            std::lock_guard<std::mutex> lk(m_);
            job_ = job; gen_++;
        }
        cv_.notify_all();
        work(*job);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&]() { return job->finished == job->n; });
        job_.reset();
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
    void loop() {
        unsigned long long seen = 0;
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&]() { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                job = job_;
            }
            if (job) work(*job);
        }
    }
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::shared_ptr<Job> job_;
    unsigned long long gen_ = 0;
    bool stop_ = false;
};

}  // namespace stk
