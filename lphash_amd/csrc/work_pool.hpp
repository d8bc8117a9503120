// work_pool.hpp -- a few parked threads that run fn(0) .. fn(n - 1) on request, the caller's thread among them.  The batch reader used to start
// one std::thread per piece per batch (15 thread creations of ~40 us for every batch of a 16-thread parse: most of the per-batch cost of small
// batches); the pool is created once, before a caller's timer when the caller asks (FastxBatchReader::prealloc).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstddef>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace lphq {

class WorkPool {
public:
    explicit WorkPool(int threads) {
        for (int t = 1; t < threads; ++t) th_.emplace_back([this] { worker(); });
    }
    ~WorkPool() {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    WorkPool(const WorkPool&) = delete;
    WorkPool& operator=(const WorkPool&) = delete;
    int threads() const { return (int)th_.size() + 1; }

    // fn(i) for every i in [0, n), each exactly once, on the pool's threads and the caller's; returns when all are done
    void run(size_t n, const std::function<void(size_t)>& fn) {
        if (n == 0) return;
        if (n == 1 || th_.empty()) {
            for (size_t i = 0; i < n; ++i) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            fn_ = &fn;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            pending_ = n;
            ++gen_;
        }
        cv_.notify_all();
        size_t done = 0;
        for (size_t i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < n; ++done) fn(i);
        std::unique_lock<std::mutex> g(mu_);
        pending_ -= done;
        done_cv_.wait(g, [&] { return pending_ == 0 && busy_ == 0; });  // (no worker still holds a pointer to fn)
        fn_ = nullptr;
    }

private:
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> g(mu_);
        for (;;) {
            cv_.wait(g, [&] { return quit_ || gen_ != seen; });
            if (quit_) return;
            seen = gen_;
            if (!fn_) continue;
            const std::function<void(size_t)>* fn = fn_;
            const size_t n = n_;
            ++busy_;
            g.unlock();
            size_t done = 0;
            for (size_t i; (i = next_.fetch_add(1, std::memory_order_relaxed)) < n; ++done) (*fn)(i);
            g.lock();
            pending_ -= done;
            --busy_;
            if (pending_ == 0 && busy_ == 0) done_cv_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    const std::function<void(size_t)>* fn_ = nullptr;
    size_t n_ = 0, pending_ = 0;
    int busy_ = 0;
    std::atomic<size_t> next_{0};
    uint64_t gen_ = 0;
    bool quit_ = false;
};

}  // namespace lphq
