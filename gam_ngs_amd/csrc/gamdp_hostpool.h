// libgamdp host side: the thread pool behind the parallel loops of a batch call (gamdp_host.cpp: parallel_for).  Standard library only,
// so that tests/test_hostpool.py can build it with ThreadSanitizer on a host without HIP.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace gamdp {

// The parallel loops of a batch call: fn(lo, hi) over [0, n) on up to 16 host threads.
// The threads are the process's own and stay: starting and joining 15 threads cost a parallel loop 0.5 - 0.8 ms, four
// loops per batch call.  One parallel loop at a time: a second caller (the other thread of a batch that goes through in pieces, the
// host thread of another device) waits its turn when its loop is large -- every loop then still runs on all threads, one after
// the other -- and runs a small loop itself, on its own thread, instead of waiting.
class HostPool {
public:
    static HostPool& get() { static HostPool p; return p; }
    template <class F>
    void run(size_t n, F& fn)
    {
        if (workers_.empty()) { fn((size_t)0, n); return; }
        std::unique_lock<std::mutex> one(use_, std::try_to_lock);
        if (!one.owns_lock()) {
            if (n < kWaitFrom) { fn((size_t)0, n); return; }
            one.lock();
        }
        const unsigned parts = (unsigned)workers_.size() + 1;
        auto body = [&](unsigned k) { fn(n * k / parts, n * (k + 1) / parts); };
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = [&body](unsigned k) { body(k); };
            parts_ = parts; next_ = 0; done_ = 0; ++epoch_;
        }
        cv_work_.notify_all();
        work();   // the caller takes parts too
        std::unique_lock<std::mutex> g(m_);
        cv_done_.wait(g, [&] { return done_ == parts_; });
        job_ = nullptr;
    }
    static constexpr size_t kWaitFrom = 32768;   // elements from which a caller that finds the pool busy waits for it
private:
    HostPool()
    {
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        try { for (unsigned k = 1; k < hw; k++) workers_.emplace_back([this] { loop(); }); } catch (...) {}   // fewer threads, or none: still correct
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_work_.notify_all();
        for (auto& t : workers_) if (t.joinable()) t.join();
    }
    void work()
    {
        for (;;) {
            unsigned k;
            std::function<void(unsigned)> job;
            {
                std::lock_guard<std::mutex> g(m_);
                if (!job_ || next_ >= parts_) return;
                k = next_++;
                job = job_;
            }
            job(k);
            bool last;
            { std::lock_guard<std::mutex> g(m_); last = ++done_ == parts_; }
            if (last) cv_done_.notify_all();
        }
    }
    void loop()
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_work_.wait(g, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
            }
            work();
        }
    }
    std::mutex use_, m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> workers_;
    std::function<void(unsigned)> job_;
    unsigned parts_ = 0, next_ = 0, done_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

}  // namespace gamdp
