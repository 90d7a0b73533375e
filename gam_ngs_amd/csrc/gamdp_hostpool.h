// libgamdp host side: the thread pool behind the parallel loops of a batch call (gamdp_host.cpp: parallel_for).  Standard library only,
// so that tests/test_hostpool.py can build it with ThreadSanitizer on a host without HIP.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace gamdp {

// The parallel loops of a batch call: fn(lo, hi) over [0, n) in up to kWidth parts.
// The threads are the process's own and stay (starting and joining 15 threads cost a parallel loop 0.5 - 0.8 ms, four loops per batch
// call).  SEVERAL LOOPS RUN AT ONCE (round 6): the host threads of the D devices of a gamdp_multi call, or the two threads of a
// batch that goes through in pieces, each post their loop as a job; the workers take parts from the jobs in turn, every caller
// works on its own job, so the host phases of D contexts overlap like the reference's N independent workers
// (lib/src/pctg/ThreadedBuildPctg.cc:150-175) instead of queueing behind one another.  A loop alone still runs kWidth wide (the
// loops are memory-bound: wider bought nothing); the pool holds up to kMaxWorkers threads, i.e. four loops at full width.
class HostPool {
public:
    static HostPool& get() { static HostPool p; return p; }
    static constexpr unsigned kWidth = 16;        // parts of one loop = threads a lone loop runs on
    static constexpr unsigned kMaxWorkers = 63;   // pool threads (beside the callers)
    static constexpr unsigned kPartsPerThread = 4;
    template <class F>
    void run(size_t n, F& fn)
    {
        if (workers_.empty() || n < 2) { fn((size_t)0, n); return; }
        // up to `width_` threads, four parts each: whoever is quick takes more of them (a loop used to wait for its slowest sixteenth --
        // on a busy 256-thread host one descheduled worker doubled a phase of the call)
        const unsigned threads = (unsigned)std::min<size_t>(std::min<size_t>(width_, workers_.size() + 1), n);
        const unsigned parts = (unsigned)std::min<size_t>((size_t)threads * kPartsPerThread, n);
        std::function<void(unsigned)> body = [&fn, n, parts](unsigned k) { fn(n * k / parts, n * (k + 1) / parts); };
        Job job;
        job.body = &body; job.parts = parts;
        {
            std::lock_guard<std::mutex> g(m_);
            open_.push_back(&job);
        }
        for (unsigned k = 1; k < threads; ++k) cv_work_.notify_one();
        // the caller works on ITS job only (another caller's job may be long: this one must not wait for it)
        std::unique_lock<std::mutex> g(m_);
        while (job.next < job.parts) {
            const unsigned k = job.next++;
            if (job.next == job.parts) close(&job);
            g.unlock();
            body(k);
            g.lock();
            ++job.done;
        }
        job.cv.wait(g, [&] { return job.done == job.parts; });
    }
    unsigned workers() const { return (unsigned)workers_.size(); }
private:
    struct Job {
        std::function<void(unsigned)>* body = nullptr;
        unsigned parts = 0, next = 0, done = 0;
        std::condition_variable cv;   // signalled under m_ when done == parts (the job lives on its caller's stack)
    };
    HostPool()
    {
        if (const char* e = std::getenv("GAMDP_HOST_WIDTH")) width_ = (unsigned)std::min(64, std::max(1, std::atoi(e)));   // (A/B; results do not depend on it)
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const unsigned nw = std::min(kMaxWorkers, hw - 1);
        try { for (unsigned k = 0; k < nw; k++) workers_.emplace_back([this] { loop(); }); } catch (...) {}   // fewer threads, or none: still correct
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; }
        cv_work_.notify_all();
        for (auto& t : workers_) if (t.joinable()) t.join();
    }
    void close(Job* j) { open_.erase(std::find(open_.begin(), open_.end(), j)); }   // (m_ held) every part is taken: off the list
    void loop()
    {
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_work_.wait(g, [&] { return stop_ || !open_.empty(); });
            if (stop_) return;
            // jobs in turn: the front job gives a part and goes to the back, so concurrent loops share the workers evenly
            Job* const j = open_.front();
            const unsigned k = j->next++;
            open_.pop_front();
            if (j->next < j->parts) open_.push_back(j);
            std::function<void(unsigned)>* const body = j->body;
            g.unlock();
            (*body)(k);
            g.lock();
            if (++j->done == j->parts) j->cv.notify_all();   // under m_: the caller cannot leave run() (and destroy the job) before we let go
        }
    }
    std::mutex m_;
    std::condition_variable cv_work_;
    std::vector<std::thread> workers_;
    std::deque<Job*> open_;   // jobs with parts left to hand out
    bool stop_ = false;
    unsigned width_ = kWidth;
};

}  // namespace gamdp
