// ThreadSanitizer test of the host thread pool (gam_ngs_amd/csrc/gamdp_hostpool.h): built and run by tests/test_hostpool.py.
// Several caller threads run parallel loops at the same time (each posts its loop as a job; the workers serve the jobs in turn),
// loops of every size follow each other back to back, and every element must be visited exactly once.  Then: two LARGE loops of
// two callers must be in flight at the same time (the host phases of the devices of a gamdp_multi call overlap) -- each loop's
// first part waits until it has seen the other loop running.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <numeric>
#include <thread>
#include <vector>

#include "gamdp_hostpool.h"

int main()
{
    using gamdp::HostPool;
    std::atomic<long> bad{0};
    auto caller = [&](unsigned seed) {
        std::vector<unsigned> hits;
        for (int rep = 0; rep < 300; ++rep) {
            const size_t n = 1 + (size_t)((seed * 2654435761u + (unsigned)rep * 40503u) % 50000u);
            hits.assign(n, 0);
            auto body = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) hits[i] += 1; };
            HostPool::get().run(n, body);
            for (size_t i = 0; i < n; ++i) if (hits[i] != 1) { bad++; break; }
        }
    };
    std::vector<std::thread> th;
    for (unsigned k = 0; k < 6; ++k) th.emplace_back(caller, k + 1);
    for (auto& t : th) t.join();
    // two callers, large loops: both must be running at once
    std::atomic<int> started[2] = {{0}, {0}};
    std::atomic<long> overlap_missed{0};
    auto big_caller = [&](int me) {
        std::vector<unsigned> hits((size_t)1 << 18, 0);
        auto body = [&](size_t lo, size_t hi) {
            started[me] = 1;
            if (lo == 0) {   // the loop's first part: wait (up to 20 s) until the other caller's loop runs too
                const auto t0 = std::chrono::steady_clock::now();
                while (!started[1 - me].load() && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(20)) std::this_thread::yield();
                if (!started[1 - me].load()) overlap_missed++;
            }
            for (size_t i = lo; i < hi; ++i) hits[i] += 1;
        };
        HostPool::get().run(hits.size(), body);
        for (size_t i = 0; i < hits.size(); ++i) if (hits[i] != 1) { bad++; break; }
    };
    {
        std::thread a(big_caller, 0), b(big_caller, 1);
        a.join(); b.join();
    }
    // many callers with large loops at once (more jobs than the pool has width for): all complete, all elements once
    {
        std::vector<std::thread> many;
        for (unsigned k = 0; k < 8; ++k) many.emplace_back([&, k] {
            for (int rep = 0; rep < 20; ++rep) {
                std::vector<unsigned> hits(40000 + 977 * k + (size_t)rep, 0);
                auto body = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) hits[i] += 1; };
                HostPool::get().run(hits.size(), body);
                for (size_t i = 0; i < hits.size(); ++i) if (hits[i] != 1) { bad++; break; }
            }
        });
        for (auto& t : many) t.join();
    }
    // one caller alone: the pool must actually spread the work (more than one thread id seen on a big loop; a caller that is quick
    // may take every part before a worker has woken up, so a few tries)
    size_t distinct = 0;
    for (int rep = 0; rep < 8 && distinct < 2; ++rep) {
        std::vector<std::thread::id> who(1 << 20);
        std::vector<double> sink(who.size());
        auto body = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; ++i) { who[i] = std::this_thread::get_id(); double x = (double)i; for (int k = 0; k < 20; ++k) x = x * 1.0000001 + 0.5; sink[i] = x; } };
        HostPool::get().run(who.size(), body);
        std::sort(who.begin(), who.end());
        distinct = (size_t)(std::unique(who.begin(), who.end()) - who.begin());
    }
    std::printf("bad %ld overlap_missed %ld distinct_threads %zu workers %u hardware %u\n", bad.load(), overlap_missed.load(), distinct,
                HostPool::get().workers(), std::thread::hardware_concurrency());
    return (bad.load() == 0 && overlap_missed.load() == 0 && (distinct > 1 || std::thread::hardware_concurrency() < 2)) ? 0 : 1;
}
