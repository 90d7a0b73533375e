// ThreadSanitizer test of the host thread pool (gam_ngs_amd/csrc/gamdp_hostpool.h): built and run by tests/test_hostpool.py.
// Several caller threads run parallel loops at the same time (one gets the pool, the others run their loop themselves), loops of
// every size follow each other back to back, and every element must be visited exactly once.
#include <atomic>
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
    std::printf("bad %ld distinct_threads %zu hardware %u\n", bad.load(), distinct, std::thread::hardware_concurrency());
    return (bad.load() == 0 && (distinct > 1 || std::thread::hardware_concurrency() < 2)) ? 0 : 1;
}
