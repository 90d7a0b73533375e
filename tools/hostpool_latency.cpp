// How long does a parallel loop on the host pool take to get going?  (gam_ngs_amd/csrc/gamdp_hostpool.h; round 6)
//   g++ -O2 -std=c++17 -pthread -I gam_ngs_amd/csrc tools/hostpool_latency.cpp -o /tmp/hostpool_latency && /tmp/hostpool_latency
// Prints, for idle gaps of 0 / 0.1 / 1 / 12 ms between loops, the wall time of (a) an empty loop of 100 000 elements and (b) a loop
// that streams 100 000 x 128 bytes (what the results phase of a batch call does).
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "gamdp_hostpool.h"

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    using gamdp::HostPool;
    const size_t n = 100000;
    std::vector<char> src(n * 40), dst(n * 88);
    auto empty = [&](size_t, size_t) {};
    auto stream = [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) { std::memset(&dst[i * 88], 0, 88); std::memcpy(&dst[i * 88], &src[i * 40], 40); } };
    HostPool::get().run(n, stream);
    for (double gap_ms : {0.0, 0.1, 1.0, 12.0}) {
        double te = 0, ts = 0;
        const int reps = 20;
        for (int r = 0; r < reps; r++) {
            if (gap_ms > 0) std::this_thread::sleep_for(std::chrono::microseconds((long)(gap_ms * 1000)));
            double t0 = now_ms();
            HostPool::get().run(n, empty);
            te += now_ms() - t0;
            if (gap_ms > 0) std::this_thread::sleep_for(std::chrono::microseconds((long)(gap_ms * 1000)));
            t0 = now_ms();
            HostPool::get().run(n, stream);
            ts += now_ms() - t0;
        }
        std::printf("gap %5.1f ms: empty loop %.3f ms, streaming loop %.3f ms (workers %u)\n", gap_ms, te / reps, ts / reps, HostPool::get().workers());
    }
    double t0 = now_ms();
    for (int r = 0; r < 20; r++) stream(0, n);
    std::printf("streaming loop on the caller alone: %.3f ms\n", (now_ms() - t0) / 20);
    return 0;
}
