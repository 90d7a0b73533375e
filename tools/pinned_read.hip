// How fast does the CPU read pinned host memory the GPU has just written?  (round 6: the results phase of a batch call)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pinned_read tools/pinned_read.hip && /tmp/pinned_read
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t n = 100000, B = 40;
    void* d = nullptr;
    hipMalloc(&d, n * B);
    hipMemset(d, 1, n * B);
    std::vector<char> dst(n * 88);
    for (unsigned flags : {(unsigned)hipHostMallocDefault, (unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocCoherent, (unsigned)hipHostMallocNumaUser}) {
        void* h = nullptr;
        if (hipHostMalloc(&h, n * B, flags) != hipSuccess) { std::printf("flags %u: alloc failed\n", flags); continue; }
        double tc = 0, tr = 0;
        for (int rep = 0; rep < 10; rep++) {
            double t0 = now_ms();
            hipMemcpyAsync(h, d, n * B, hipMemcpyDeviceToHost, 0);
            hipStreamSynchronize(0);
            tc += now_ms() - t0;
            t0 = now_ms();
            for (size_t i = 0; i < n; i++) { std::memset(&dst[i * 88], 0, 88); std::memcpy(&dst[i * 88], (char*)h + i * B, B); }
            tr += now_ms() - t0;
        }
        std::printf("hipHostMalloc flags 0x%x: D2H of %zu KB %.3f ms, one thread converts %zu records in %.3f ms\n", flags, n * B / 1024, tc / 10, n, tr / 10);
        hipHostFree(h);
    }
    {
        std::vector<char> h(n * B, 1);
        double t0 = now_ms();
        for (int rep = 0; rep < 10; rep++)
            for (size_t i = 0; i < n; i++) { std::memset(&dst[i * 88], 0, 88); std::memcpy(&dst[i * 88], h.data() + i * B, B); }
        std::printf("malloc'd source: %.3f ms\n", (now_ms() - t0) / 10);
    }
    return 0;
}
