// What do hipEventRecord / hipEventElapsedTime / hipEventQuery cost on the host?  (round 6: 0.6 - 2.4 ms of a batch call went into reading two event pairs)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/event_cost tools/event_cost.hip && /tmp/event_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(unsigned long long ticks, unsigned long long* out)
{
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (out) out[0] = t0;
}
int main()
{
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (unsigned flags : {(unsigned)hipEventDefault, (unsigned)hipEventBlockingSync}) {
        std::vector<hipEvent_t> ev(8);
        for (auto& e : ev) hipEventCreateWithFlags(&e, flags);
        double t_rec = 0, t_sync = 0, t_el = 0, t_el2 = 0;
        const int reps = 10;
        float ms = 0;
        for (int r = 0; r < reps; r++) {
            double t0 = now_ms();
            hipEventRecord(ev[0], st);
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, 100000ull, nullptr);   // 1 ms
            hipEventRecord(ev[1], st);
            hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, st, 100000ull, nullptr);
            hipEventRecord(ev[2], st);
            t_rec += now_ms() - t0;
            t0 = now_ms();
            hipStreamSynchronize(st);
            t_sync += now_ms() - t0;
            t0 = now_ms();
            hipEventElapsedTime(&ms, ev[0], ev[1]);
            t_el += now_ms() - t0;
            t0 = now_ms();
            hipEventElapsedTime(&ms, ev[1], ev[2]);
            t_el2 += now_ms() - t0;
        }
        std::printf("event flags 0x%x: 3 records + 2 launches %.3f ms, sync %.3f ms, first hipEventElapsedTime %.3f ms, second %.3f ms (last interval %.3f ms)\n",
                    flags, t_rec / reps, t_sync / reps, t_el / reps, t_el2 / reps, ms);
    }
    return 0;
}
