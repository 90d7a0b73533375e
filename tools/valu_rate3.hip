// Micro-benchmark 3: what does the 5-instruction DP cell really cost on gfx950, as a function of the
// instruction ORDER inside a wave and of the number of waves per SIMD?  Each variant runs the dependency
// structure of the kernel's row sweep (16 columns per lane, left chain through max3 -> and -> max3) with a
// hand-fixed schedule (volatile asm keeps the order).  Results are printed in cycles per cell per SIMD,
// calibrated against a pure v_max3_i32 stream (4 cycles per wave-instruction) measured in the same run.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define DOT4(d, w, b, l) asm volatile("v_dot4_u32_u8 %0, %1, %2, %3" : "=v"(d) : "v"(w), "v"(b), "v"(l))
#define DOT8(d, w, b, l) asm volatile("v_dot8_u32_u4 %0, %1, %2, %3" : "=v"(d) : "v"(w), "v"(b), "v"(l))
#define PERMADD(d, w, b, l) asm volatile("v_perm_b32 %0, %2, %2, %1\n v_add_u32 %0, %0, %3" : "=&v"(d) : "v"(w), "v"(b), "v"(l))
#define OR1(d, s) asm volatile("v_or_b32 %0, 1, %1" : "=v"(d) : "v"(s))
#define MAX3(d, a, b, c) asm volatile("v_max3_i32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c))
#define ALB(acc, r) asm volatile("v_alignbit_b32 %0, %1, %0, 2" : "+v"(acc) : "v"(r))
#define AND4(d, r) asm volatile("v_and_b32 %0, -4, %1" : "=v"(d) : "v"(r))
#define SHR1(d, s) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s))
#define SHL1(d, s) asm volatile("s_nop 1\n v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(s))

constexpr int C = 16;

// VARIANT 0: per cell  dot4(c+2) or(c+1) max3(c) alignbit(c) and(c)      -- B A B B A (close to what the compiler emits)
// VARIANT 1: per cell  max3(c) and(c) or(c+2) dot4(c+2) alignbit(c)       -- B A A B B (the two full-rate ops adjacent)
// VARIANT 2: per row   all dot4, all or, then per cell max3 and, then all alignbit  (phases)
// VARIANT 3: per cell  max3(c) and(c) alignbit(c-1) dot4(c+2) or(c+2)     -- chain first, A ops split
// VARIANT 4: as 1 but without the alignbit (what a direction-free fill would cost)
// VARIANT 5: as 1 but without the or (pre-tagged up source)
template <int VARIANT>
__global__ __launch_bounds__(256) void k(unsigned* out, int rows, unsigned seed)
{
    unsigned Lp[C], acc[C], W[C], D[C], U[C], R[C];
    unsigned brow = seed * 0x01010101u, Lin = seed, x = seed;
#pragma unroll
    for (int c = 0; c < C; ++c) { Lp[c] = threadIdx.x * 4 + c * 64 + seed; acc[c] = 0; W[c] = 1u << (8 * ((threadIdx.x + c) & 3)); D[c] = U[c] = R[c] = 0; }
    for (int r = 0; r < rows; ++r) {
        if (VARIANT == 0) {
            DOT4(D[0], W[0], brow, Lp[0]); DOT4(D[1], W[1], brow, Lp[1]); OR1(U[0], Lp[1]);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (c + 2 < C) DOT4(D[c + 2], W[c + 2], brow, Lp[c + 2]);
                if (c + 1 < C) OR1(U[c + 1], (c + 2 < C) ? Lp[c + 2] : x);
                MAX3(R[c], D[c], U[c], (c == 0) ? Lin : Lp[(c + C - 1) % C]);
                ALB(acc[c], R[c]);
                AND4(Lp[c], R[c]);
                if (c == 0) SHL1(x, Lp[0]);
            }
        } else if (VARIANT == 1 || VARIANT == 4 || VARIANT == 5 || VARIANT == 6 || VARIANT == 7) {
            DOT4(D[0], W[0], brow, Lp[0]); DOT4(D[1], W[1], brow, Lp[1]); OR1(U[0], Lp[1]); OR1(U[1], Lp[2]);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                MAX3(R[c], D[c], (VARIANT == 5) ? Lp[(c + 1) % C] : U[c], (c == 0) ? Lin : Lp[(c + C - 1) % C]);
                AND4(Lp[c], R[c]);
                if (VARIANT != 5 && c + 2 < C) OR1(U[c + 2], (c + 3 < C) ? Lp[c + 3] : x);
                if (c + 2 < C) {
                    if (VARIANT == 6) DOT8(D[c + 2], W[c + 2], brow, Lp[c + 2]);
                    else if (VARIANT == 7) PERMADD(D[c + 2], W[c + 2], brow, Lp[c + 2]);
                    else DOT4(D[c + 2], W[c + 2], brow, Lp[c + 2]);
                }
                if (VARIANT != 4) ALB(acc[c], R[c]);
                if (c == 0) SHL1(x, Lp[0]);
            }
        } else if (VARIANT == 2) {
#pragma unroll
            for (int c = 0; c < C; ++c) DOT4(D[c], W[c], brow, Lp[c]);
#pragma unroll
            for (int c = 0; c + 1 < C; ++c) OR1(U[c], Lp[c + 1]);
            OR1(U[C - 1], x);
#pragma unroll
            for (int c = 0; c < C; ++c) { MAX3(R[c], D[c], U[c], (c == 0) ? Lin : Lp[(c + C - 1) % C]); AND4(Lp[c], R[c]); }
            SHL1(x, Lp[0]);
#pragma unroll
            for (int c = 0; c < C; ++c) ALB(acc[c], R[c]);
        } else if (VARIANT == 3) {
            DOT4(D[0], W[0], brow, Lp[0]); DOT4(D[1], W[1], brow, Lp[1]); OR1(U[0], Lp[1]); OR1(U[1], Lp[2]);
#pragma unroll
            for (int c = 0; c < C; ++c) {
                MAX3(R[c], D[c], U[c], (c == 0) ? Lin : Lp[(c + C - 1) % C]);
                AND4(Lp[c], R[c]);
                if (c > 0) ALB(acc[c - 1], R[c - 1]);
                if (c + 2 < C) DOT4(D[c + 2], W[c + 2], brow, Lp[c + 2]);
                if (c + 2 < C) OR1(U[c + 2], (c + 3 < C) ? Lp[c + 3] : x);
                if (c == 0) SHL1(x, Lp[0]);
            }
            ALB(acc[C - 1], R[C - 1]);
        }
        SHR1(Lin, Lp[C - 1]);
    }
    unsigned s = Lin + x;
#pragma unroll
    for (int c = 0; c < C; ++c) s += Lp[c] + acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__device__ long long g_clk[2];
__global__ __launch_bounds__(256) void k_cal(unsigned* out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed | 1, c = seed * 7 + 3;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %3, %3, %1, %2\n v_max3_i32 %4, %4, %1, %2\n v_max3_i32 %5, %5, %1, %2"
                         : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = c1 - c0; g_clk[1] = w1 - w0; }
}

static double cyc_per_ms = 0;  // SIMD cycles per millisecond, from the calibration stream

template <class F>
float timed(F f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int V>
void run(const char* name, unsigned* d)
{
    printf("%-46s", name);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        const int rows = 20000;
        float ms = timed([&] { k<V><<<256 * w, 256>>>(d, rows, 1); });
        const double cells = (double)rows * C * w;  // per SIMD, in units of 64-lane cell groups
        printf("  w%d:%6.2f", w, ms * cyc_per_ms / cells);
    }
    printf("   cycles / 64 cells / SIMD\n");
}

int main()
{
    unsigned* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    {
        const int iters = 4000, w = 4;
        float ms = timed([&] { k_cal<<<256 * w, 256>>>(d, iters, 1); });
        cyc_per_ms = 4.0 * (double)iters * 64 * w / ms;  // v_max3_i32 = 4 cycles per wave-instruction
        printf("calibration: %.3f GHz effective (v_max3_i32 := 4 cycles)\n", cyc_per_ms * 1e-6);
        long long h[2];
        hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof h);
        printf("s_memtime ticks %lld, s_memrealtime ticks %lld (100 MHz) -> s_memtime runs at %.3f GHz; kernel %.3f ms\n", h[0], h[1],
               (double)h[0] / ((double)h[1] / 100e6) * 1e-9, ms);
    }
    run<0>("0: dot4 or max3 alignbit and   (B A B B A)", d);
    run<1>("1: max3 and or dot4 alignbit   (B A A B B)", d);
    run<2>("2: phases (all dot4, all or, chain, all alb)", d);
    run<3>("3: max3 and alb dot4 or        (B A B B A')", d);
    run<4>("4: variant 1 without alignbit", d);
    run<5>("5: variant 1 without or", d);
    run<6>("6: variant 1 with v_dot8_u32_u4", d);
    run<7>("7: variant 1 with v_perm + v_add (N-aware)", d);
    return 0;
}
