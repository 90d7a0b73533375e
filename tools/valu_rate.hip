// Micro-benchmark: issue rate of the integer VALU ops the DP kernel is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    unsigned b = seed | 1, c = seed * 7 + 3;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { REP16(asm volatile("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %3, %3, %1, %2\n v_max3_i32 %4, %4, %1, %2\n v_max3_i32 %5, %5, %1, %2" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP16(asm volatile("v_and_b32 %0, %0, %1\n v_and_b32 %3, %3, %1\n v_and_b32 %4, %4, %1\n v_and_b32 %5, %5, %1" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 2) { REP16(asm volatile("v_dot4_u32_u8 %0, %1, %2, %0\n v_dot4_u32_u8 %3, %1, %2, %3\n v_dot4_u32_u8 %4, %1, %2, %4\n v_dot4_u32_u8 %5, %1, %2, %5" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 3) { REP16(asm volatile("v_alignbit_b32 %0, %1, %0, 2\n v_alignbit_b32 %3, %1, %3, 2\n v_alignbit_b32 %4, %1, %4, 2\n v_alignbit_b32 %5, %1, %5, 2" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 4) { REP16(asm volatile("v_pk_max_i16 %0, %0, %1\n v_pk_max_i16 %3, %3, %1\n v_pk_max_i16 %4, %4, %1\n v_pk_max_i16 %5, %5, %1" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 5) { REP16(asm volatile("v_pk_add_u16 %0, %0, %1\n v_pk_add_u16 %3, %3, %1\n v_pk_add_u16 %4, %4, %1\n v_pk_add_u16 %5, %5, %1" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 6) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %3, %3, %1, %2\n v_perm_b32 %4, %4, %1, %2\n v_perm_b32 %5, %5, %1, %2" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 7) { REP16(asm volatile("v_or_b32 %0, %0, %1\n v_or_b32 %3, %3, %1\n v_or_b32 %4, %4, %1\n v_or_b32 %5, %5, %1" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 8) { REP16(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %3, %3, %1\n v_add_u32 %4, %4, %1\n v_add_u32 %5, %5, %1" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 9) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n v_pk_fma_f32 %2, %2, %1, %2" : "+v"(*(double*)&a0), "+v"(*(double*)&a2), "+v"(*(double*)&a4)); asm volatile("" : "+v"(a6));) }
        if (OP == 10) { REP16(asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %3, %3, %1\n v_max_i32 %4, %4, %1\n v_max_i32 %5, %5, %1" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 11) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %0\n v_fma_f32 %3, %3, %1, %3\n v_fma_f32 %4, %4, %1, %4\n v_fma_f32 %5, %5, %1, %5" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 12) { REP16(asm volatile("v_pk_sub_u16 %0, %0, %1\n v_pk_min_u16 %3, %3, %1\n v_pk_mad_u16 %4, %4, %1, %2\n v_pk_lshlrev_b16 %5, 2, %5" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int OP>
void run(const char* name, int wpe)
{
    unsigned* d;
    const int blocks = 256 * wpe, iters = 4000;
    hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 1);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ninstr = (double)iters * 64 * (OP == 9 ? 0.5 : 1.0);   // per wave
    const double waves_per_simd = wpe;                                  // 256-thread blocks: 4 waves -> one per SIMD
    const double cyc = ms * 1e-3 * 2.4e9;                               // at nominal 2.4 GHz
    printf("%-22s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (@2.4GHz nominal)\n", name, wpe, ms,
           cyc / (ninstr * waves_per_simd));
    hipFree(d);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_max3_i32", w); run<1>("v_and_b32", w); run<2>("v_dot4_u32_u8", w); run<3>("v_alignbit_b32", w);
        run<7>("v_or_b32", w); run<8>("v_add_u32", w); run<10>("v_max_i32", w); run<6>("v_perm_b32", w);
        run<4>("v_pk_max_i16", w); run<5>("v_pk_add_u16", w); run<12>("pk16 mix", w); run<11>("v_fma_f32", w); run<9>("v_pk_fma_f32", w);
    }
    return 0;
}
