// Micro-benchmark 2: which integer/float VALU ops issue at the full rate on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
#define BODY(INS) { REP16(asm volatile(INS : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));) }
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    unsigned b = seed | 1, c = seed * 7 + 3;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) BODY("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %3, %3, %1, %2\n v_max3_f32 %4, %4, %1, %2\n v_max3_f32 %5, %5, %1, %2")
        if (OP == 1) BODY("v_max_f32 %0, %0, %1\n v_max_f32 %3, %3, %1\n v_max_f32 %4, %4, %1\n v_max_f32 %5, %5, %1")
        if (OP == 2) BODY("v_lshl_or_b32 %0, %0, 2, %1\n v_lshl_or_b32 %3, %3, 2, %1\n v_lshl_or_b32 %4, %4, 2, %1\n v_lshl_or_b32 %5, %5, 2, %1")
        if (OP == 3) BODY("v_and_or_b32 %0, %0, %1, %2\n v_and_or_b32 %3, %3, %1, %2\n v_and_or_b32 %4, %4, %1, %2\n v_and_or_b32 %5, %5, %1, %2")
        if (OP == 4) BODY("v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %3, %3, %1, %2\n v_bfi_b32 %4, %4, %1, %2\n v_bfi_b32 %5, %5, %1, %2")
        if (OP == 5) BODY("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %3, %3, %1, vcc\n v_cndmask_b32 %4, %4, %1, vcc\n v_cndmask_b32 %5, %5, %1, vcc")
        if (OP == 6) BODY("v_add3_u32 %0, %0, %1, %2\n v_add3_u32 %3, %3, %1, %2\n v_add3_u32 %4, %4, %1, %2\n v_add3_u32 %5, %5, %1, %2")
        if (OP == 7) BODY("v_lshl_add_u32 %0, %0, 2, %1\n v_lshl_add_u32 %3, %3, 2, %1\n v_lshl_add_u32 %4, %4, 2, %1\n v_lshl_add_u32 %5, %5, 2, %1")
        if (OP == 8) BODY("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %3, %3, %1, %2\n v_mad_u32_u24 %4, %4, %1, %2\n v_mad_u32_u24 %5, %5, %1, %2")
        if (OP == 9) BODY("v_cmp_eq_u32 vcc, %0, %1\n v_cmp_eq_u32 vcc, %3, %1\n v_cmp_eq_u32 vcc, %4, %1\n v_cmp_eq_u32 vcc, %5, %1")
        if (OP == 10) BODY("v_mov_b32_dpp %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
        if (OP == 11) BODY("v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %3, 2, %3\n v_lshlrev_b32 %4, 2, %4\n v_lshlrev_b32 %5, 2, %5")
        if (OP == 12) BODY("v_bfe_u32 %0, %0, 2, 2\n v_bfe_u32 %3, %3, 2, 2\n v_bfe_u32 %4, %4, 2, 2\n v_bfe_u32 %5, %5, 2, 2")
        if (OP == 13) BODY("v_xor_b32 %0, %0, %1\n v_xor_b32 %3, %3, %1\n v_xor_b32 %4, %4, %1\n v_xor_b32 %5, %5, %1")
        if (OP == 14) BODY("v_min_u32 %0, %0, %1\n v_min_u32 %3, %3, %1\n v_min_u32 %4, %4, %1\n v_min_u32 %5, %5, %1")
        if (OP == 15) BODY("v_sub_u32 %0, %0, %1\n v_sub_u32 %3, %3, %1\n v_sub_u32 %4, %4, %1\n v_sub_u32 %5, %5, %1")
        if (OP == 16) BODY("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %3, %3, %1, %2\n v_med3_f32 %4, %4, %1, %2\n v_med3_f32 %5, %5, %1, %2")
        if (OP == 17) BODY("v_pk_max_f16 %0, %0, %1\n v_pk_max_f16 %3, %3, %1\n v_pk_max_f16 %4, %4, %1\n v_pk_max_f16 %5, %5, %1")
        if (OP == 18) BODY("v_addc_co_u32 %0, vcc, %0, %0, vcc\n v_addc_co_u32 %3, vcc, %3, %3, vcc\n v_addc_co_u32 %4, vcc, %4, %4, vcc\n v_addc_co_u32 %5, vcc, %5, %5, vcc")
        if (OP == 19) BODY("v_max3_u32 %0, %0, %1, %2\n v_max3_u32 %3, %3, %1, %2\n v_max3_u32 %4, %4, %1, %2\n v_max3_u32 %5, %5, %1, %2")
        if (OP == 20) BODY("v_add_f32 %0, %0, %1\n v_add_f32 %3, %3, %1\n v_add_f32 %4, %4, %1\n v_add_f32 %5, %5, %1")
        if (OP == 21) BODY("v_max3_i16 %0, %0, %1, %2\n v_max3_i16 %3, %3, %1, %2\n v_max3_i16 %4, %4, %1, %2\n v_max3_i16 %5, %5, %1, %2")
        if (OP == 22) BODY("v_sad_u8 %0, %0, %1, %2\n v_sad_u8 %3, %3, %1, %2\n v_sad_u8 %4, %4, %1, %2\n v_sad_u8 %5, %5, %1, %2")
        if (OP == 23) BODY("v_and_b32 %0, %0, %1\n v_max3_f32 %3, %3, %1, %2\n v_or_b32 %4, %4, %1\n v_dot4_u32_u8 %5, %1, %2, %5")
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
}
template <int OP>
void run(const char* name)
{
    unsigned* d;
    const int wpe = 4, blocks = 256 * wpe, iters = 4000;
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
    printf("%-18s %.3f ms -> %.2f nominal cycles / wave-instr / SIMD (4 waves/SIMD)\n", name, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wpe));
    hipFree(d);
}
int main()
{
    run<13>("v_xor_b32"); run<0>("v_max3_f32"); run<1>("v_max_f32"); run<16>("v_med3_f32"); run<20>("v_add_f32"); run<19>("v_max3_u32");
    run<2>("v_lshl_or_b32"); run<3>("v_and_or_b32"); run<4>("v_bfi_b32"); run<5>("v_cndmask_b32"); run<6>("v_add3_u32");
    run<7>("v_lshl_add_u32"); run<8>("v_mad_u32_u24"); run<9>("v_cmp_eq_u32"); run<10>("v_mov_dpp wshr"); run<11>("v_lshlrev_b32");
    run<12>("v_bfe_u32"); run<14>("v_min_u32"); run<15>("v_sub_u32"); run<17>("v_pk_max_f16"); run<18>("v_addc_co_u32"); run<21>("v_max3_i16");
    run<22>("v_sad_u8"); run<23>("mix and/max3f/or/dot4");
    return 0;
}
