// Micro-benchmark (round 2): issue rate of candidate cell instructions on gfx950 with several waves per SIMD.
// Each kernel issues 4 independent streams of one opcode; prints cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate4.hip -o tools/valu_rate4 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(x) x x x x x x x x x x x x x x x x
#define OP3(name) REP16(asm volatile(name " %0, %0, %1, %2\n " name " %3, %3, %1, %2\n " name " %4, %4, %1, %2\n " name " %5, %5, %1, %2" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));)
#define OP2(name) REP16(asm volatile(name " %0, %0, %1\n " name " %3, %3, %1\n " name " %4, %4, %1\n " name " %5, %5, %1" : "+v"(a0), "+v"(b), "+v"(c), "+v"(a1), "+v"(a2), "+v"(a3));)

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    unsigned b = seed | 1, c = seed * 7 + 3;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) { OP3("v_max3_i32") }
        if (OP == 1) { OP3("v_max3_f32") }
        if (OP == 2) { OP3("v_maximum3_f32") }
        if (OP == 3) { OP2("v_max_f32") }
        if (OP == 4) { OP2("v_max_i32") }
        if (OP == 5) { OP2("v_max_u32") }
        if (OP == 6) { OP3("v_pk_maximum3_f16") }
        if (OP == 7) { OP2("v_pk_max_i16") }
        if (OP == 8) { OP2("v_pk_add_f16") }
        if (OP == 9) { OP3("v_med3_i32") }
        if (OP == 10) { OP3("v_add3_u32") }
        if (OP == 11) { OP3("v_lshl_add_u32") }
        if (OP == 12) { OP3("v_and_or_b32") }
        if (OP == 13) { OP3("v_bfi_b32") }
        if (OP == 14) { OP3("v_sad_u8") }
        if (OP == 15) { OP3("v_mad_u32_u24") }
        if (OP == 16) { OP3("v_dot4_u32_u8") }
        if (OP == 17) { OP3("v_dot8_u32_u4") }
        if (OP == 18) { OP3("v_perm_b32") }
        if (OP == 19) { OP2("v_add_u32") }
        if (OP == 20) { OP2("v_add_f32") }
        if (OP == 21) { OP3("v_fma_f32") }
        if (OP == 22) { OP2("v_min_f32") }
        if (OP == 23) { OP3("v_max3_u32") }
        if (OP == 24) { OP3("v_alignbit_b32") }
        if (OP == 25) { OP2("v_lshlrev_b32") }
        if (OP == 26) { OP3("v_min3_f32") }
        if (OP == 27) { OP3("v_max3_i16") }
        if (OP == 28) { OP2("v_sub_u32") }
        if (OP == 29) { OP2("v_mul_f32") }
        if (OP == 30) { OP3("v_med3_f32") }
        if (OP == 31) { OP2("v_cndmask_b32") }   // uses vcc implicitly
        if (OP == 32) { REP16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(c), "+v"(a2), "+v"(a3));) }
        if (OP == 33) { REP16(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(c), "+v"(a2), "+v"(a3));) }
        if (OP == 34) { REP16(asm volatile("v_max_i32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %2, %3, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %4, %5, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_i32_dpp %1, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(b), "+v"(a1), "+v"(c), "+v"(a2), "+v"(a3));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b + c;
}

static double g_clk_ghz = 2.4;
template <int OP>
void run(const char* name, int wpe)
{
    unsigned* d;
    const int blocks = 256 * wpe, iters = 2000;
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
    const double ninstr = (double)iters * 64;
    const double cyc = ms * 1e-3 * g_clk_ghz * 1e9;
    printf("%-20s w/SIMD=%d %8.3f ms -> %.2f cycles per wave-instruction per SIMD (@%.1f GHz nominal)\n", name, wpe, ms, cyc / (ninstr * wpe), g_clk_ghz);
    hipFree(d);
}

int main()
{
    for (int w : {1, 4, 8}) {
        run<0>("v_max3_i32", w); run<23>("v_max3_u32", w); run<1>("v_max3_f32", w); run<26>("v_min3_f32", w); run<2>("v_maximum3_f32", w);
        run<3>("v_max_f32", w); run<22>("v_min_f32", w); run<4>("v_max_i32", w); run<5>("v_max_u32", w); run<30>("v_med3_f32", w); run<9>("v_med3_i32", w);
        run<6>("v_pk_maximum3_f16", w); run<7>("v_pk_max_i16", w); run<8>("v_pk_add_f16", w); run<27>("v_max3_i16", w);
        run<10>("v_add3_u32", w); run<11>("v_lshl_add_u32", w); run<12>("v_and_or_b32", w); run<13>("v_bfi_b32", w); run<14>("v_sad_u8", w);
        run<15>("v_mad_u32_u24", w); run<16>("v_dot4_u32_u8", w); run<17>("v_dot8_u32_u4", w); run<18>("v_perm_b32", w);
        run<19>("v_add_u32", w); run<28>("v_sub_u32", w); run<20>("v_add_f32", w); run<29>("v_mul_f32", w); run<21>("v_fma_f32", w);
        run<24>("v_alignbit_b32", w); run<25>("v_lshlrev_b32", w); run<31>("v_cndmask_b32", w);
        run<32>("v_mov_dpp wave_shr", w); run<33>("v_mov_dpp row_shr", w); run<34>("v_max_i32_dpp row_shr", w);
        printf("\n");
    }
    return 0;
}
