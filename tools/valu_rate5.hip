// Micro-benchmark 5 (round 2): the direction-free cell as it is (v_dot4_u32_u8 + v_max3_i32, one cell per lane and
// instruction pair) against a packed-f16 cell that carries TWO tasks per lane (v_perm_b32 + v_pk_fma_f16 +
// v_pk_maximum3_f16 per cell PAIR), each with the dependency structure of the row sweep (17 columns per lane, left
// chain through the max, two DPP hand-offs per row) and 5 waves per SIMD.  Prints ns per cell per lane-column-row.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate5.hip -o tools/valu_rate5 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 as_h2(u32 x) { return __builtin_bit_cast(h2, x); }
__device__ __forceinline__ u32 as_u(h2 x) { return __builtin_bit_cast(u32, x); }
constexpr int C = 17, ROWS = 16;

template <int V>
__global__ __launch_bounds__(64, 5) void k(u32* out, const u32* in, int blocks16)
{
    const int lane = threadIdx.x;
    u32 Lp[C], W[C + ROWS - 1];
#pragma unroll
    for (int c = 0; c < C; ++c) Lp[c] = in[lane + 64 * c];
#pragma unroll
    for (int k2 = 0; k2 < C + ROWS - 1; ++k2) W[k2] = in[lane * 3 + k2];
    u32 Lin = in[lane + 7], x = in[lane + 9], dl = in[lane + 11], dr = in[lane + 13];
    const h2 three = {(_Float16)3.0f, (_Float16)3.0f};
    for (int b = 0; b < blocks16; ++b) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const u32 browA = in[(b * 16 + r) & 1023], browB = in[((b * 16 + r) & 1023) + 1024];
            u32 L = Lin;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const u32 U = (c < C - 1) ? Lp[c + 1] : x;
                if (V == 0) {
                    const int D = (int)__builtin_amdgcn_udot4(W[r + c], browA, Lp[c], false);
                    L = (u32)max(max(D, (int)U), (int)L);
                } else {
                    const u32 m = __builtin_amdgcn_perm(browB, browA, W[r + c]);
                    const h2 D = __builtin_elementwise_fma(as_h2(m), three, as_h2(Lp[c]));
                    L = as_u(__builtin_elementwise_maximum(__builtin_elementwise_maximum(D, as_h2(U)), as_h2(L)));
                }
                Lp[c] = L;
                if (c == 0) {
                    x = (u32)__builtin_amdgcn_update_dpp((int)x, (int)Lp[0], 0x130, 0xf, 0xf, false);
                    if (V == 1) x = as_u(as_h2(x) + as_h2(dr));
                }
            }
            Lin = (u32)__builtin_amdgcn_update_dpp((int)Lin, (int)L, 0x138, 0xf, 0xf, false);
            if (V == 1) Lin = as_u(as_h2(Lin) + as_h2(dl));
        }
#pragma unroll
        for (int k2 = 0; k2 < C - 1; ++k2) W[k2] = W[k2 + ROWS];
#pragma unroll
        for (int k2 = C - 1; k2 < C + ROWS - 1; ++k2) W[k2] = in[(b + k2 + lane) & 2047];
    }
    u32 s = Lin + x;
#pragma unroll
    for (int c = 0; c < C; ++c) s += Lp[c];
    out[blockIdx.x * 64 + lane] = s;
}

template <int V>
void run(const char* name, u32* out, u32* in)
{
    const int blocks16 = 2000, grid = 256 * 20;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<V><<<grid, 64>>>(out, in, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<V><<<grid, 64>>>(out, in, blocks16);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double cells = (double)grid * 64 * C * ROWS * blocks16 * (V == 1 ? 2 : 1);
    printf("%-34s %8.2f ms  %7.2f Tcells/s (%d task%s per lane)\n", name, ms, cells / ms / 1e9, V == 1 ? 2 : 1, V == 1 ? "s" : "");
}

int main()
{
    u32 *out, *in;
    (void)hipMalloc(&out, 256 * 20 * 64 * 4);
    (void)hipMalloc(&in, 1 << 20);
    (void)hipMemset(in, 0x11, 1 << 20);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("dot4 + max3 (i32, 1 task)", out, in);
        run<1>("perm + pk_fma + pk_maximum3 (f16x2)", out, in);
    }
    return 0;
}
