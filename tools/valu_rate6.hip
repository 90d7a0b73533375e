// Micro-benchmark 6 (round 3): the packed-f16 row sweep of kernel_pair.inc at 1..5 wavefronts per SIMD, in two
// instruction orders:
//   V = 1  the source order of round 2 (cell by cell: v_perm -> v_pk_fma -> v_pk_maximum3, every instruction depends on
//          the one before it; the compiler pads the hazards with s_nop)
//   V = 2  software-pipelined by hand: max3 of cell g, fma of cell g+1, perm of cell g+2 per scheduling group
//          (__builtin_amdgcn_sched_barrier between groups), so no instruction depends on its predecessor
// Question: what does a SIMD with 1 / 2 / 3 wavefronts issue?  (The tail of a short launch and the merge-block rounds
// run at 1-2 wavefronts per SIMD.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate6.hip -o tools/valu_rate6 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2 as_h2(u32 x) { return __builtin_bit_cast(h2, x); }
__device__ __forceinline__ u32 as_u(h2 x) { return __builtin_bit_cast(u32, x); }
constexpr int C = 17, ROWS = 16;
constexpr int TN = 256;   // (small tables: 3 KB of LDS per wavefront, so that 5 wavefronts per SIMD fit a CU)
__shared__ u32 s_tab[2][TN + 16];
__shared__ u32 s_w[TN + 16];

template <int V>
__global__ __launch_bounds__(64, 5) void k(u32* out, const u32* in, int blocks16)
{
    const int lane = threadIdx.x;
    for (int i = lane; i < TN + 16; i += 64) { s_tab[0][i] = in[i]; s_tab[1][i] = in[i + 4096]; s_w[i] = in[i + 8192]; }
    __syncthreads();
    u32 Lp[C], W[C + ROWS - 1];
#pragma unroll
    for (int c = 0; c < C; ++c) Lp[c] = in[lane + 64 * c];
#pragma unroll
    for (int k2 = 0; k2 < C + ROWS - 1; ++k2) W[k2] = in[lane * 3 + k2];
    u32 Lin = in[lane + 7], xk = in[lane + 9], dl = in[lane + 11], dr = in[lane + 13];
    const h2 three = {(_Float16)3.0f, (_Float16)3.0f};
    for (int b = 0; b < blocks16; ++b) {
        const u32* ta = &s_tab[0][(b * 16 + lane) & (TN - 1)];
        const u32* tb = &s_tab[1][(b * 16 + lane) & (TN - 1)];
        const u32* wa = &s_w[(b * 16 + 3 * lane) & (TN - 1)];
        if constexpr (V == 1) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                W[C - 1 + r] = wa[r];
                const u32 browA = ta[r], browB = tb[r];
                u32 L = Lin, x = 0xFC00FC00u;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const u32 U = (c < C - 1) ? Lp[c + 1] : x;
                    const u32 m = __builtin_amdgcn_perm(browB, browA, W[r + c]);
                    const h2 D = __builtin_elementwise_fma(as_h2(m), three, as_h2(Lp[c]));
                    L = as_u(__builtin_elementwise_maximum(__builtin_elementwise_maximum(D, as_h2(U)), as_h2(L)));
                    Lp[c] = L;
                    if (c == 0) {
                        xk = (u32)__builtin_amdgcn_update_dpp((int)xk, (int)Lp[0], 0x130, 0xf, 0xf, false);
                        x = as_u(as_h2(xk) + as_h2(dr));
                    }
                }
                Lin = (u32)__builtin_amdgcn_update_dpp((int)Lin, (int)L, 0x138, 0xf, 0xf, false);
                Lin = as_u(as_h2(Lin) + as_h2(dl));
            }
        } else {
            // flat cell index g = r * C + c; stage s of group g: max3(g), fma(g + 1), perm(g + 2)
            u32 tA[ROWS], tB[ROWS];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { W[C - 1 + r] = wa[r]; tA[r] = ta[r]; tB[r] = tb[r]; }
            u32 m[3];   // perm results in flight (cell g % 3)
            u32 D[2];   // fma results in flight
            u32 L = Lin, x = 0xFC00FC00u;
            auto do_perm = [&](const int g) __attribute__((always_inline)) {
                const int r = g / C, c = g % C;
                m[g % 3] = __builtin_amdgcn_perm(tB[r], tA[r], W[r + c]);
            };
            auto do_fma = [&](const int g) __attribute__((always_inline)) {
                const int c = g % C;
                D[g % 2] = as_u(__builtin_elementwise_fma(as_h2(m[g % 3]), three, as_h2(Lp[c])));
            };
            do_perm(0); do_perm(1); do_fma(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int g = r * C + c;
                const u32 U = (c < C - 1) ? Lp[(c < C - 1) ? c + 1 : c] : x;
                if (c == 0) L = Lin;
                L = as_u(__builtin_elementwise_maximum(__builtin_elementwise_maximum(as_h2(D[g % 2]), as_h2(U)), as_h2(L)));
                // fma(g+1) reads the OLD Lp[c+1] (as does the max3 above, as its `up` source); Lp[c] is written after
                if (g + 1 < ROWS * C) do_fma(g + 1);
                Lp[c] = L;
                if (g + 2 < ROWS * C) do_perm(g + 2);
                if (c == 2) {   // the hand-off of column 0, two groups after its max3 (DPP needs two wait states)
                    xk = (u32)__builtin_amdgcn_update_dpp((int)xk, (int)Lp[0], 0x130, 0xf, 0xf, false);
                }
                if (c == 4) x = as_u(as_h2(xk) + as_h2(dr));
                __builtin_amdgcn_sched_barrier(0);
                if (c == C - 1) {
                    Lin = (u32)__builtin_amdgcn_update_dpp((int)Lin, (int)L, 0x138, 0xf, 0xf, false);
                    Lin = as_u(as_h2(Lin) + as_h2(dl));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            }
        }
#pragma unroll
        for (int k2 = 0; k2 < C - 1; ++k2) W[k2] = W[k2 + ROWS];
    }
    u32 s = Lin + xk;
#pragma unroll
    for (int c = 0; c < C; ++c) s += Lp[c];
    out[blockIdx.x * 64 + lane] = s;
}

template <int V>
void run(const char* name, u32* out, u32* in, int waves_per_simd)
{
    const int blocks16 = 1500, grid = 256 * 4 * waves_per_simd;
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
    const double cells = (double)grid * 64 * C * ROWS * blocks16 * 2;
    const double row_ns = ms * 1e6 / ((double)ROWS * blocks16);
    printf("%-28s %d waves/SIMD %8.2f ms  %7.2f Tcells/s  %7.1f ns per row-time of a wave\n", name, waves_per_simd, ms, cells / ms / 1e9, row_ns);
}

int main()
{
    u32 *out, *in;
    (void)hipMalloc(&out, 256 * 32 * 64 * 4);
    (void)hipMalloc(&in, 1 << 20);
    (void)hipMemset(in, 0x11, 1 << 20);
    for (int w = 1; w <= 5; ++w) {
        run<1>("source order (round 2)", out, in, w);
        run<2>("software-pipelined", out, in, w);
    }
    return 0;
}
