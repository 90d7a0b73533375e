// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the band-150 strips use
// (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern before trusting an absolute").  Every kernel below touches a KNOWN number of bytes of a buffer far larger
// than the Infinity Cache (so nothing is served on-die), in pieces of one width at one stride; the counters of each kernel
// divided by its byte count are the factors tools/hbm_gran.sh prints.
//   hipcc --offload-arch=gfx950 -O2 -o tools/hbm_gran tools/hbm_gran.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- tools/hbm_gran        (and WRITE_SIZE, TCC_EA0_* in passes of their own)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef uint32_t u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// Every lane moves 16 B.  A "piece" is LP consecutive lanes (LP * 16 B contiguous); pieces lie STRIDE bytes apart; piece p of
// the whole grid is at (perm(p)) * STRIDE where perm scatters neighbouring pieces far apart (so that the two halves of a line,
// when both are touched, are touched by different wave instructions at different times: SCATTER = 1) or keeps them in order.
template <int LP, int STRIDE, bool WRITE, bool SCATTER>
__global__ void k_pieces(u32* buf, const uint64_t n_pieces, u32* sink)
{
    const uint64_t lane_global = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_grid = (uint64_t)gridDim.x * blockDim.x / LP;   // pieces per sweep of the grid
    u32 acc = 0;
    for (uint64_t p = lane_global / LP; p < n_pieces; p += per_grid) {
        uint64_t q = p;
        if (SCATTER) q = (p * 0x9E3779B1ull) % n_pieces;   // (n_pieces is a power of two times an odd number: a permutation when gcd = 1; close enough otherwise)
        u32x4* ptr = (u32x4*)((char*)buf + q * (uint64_t)STRIDE) + (lane_global % LP);
        if (WRITE) { const u32x4 v = {(u32)p, 1u, 2u, 3u}; *ptr = v; }
        else { const u32x4 v = *ptr; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (!WRITE && acc == 0x12345678u) sink[0] = acc;
}

// a lane reads its own 128-B line as eight consecutive 16-B loads (eight wave instructions): how many memory-side requests is a line fetched piecemeal?
__global__ void k_line_by_lane(u32* buf, const uint64_t n_lines, u32* sink)
{
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (uint64_t i = i0; i < n_lines; i += step) {
        const uint64_t q = (i * 0x9E3779B1ull) % n_lines;
        const u32x4* ptr = (const u32x4*)((char*)buf + q * 128ull);
#pragma unroll
        for (int g = 0; g < 8; ++g) { const u32x4 v = ptr[g]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// one dword per lane at a stride (the walk's direction-word reads: 4 B out of a line)
template <int STRIDE, bool WRITE>
__global__ void k_dwords(u32* buf, const uint64_t n, u32* sink)
{
    const uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, step = (uint64_t)gridDim.x * blockDim.x;
    u32 acc = 0;
    for (uint64_t i = i0; i < n; i += step) {
        const uint64_t q = (i * 0x9E3779B1ull) % n;
        u32* ptr = (u32*)((char*)buf + q * (uint64_t)STRIDE);
        if (WRITE) *ptr = (u32)i; else acc += *ptr;
    }
    if (!WRITE && acc == 0x12345678u) sink[0] = acc;
}

// a lane's 80 B (five 16-B stores, as img_store<19, true> writes a direction row), LP lanes contiguous (LP * 80 B), pieces STRIDE apart
template <int LP, int STRIDE, int ROWB>
__global__ void k_rows80(u32* buf, const uint64_t n_pieces)
{
    const uint64_t lane_global = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t per_grid = (uint64_t)gridDim.x * blockDim.x / LP;
    for (uint64_t p = lane_global / LP; p < n_pieces; p += per_grid) {
        const uint64_t q = (p * 0x9E3779B1ull) % n_pieces;
        u32x4* ptr = (u32x4*)((char*)buf + q * (uint64_t)STRIDE + (lane_global % LP) * ROWB);
#pragma unroll
        for (int g = 0; g < 5; ++g) { const u32x4 v = {(u32)p, (u32)g, 2u, 3u}; ptr[g] = v; }
    }
}

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    const uint64_t bytes = 6ull << 30;   // 6 GiB: 24 x the Infinity Cache
    u32 *buf = nullptr, *sink = nullptr;
    CHK(hipMalloc(&buf, bytes));
    CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(buf, 1, bytes));
    CHK(hipDeviceSynchronize());
    const dim3 grid(256 * 16), block(256);
    const uint64_t touched = 1ull << 30;   // every kernel moves 1 GiB of payload
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto report = [&](const char* name, uint64_t payload) -> int {
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-34s payload %.3f GiB  %.3f ms  %.1f GB/s\n", name, payload / 1073741824.0, ms, payload / 1e6 / ms);
        return 0;
    };
#define RUN_PIECES(LP, STRIDE, WR, SC, NAME) do { const uint64_t np = touched / (LP * 16); if (np * (uint64_t)STRIDE > bytes) { std::fprintf(stderr, "buffer too small for %s\n", NAME); return 1; } \
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_pieces<LP, STRIDE, WR, SC>), grid, block, 0, 0, buf, np, sink); if (report(NAME, np * LP * 16)) return 1; } while (0)
    // reads
    RUN_PIECES(64, 1024, false, false, "rd_wide_1024B_inorder");
    RUN_PIECES(8, 128, false, true, "rd_128B_of_128B_scatter");
    RUN_PIECES(4, 128, false, true, "rd_64B_of_128B_scatter");
    RUN_PIECES(4, 64, false, true, "rd_64B_of_64B_scatter");
    RUN_PIECES(2, 128, false, true, "rd_32B_of_128B_scatter");
    RUN_PIECES(2, 64, false, true, "rd_32B_of_64B_scatter");
    RUN_PIECES(2, 32, false, true, "rd_32B_of_32B_scatter");
    RUN_PIECES(1, 64, false, true, "rd_16B_of_64B_scatter");
    RUN_PIECES(10, 256, false, true, "rd_160B_of_256B_scatter");
    { const uint64_t n = touched / 32; static_assert(128ull * ((1ull << 30) / 32) <= (6ull << 30), "inside the buffer"); CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_dwords<128, false>), grid, block, 0, 0, buf, n, sink); if (report("rd_4B_of_128B_scatter", n * 4)) return 1; }
    { const uint64_t n = touched / 128; CHK(hipEventRecord(e0)); hipLaunchKernelGGL(k_line_by_lane, grid, block, 0, 0, buf, n, sink); if (report("rd_128B_line_by_one_lane_8x16B", n * 128)) return 1; }
    // writes
    RUN_PIECES(64, 1024, true, false, "wr_wide_1024B_inorder");
    RUN_PIECES(8, 128, true, true, "wr_128B_of_128B_scatter");
    RUN_PIECES(4, 128, true, true, "wr_64B_of_128B_scatter");
    RUN_PIECES(4, 64, true, true, "wr_64B_of_64B_scatter");
    RUN_PIECES(2, 128, true, true, "wr_32B_of_128B_scatter");
    RUN_PIECES(2, 64, true, true, "wr_32B_of_64B_scatter");
    RUN_PIECES(1, 64, true, true, "wr_16B_of_64B_scatter");
    { const uint64_t np = (bytes - 4096) / 5120; CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_rows80<2, 5120, 80>), grid, block, 0, 0, buf, np); if (report("wr_2x80B_of_5120B_scatter", np * 160)) return 1; }
    { const uint64_t np = (bytes - 4096) / (5120 + 48); CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_rows80<2, 5120 + 48, 80>), grid, block, 0, 0, buf, np); if (report("wr_2x80B_unaligned_scatter", np * 160)) return 1; }
    { const uint64_t np = (bytes - 4096) / 8192; CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_rows80<2, 8192, 128>), grid, block, 0, 0, buf, np); if (report("wr_2x80B_in_128B_rows_scatter", np * 160)) return 1; }
    { const uint64_t np = touched / (64 * 80); CHK(hipEventRecord(e0)); hipLaunchKernelGGL((k_rows80<64, 5120, 80>), grid, block, 0, 0, buf, np); if (report("wr_64x80B_contiguous", np * 64 * 80)) return 1; }
    CHK(hipDeviceSynchronize());
    return 0;
}
