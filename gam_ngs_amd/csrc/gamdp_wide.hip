// libgamdp, gfx950: find_alignment for bands wider than the 64-lane systolic kernels cover (2 * band + 1 > 64 * 17, i.e. band > 543).
//
// The reference takes any band (lib/include/alignment/banded_smith_waterman.hpp:66: a constructor argument); gam-merge itself only
// ever passes 150 (:38).  This kernel is the correct-at-any-speed path for the rest: it has no throughput
// target and no tuned shape -- one workgroup per call, the whole band matrix H as int32 in the call's scratch slot (what the
// reference allocates, banded_smith_waterman.cc:102-107, at half the width), and
//   * the fill (banded_smith_waterman.cc:112-171) as a skewed sweep: cell (i, j) is due at step t = 2 i + j, when its three sources
//     (i-1, j) [t-2], (i-1, j+1) [t-1] and (i, j-1) [t-1] are there; the ~Y/2 cells of a step go over the workgroup's threads, a
//     barrier between steps;
//   * the end-cell search (:174-212) over the threads with a reduction that keeps the reference's scan order (first maximum wins);
//   * the traceback (:217-311) by one thread, reading H exactly as the reference does, with first_match_pos / last_match_pos
//     (my_alignment.cc:167-193, 228-262) taken on the way.
// Everything the tuned kernels derive (direction images, strips, packed cells) is absent here on purpose.
#include <hip/hip_runtime.h>

#include "gamdp_dev.h"

namespace gamdp {
namespace {

constexpr int WT = 256;          // threads per workgroup
constexpr int W_GAP = -8;        // GAP_SCORE, my_alignment.hpp:46
constexpr int W_MAXGAP = 10;     // FORCE_MAXGAP_LEN, banded_smith_waterman.hpp:37
enum { W_ST_OK = 0, W_ST_EMPTY = 1, W_ST_OUT_OF_RANGE = 2 };

typedef const __attribute__((address_space(1))) u32* wcptr;
typedef __attribute__((address_space(1))) int* wiptr;

__device__ __forceinline__ int w_code(const u32* p2, const u32* pn, int64_t idx)
{
    const wcptr g2 = (wcptr)p2, gn = (wcptr)pn;
    const int n = (gn[idx >> 5] >> (idx & 31)) & 1;
    const int c = (g2[idx >> 4] >> ((idx & 15) * 2)) & 3;
    return n ? 4 : c;
}
__device__ __forceinline__ int w_score(int p, int q) { return p == q ? 5 : ((p == 4 || q == 4) ? 0 : -4); }   // banded_smith_waterman.cc:80-88
__device__ __forceinline__ int w_max(int a, int b) { return a > b ? a : b; }

struct WTask {
    const u32 *a2, *an, *b2, *bn;
    int64_t a_base, b_base, end_a, la, ba, bb, X, Y, w;
    bool fs, fe;
    wiptr H;
    __device__ __forceinline__ int a_at(int64_t pos) const { return w_code(a2, an, a_base + pos); }
    __device__ __forceinline__ int b_at(int64_t i) const { return w_code(b2, bn, b_base + bb + i); }
    __device__ __forceinline__ int64_t pos_of(int64_t i, int64_t j) const { return ba + i + j - w; }
    // a cell of the reference's zero-initialised matrix: written by the fill iff 0 <= pos < |a|, zero otherwise
    __device__ __forceinline__ int h(int64_t i, int64_t j) const
    {
        const int64_t pos = pos_of(i, j);
        return (pos >= 0 && pos < la) ? H[i * Y + j] : 0;
    }
};

// one cell of the fill; the sources are in H (earlier steps)
__device__ __forceinline__ void w_cell(const WTask& t, const int64_t i, const int64_t j)
{
    const int64_t pos = t.pos_of(i, j);
    if (pos < 0 || pos >= t.la) return;
    const wiptr H = t.H;
    const int64_t Y = t.Y;
    if (i == 0) {   // row 0, :112-132 (left WITHOUT a gap penalty: the running maximum)
        const int d = w_score(t.a_at(pos), t.b_at(0));
        const bool chain = pos > 0 && j > 0;
        int v;
        if (!t.fs || pos <= W_MAXGAP) v = chain ? w_max(w_max(d, W_GAP), H[j - 1]) : w_max(W_GAP, d);
        else v = chain ? w_max(d, H[j - 1]) : d;
        H[j] = v;
        return;
    }
    const int d = w_score(t.a_at(pos), t.b_at(i));
    const bool has_up = j < Y - 1, has_left = j > 0;
    const int up = has_up ? H[(i - 1) * Y + j + 1] + W_GAP : W_GAP;
    int v;
    if (pos == 0) {   // :139-151
        if (!t.fs || i <= W_MAXGAP) v = has_up ? w_max(w_max(d, up), W_GAP) : w_max(d, W_GAP);
        else v = has_up ? w_max(d, up) : d;
    } else {          // :152-168
        const int dg = H[(i - 1) * Y + j] + d;
        const int left = has_left ? H[i * Y + j - 1] + W_GAP : W_GAP;
        if (has_up && has_left) v = w_max(w_max(dg, up), left);
        else if (has_up) v = w_max(dg, up);
        else if (has_left) v = w_max(dg, left);
        else v = dg;
    }
    H[i * Y + j] = v;
}

__global__ __launch_bounds__(WT) void k_align_w(const LaunchParams p)
{
    __shared__ u32 s_task;
    __shared__ int s_val[WT];
    __shared__ long long s_idx[WT];
    const int tid = (int)threadIdx.x;
    u32* const slot = p.scratch + (u64)blockIdx.x * p.slot_words;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_task = atomicAdd(p.cursor, 1u);
        __syncthreads();
        const u32 ti = s_task;
        if (ti >= p.n_tasks) return;
        const DevTask& d = p.tasks[ti];
        WTask t;
        t.a2 = d.a2; t.an = d.an; t.b2 = d.b2; t.bn = d.bn;
        t.a_base = d.a_base; t.b_base = d.b_base; t.end_a = d.end_a;
        t.la = d.alen; t.ba = d.begin_a; t.bb = d.begin_b; t.X = d.X; t.w = d.band; t.Y = 2 * (int64_t)d.band + 1;
        t.fs = (d.flags & TF_FORCE_START) != 0; t.fe = (d.flags & TF_FORCE_END) != 0;
        t.H = (wiptr)slot;
        const int64_t X = t.X, Y = t.Y;

        // ---- fill: step s takes the cells (i, s - 2 i)
        const int64_t steps = 2 * (X - 1) + Y;
        for (int64_t s = 0; s < steps; ++s) {
            const int64_t i_lo = s - (Y - 1) > 0 ? (s - (Y - 1) + 1) / 2 : 0;
            const int64_t i_hi = s / 2 < X - 1 ? s / 2 : X - 1;
            for (int64_t i = i_lo + tid; i <= i_hi; i += WT) w_cell(t, i, s - 2 * i);
            __syncthreads();   // (a workgroup's wavefronts share one vector L1: its stores are visible to all of them behind the barrier)
        }

        // ---- end cell, :174-212: scan order = the last row's columns left to right (unless force_end), then the pos == end_a
        // anti-diagonal downwards; strict '>' -- the first maximum wins.  Candidate k of that order: k < Y the last row, then the rest.
        const bool ge = (u64)t.end_a >= (u64)(t.ba + t.w);                       // [types] unsigned compare
        const int64_t di0 = ge ? t.end_a - (t.ba + t.w) : 0;
        const int64_t dj0 = ge ? 2 * t.w : 2 * t.w - (t.ba + t.w - t.end_a);
        int64_t dn = 0;                                                        // cells (di0 + k, dj0 - k), k = 0 .. dn - 1
        if (dj0 >= 0 && di0 < X) dn = (X - di0 < dj0 + 1) ? X - di0 : dj0 + 1;
        int best = 0;
        long long best_k = -1;
        for (int64_t k = tid; k < Y + dn; k += WT) {
            bool eligible;
            int64_t i, j;
            if (k < Y) {
                i = X - 1; j = k;
                const int64_t pos = t.pos_of(i, j);
                eligible = !t.fe && pos >= 0 && (u64)pos <= (u64)t.end_a;
            } else {
                i = di0 + (k - Y); j = dj0 - (k - Y);
                eligible = !t.fe || ((u64)i >= (u64)(X - 1 - W_MAXGAP) && i < X);   // [types] x_size - 1 - FORCE_MAXGAP_LEN wraps when x_size <= 10
            }
            if (!eligible) continue;
            const int v = t.h(i, j);
            if (best_k < 0 || v > best) { best = v; best_k = k; }
        }
        s_val[tid] = best; s_idx[tid] = best_k;
        __syncthreads();
        if (tid == 0) {
            int bv = 0;
            long long bk = -1;
            for (int q = 0; q < WT; ++q) {
                const long long k = s_idx[q];
                if (k < 0) continue;
                if (bk < 0 || s_val[q] > bv || (s_val[q] == bv && k < bk)) { bv = s_val[q]; bk = k; }
            }
            DevResult res;
            res.begin_a = res.begin_b = res.score = 0;
            res.n_match = res.length = 0;
            res.first_a = res.first_b = res.last_a = res.last_b = 0;
            res.flags = (u32)W_ST_EMPTY << 8;   // :215
            if (bk >= 0) {
                int64_t x = bk < Y ? X - 1 : di0 + (bk - Y), y = bk < Y ? bk : dj0 - (bk - Y);
                int64_t pos = t.pos_of(x, y);
                const bool want_ops = (d.flags & TF_WANT_OPS) != 0;
                uint8_t* const ops = p.ops_buf + d.ops_off;
                const u64 ops_cap = d.ops_cap;
                u32 len = 0, nm = 0;
                bool have_first = false, have_last = false, thrown = false;
                int64_t fa = 0, fb = 0, la_ = 0, lb_ = 0;
                u32 consumed_a = 0, consumed_b = 0;
                // ---- traceback, :217-311 (ops in traceback order: the host reverses them)
                while (x >= 0 && y >= 0 && pos >= 0) {
                    if (pos >= t.la) { thrown = true; break; }   // a.at(pos) throws
                    const int pa = t.a_at(pos), pb = t.b_at(x);
                    const int sc = w_score(pa, pb);
                    const int h = t.h(x, y);
                    const bool is_match = pa == pb || pa == 4 || pb == 4;
                    int op;   // 0 GAP_A (consumes b), 1 GAP_B (consumes a), 2 diagonal
                    if (pos == 0) {
                        const bool left_ok = !(t.fs && x > W_MAXGAP);
                        if (h == sc) op = 2;
                        else if (y == Y - 1 || (left_ok && h == W_GAP)) op = 1;
                        else op = 0;
                    } else {
                        const int dg = (x > 0 ? t.h(x - 1, y) : 0) + sc;
                        bool up_ok = true;
                        int up = (x > 0 && y < Y - 1) ? t.h(x - 1, y + 1) + W_GAP : W_GAP;
                        if (t.fs && x == 0) {
                            if (pos <= W_MAXGAP) up = W_GAP;
                            else up_ok = false;
                        }
                        if (h == dg) op = 2;
                        else if (y < Y - 1 && y > 0 && up_ok && h == up) op = 0;
                        else if (y < Y - 1 && y > 0) op = 1;
                        else if (y < Y - 1) op = 0;
                        else op = 1;
                    }
                    if (op == 2) {
                        if (is_match) {
                            nm++;
                            if (!have_last) { have_last = true; la_ = pos; lb_ = t.bb + x; }   // the first MATCH the walk meets is the alignment's last
                            have_first = true; fa = pos; fb = t.bb + x;
                        }
                        if (want_ops && len < ops_cap) ops[len] = is_match ? 2 : 3;
                        x--; consumed_a++; consumed_b++;
                    } else if (op == 1) {
                        if (want_ops && len < ops_cap) ops[len] = 1;
                        y--; consumed_a++;
                    } else {
                        if (want_ops && len < ops_cap) ops[len] = 0;
                        x--; y++; consumed_b++;
                    }
                    len++;
                    pos = t.pos_of(x, y);
                }
                if (thrown) res.flags = (u32)W_ST_OUT_OF_RANGE << 8;
                else {
                    res.begin_a = (int32_t)(pos + 1);
                    res.begin_b = (int32_t)(t.bb + x + 1);
                    res.score = bv;
                    res.n_match = nm; res.length = len;
                    // first_match_pos without a MATCH: the coordinates behind the last op; last_match_pos without one: the begin coordinates
                    res.first_a = have_first ? (int32_t)fa : (int32_t)(res.begin_a + (int32_t)consumed_a);
                    res.first_b = have_first ? (int32_t)fb : (int32_t)(res.begin_b + (int32_t)consumed_b);
                    res.last_a = have_last ? (int32_t)la_ : res.begin_a;
                    res.last_b = have_last ? (int32_t)lb_ : res.begin_b;
                    res.flags = (have_first ? 1u : 0u) | (have_last ? 2u : 0u) | ((u32)W_ST_OK << 8);
                }
            }
            p.results[d.res_idx] = res;
            if (p.stats != nullptr) atomicAdd(p.stats + LS_UNITS, 1u);
        }
    }
}

}  // namespace

int launch_wide(const LaunchParams& p, unsigned n_slots, void* stream)
{
    LaunchParams lp = p;
    void* args[] = {&lp};
    return (int)hipLaunchKernel((const void*)k_align_w, dim3(n_slots), dim3(WT), args, 0, static_cast<hipStream_t>(stream));
}
unsigned wide_static_lds()
{
    hipFuncAttributes a;
    return hipFuncGetAttributes(&a, (const void*)k_align_w) == hipSuccess ? (unsigned)a.sharedSizeBytes : 0u;
}

}  // namespace gamdp
