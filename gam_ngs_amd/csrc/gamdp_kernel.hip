// gfx950 (MI355X / CDNA4) kernels for the banded semi-global alignment of gam-merge
// (reference: lib/src/alignment/banded_smith_waterman.cc:69-322, BandedSmithWaterman::find_alignment).
//
// One wavefront (64 lanes) owns one alignment task; a launch is a persistent grid of waves that pull
// tasks from an atomic cursor.  Band matrix H[i][j], i = row (b index), j = band column,
// pos = begin_a + i + j - band (a index).  Dependencies of (i,j): (i-1,j) diag, (i-1,j+1) up, (i,j-1) left.
//
//   Row-systolic sweep.  Lane l owns band columns [C*l, C*l+C) and processes row i at "row-time"
//   tau = i + l.  Inside a row-time every lane first computes its column 0, hands it to lane l-1 (the
//   `up` source of that lane's last column, one row behind) with one DPP wave shift, sweeps columns
//   1..C-1, and hands its last column to lane l+1 (the `left` source of that lane's column 0 at the next
//   row-time) with a second DPP shift.  All 64 lanes are busy on every row-time except 63 ramp row-times.
//
//   Tilted, tagged scores.  We keep G4 = 4*(H + 16*i + 8*j) + tag.  In this tilt the gap penalty of
//   `up` and `left` is 0 and `diag` adds 4*(S+16)+2 in {86 match, 50 mismatch, 66 N-vs-base}; the two low
//   bits carry the winner (2 diag, 1 up, 0 left) so ONE v_max3_i32 picks the value and, on ties, the
//   traceback preference diag > up > left of the reference (:273-304).  A cell is
//       v_dot4_u32_u8 (one-hot(a) . scorerow(b) + H_diag)   | v_perm_b32 + v_add for the N-aware kernels
//       v_or_b32      (tag the `up` source)
//       v_max3_i32
//       v_alignbit_b32 (append the 2-bit direction to the lane's 16-row direction word)
//       v_and_b32     (strip the tag for the left chain / next row)
//   Direction words (2 bit/cell) are the only per-cell HBM traffic: 16 B/lane coalesced stores.
//
//   Everything the reference treats specially is kept exact: row 0 (gap-free running max, :112-132),
//   the pos==0 column (:141-155), force_start/force_end windows, the end-cell scan order (:174-212),
//   zero-valued cells outside a, and the traceback rules for row 0 / pos 0 (:227-258), for which the
//   needed H values are spilled to small side buffers.
#include <hip/hip_runtime.h>

#include "gamdp_dev.h"

namespace gamdp {
namespace {

constexpr int NEG = -(1 << 30);
constexpr int ROWS = 16;  // row-times per direction word
constexpr int GAP = -8;   // GAP_SCORE, my_alignment.hpp:46
constexpr int FORCE_MAXGAP = 10;

enum { ST_OK = 0, ST_EMPTY = 1, ST_OUT_OF_RANGE = 2 };

// ---- packed sequence access ---------------------------------------------------------------------
__device__ __forceinline__ u32 fetch16(const u32* __restrict__ p2, int64_t idx)
{  // 16 bases starting at base idx (any alignment, idx may be negative: pads)
    const int64_t w = idx >> 4;
    const u32 sh = (u32)(idx & 15) * 2u;
    return __builtin_amdgcn_alignbit(p2[w + 1], p2[w], sh);
}
__device__ __forceinline__ u32 fetch16n(const u32* __restrict__ pn, int64_t idx)
{
    const int64_t w = idx >> 5;
    const u32 sh = (u32)(idx & 31);
    return __builtin_amdgcn_alignbit(pn[w + 1], pn[w], sh) & 0xFFFFu;
}
__device__ __forceinline__ int code_at(const u32* __restrict__ p2, const u32* __restrict__ pn, int64_t idx)
{
    const int n = (pn[idx >> 5] >> (idx & 31)) & 1;
    const int c = (p2[idx >> 4] >> ((idx & 15) * 2)) & 3;
    return n ? 4 : c;
}
__device__ __forceinline__ int score_of(int p, int q) { return p == q ? 5 : ((p == 4 || q == 4) ? 0 : -4); }

__device__ __forceinline__ int wave_shl1(int v)  // lane l <- lane l+1 ; lane 63 <- NEG
{
    return __builtin_amdgcn_update_dpp(NEG, v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_shr1(int v)  // lane l <- lane l-1 ; lane 0 <- NEG
{
    return __builtin_amdgcn_update_dpp(NEG, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }

// uniform per-task values
struct Tk {
    const u32 *a2, *an, *b2, *bn;
    int64_t a_base, b_base, end_a;
    int alen, blen, begin_a, begin_b, X, band, Y;
    bool fs, fe;
    int iA;       // first row of the pos==end_a anti-diagonal scan (:192)
    int eaRel;    // end_a - begin_a + band clamped to int: band column of pos==end_a in row 0
    u32* dir;
    int *h0row, *pos0, *lastrow, *adh;
};

template <int C>
__device__ __forceinline__ u64 dir_index(int blk, int lane, int c)
{
    constexpr int G = C / 4, REM = C % 4;
    const u64 base = (u64)blk * (u64)(C * 64);
    if (c < 4 * G) return base + (u64)((c >> 2) * 256 + lane * 4 + (c & 3));
    return base + (u64)(G * 256 + lane * REM + (c - 4 * G));
}

// ---- one block of 16 row-times --------------------------------------------------------------------
template <int C, int CE, bool HASN, bool SLOW>
__device__ __forceinline__ void do_block(int (&Lp)[C], u32 (&acc)[C], u32 (&W)[C + 15], int& Lin, const Tk& t,
                                         const int blk, const int lane, const int LE, const int kill_c)
{
    const int tau0 = blk * ROWS;
    const int64_t sA = t.a_base + t.begin_a - t.band + (int64_t)(C - 1) * (lane + 1) + tau0;
    const int64_t sB = t.b_base + t.begin_b + tau0 - lane;
    const u32 abits = fetch16(t.a2, sA), bbits = fetch16(t.b2, sB);
    u32 anb = 0, bnb = 0;
    if (HASN) {
        anb = fetch16n(t.an, sA);
        bnb = fetch16n(t.bn, sB);
    }

#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        u32 ca = (abits >> (2 * r)) & 3u;
        u32 cb = (bbits >> (2 * r)) & 3u;
        u32 brow, bhi = 0;
        if (HASN) {
            if ((anb >> r) & 1u) ca = 4;
            W[C - 1 + r] = 0x0C0C0C00u | ca;  // v_perm selector: byte0 = table[ca]
            const bool bN = (bnb >> r) & 1u;
            brow = bN ? 0x42424242u : (0x32323232u + (0x24u << (cb * 8u)));
            bhi = bN ? 0x56u : 0x42u;
        } else {
            W[C - 1 + r] = 1u << (ca * 8u);  // one-hot byte per base
            brow = 0x32323232u + (0x24u << (cb * 8u));
        }

        // slow-path per-row values
        const int row = tau0 + r - lane;
        bool act = true;
        int cm1 = 0, cE = 0, Zst = 0, ZL = 0, cap0 = 0, capE = 0;
        if (SLOW) {
            act = row >= 1;
            cm1 = (t.band - t.begin_a - row - 1) - C * lane;  // column whose pos == -1
            cE = (t.eaRel - row) - C * lane;                  // column whose pos == end_a
            Zst = 32 * row + 32 * (t.band - t.begin_a - 1);   // G4 of H = 0 at the pos == -1 cell
            ZL = (t.fs && row > FORCE_MAXGAP) ? NEG : Zst;    // ... as a `left` source (:150-155)
        }

        int L = Lin;
        int x = NEG;
        auto cell = [&](const int c) __attribute__((always_inline)) {
            int D;
            if (HASN) D = Lp[c] + (int)__builtin_amdgcn_perm(bhi, brow, W[r + c]);
            else D = (int)__builtin_amdgcn_udot4(W[r + c], brow, (u32)Lp[c], false);
            int Uc = (c < C - 1) ? (Lp[(c < C - 1) ? c + 1 : c] | 1) : (x | 1);
            if (CE >= 0) {
                if (c == CE && CE < C - 1) Uc = (lane == LE) ? NEG : Uc;
            } else {
                Uc = (c == kill_c) ? NEG : Uc;
            }
            const int R = imax3(D, Uc, L);
            acc[c] = __builtin_amdgcn_alignbit((u32)R, acc[c], 2);
            const int Lc = R & ~3;
            if (SLOW) {
                const bool m1 = (cm1 == c);
                cap0 = (cm1 == c - 1) ? R : cap0;
                capE = (cE == c) ? R : capE;
                Lp[c] = m1 ? Zst : Lc;
                L = m1 ? ZL : Lc;
            } else {
                Lp[c] = Lc;
                L = Lc;
            }
        };

        if (!SLOW || act) cell(0);
        x = wave_shl1(Lp[0]);
        x = (lane >= LE) ? NEG : x;
        if (!SLOW || act) {
#pragma unroll
            for (int c = 1; c < C; ++c) cell(c);
        }
        Lin = wave_shr1(L);

        if (SLOW) {
            if (act && row <= t.X - 1) {
                const int c0 = cm1 + 1;
                if (c0 >= 0 && c0 < C) {
                    const int j0 = C * lane + c0;
                    if (j0 < t.Y) t.pos0[row] = (cap0 >> 2) - 16 * row - 8 * j0;
                }
                if (cE >= 0 && cE < C) {
                    const int jE = C * lane + cE;
                    if (jE < t.Y) t.adh[row - t.iA] = (capE >> 2) - 16 * row - 8 * jE;
                }
                if (row == t.X - 1) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const int j = C * lane + c;
                        if (j < t.Y) t.lastrow[j] = (Lp[c] >> 2) - 16 * row - 8 * j;
                    }
                }
            }
        }
    }

    // direction words of this block: 16 B / lane coalesced
    {
        constexpr int G = C / 4, REM = C % 4;
        u32* blkp = t.dir + (u64)blk * (u64)(C * 64);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            uint4 v = make_uint4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
            *reinterpret_cast<uint4*>(blkp + g * 256 + lane * 4) = v;
        }
#pragma unroll
        for (int e = 0; e < REM; ++e) blkp[G * 256 + lane * REM + e] = acc[4 * G + e];
    }
    // slide the a window
#pragma unroll
    for (int k = 0; k < C - 1; ++k) W[k] = W[k + ROWS];
}

// ---- the whole task -------------------------------------------------------------------------------
template <int C, int CE, bool HASN>
__device__ __forceinline__ void run_task(const DevTask& dt, const LaunchParams& p, u32* slot, const int lane)
{
    Tk t;
    t.a2 = dt.a2; t.an = dt.an; t.b2 = dt.b2; t.bn = dt.bn;
    t.a_base = dt.a_base; t.b_base = dt.b_base; t.end_a = dt.end_a;
    t.alen = dt.alen; t.blen = dt.blen; t.begin_a = dt.begin_a; t.begin_b = dt.begin_b;
    t.X = dt.X; t.band = dt.band; t.Y = 2 * dt.band + 1;
    t.fs = dt.flags & TF_FORCE_START; t.fe = dt.flags & TF_FORCE_END;
    t.dir = slot;
    t.h0row = reinterpret_cast<int*>(slot + p.dir_words);
    t.pos0 = t.h0row + p.ypad;
    t.lastrow = t.pos0 + p.ypad;
    t.adh = t.lastrow + p.ypad;
    {
        const int64_t rel = t.end_a - t.begin_a + t.band;  // may be negative
        t.eaRel = (int)min(max(rel, (int64_t)-(1 << 30)), (int64_t)(1 << 30));
        const bool ge = t.end_a >= (int64_t)t.begin_a + t.band;
        const int64_t ia = ge ? t.end_a - ((int64_t)t.begin_a + t.band) : 0;
        t.iA = (int)min(ia, (int64_t)(1 << 30));
    }
    const int X = t.X, Y = t.Y, w = t.band;
    const int LE = (Y - 1) / C;                              // lane holding the last band column
    const int ce_rt = (Y - 1) % C;
    const int kill_c = (CE < 0 && lane == LE) ? ce_rt : -1;  // generic kernels: runtime edge column

    // ---- phase A: row 0 (:112-132): running max without gap penalty along j --------------------------
    int Lp[C];
    u32 acc[C];
    u32 W[C + 15];
    int Lin = NEG;
    {
        const int cb0 = code_at(t.b2, t.bn, t.b_base + t.begin_b);
        int e[C];
        bool q[C];
        int run = NEG;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = C * lane + c;
            const int pos = t.begin_a - w + j;
            q[c] = (j < Y) && pos >= 0 && pos < t.alen;  // host guarantees force_start never needs pos >= alen
            int v = NEG;
            if (q[c]) {
                const int d = score_of(code_at(t.a2, t.an, t.a_base + pos), cb0);
                v = (t.fs && pos > FORCE_MAXGAP) ? d : max(d, GAP);
            }
            run = max(run, v);
            e[c] = run;
        }
        // exclusive max-scan of lane totals
        int incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o, 64);
            if (lane >= o) incl = max(incl, up);
        }
        int pre = __shfl_up(incl, 1, 64);
        if (lane == 0) pre = NEG;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = C * lane + c;
            const int h = q[c] ? max(pre, e[c]) : 0;
            if (j < Y) t.h0row[j] = h;
            Lp[c] = (j < Y) ? 4 * (h + 8 * j) : NEG;
            acc[c] = 0;
        }
        if (lane > LE) {
#pragma unroll
            for (int c = 0; c < C; ++c) Lp[c] = NEG;
        }
        // a window for row-time 0: W[k] <-> a index begin_a - band + (C-1)*lane + k , k < C-1
        const int64_t s0 = t.a_base + t.begin_a - w + (int64_t)(C - 1) * lane;
        const u32 ab = fetch16(t.a2, s0);
        const u32 an = HASN ? fetch16n(t.an, s0) : 0u;
#pragma unroll
        for (int k = 0; k < C - 1; ++k) {
            u32 ca = (ab >> (2 * k)) & 3u;
            if (HASN) {
                if ((an >> k) & 1u) ca = 4;
                W[k] = 0x0C0C0C00u | ca;
            } else {
                W[k] = 1u << (ca * 8u);
            }
        }
#pragma unroll
        for (int k = C - 1; k < C + 15; ++k) W[k] = 0;
    }

    // ---- phase B: rows 1..X-1 -------------------------------------------------------------------------
    const int nblk = (X - 1 + LE) / ROWS + 1;
    {
        const int64_t iE0 = t.end_a - t.begin_a - w, iE1 = t.end_a - t.begin_a + w;
        for (int blk = 0; blk < nblk; ++blk) {
            const int tau0 = blk * ROWS;
            const bool fast = (tau0 - LE >= 1) && (t.begin_a - w + tau0 >= 1) && (tau0 + ROWS - 1 < X - 1) &&
                              ((int64_t)(tau0 + ROWS - 1) < iE0 || (int64_t)(tau0 - LE) > iE1);
            if (fast) do_block<C, CE, HASN, false>(Lp, acc, W, Lin, t, blk, lane, LE, kill_c);
            else do_block<C, CE, HASN, true>(Lp, acc, W, Lin, t, blk, lane, LE, kill_c);
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop L1 lines cached by the slot's previous task

    // ---- phase C: end cell (:174-212), first maximum in scan order wins --------------------------------
    int best = NEG, bkey = 0x7fffffff;  // key = scan position
    {
        const int* lr = (X == 1) ? t.h0row : t.lastrow;
        if (!t.fe) {
            for (int j = lane; j < Y; j += 64) {
                const int64_t pos = (int64_t)t.begin_a + (X - 1) + j - w;
                if (pos >= 0 && pos <= t.end_a) {
                    const int v = (pos < t.alen) ? lr[j] : 0;  // cells outside a keep their zero
                    if (v > best || (v == best && j < bkey)) { best = v; bkey = j; }
                }
            }
        }
        // anti-diagonal pos == end_a: cells (iA + k, jA - k)
        const bool ge = t.end_a >= (int64_t)t.begin_a + w;
        const int64_t jA64 = ge ? (int64_t)2 * w : (int64_t)2 * w - ((int64_t)t.begin_a + w - t.end_a);
        if (jA64 >= 0 && t.iA < X) {
            const int jA = (int)jA64;
            const int cnt = min(X - t.iA, jA + 1);
            for (int k = lane; k < cnt; k += 64) {
                const int i = t.iA + k, j = jA - k;
                bool ok = true;
                if (t.fe) ok = (X >= FORCE_MAXGAP + 1) && (i >= X - 1 - FORCE_MAXGAP);  // unsigned compare in the reference
                if (ok) {
                    int v = 0;
                    if (t.end_a < t.alen) v = (i == 0) ? t.h0row[j] : ((i == X - 1) ? lr[j] : t.adh[k]);
                    const int key = Y + k;
                    if (v > best || (v == best && key < bkey)) { best = v; bkey = key; }
                }
            }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const int ov = __shfl_xor(best, o, 64), ok = __shfl_xor(bkey, o, 64);
            if (ov > best || (ov == best && ok < bkey)) { best = ov; bkey = ok; }
        }
    }

    DevResult res;
    res.begin_a = res.begin_b = res.score = 0;
    res.n_match = res.length = 0;
    res.first_a = res.first_b = res.last_a = res.last_b = 0;
    res.flags = ST_EMPTY << 8;
    if (bkey != 0x7fffffff) {
        int x, y;
        if (bkey < Y) { x = X - 1; y = bkey; }
        else {
            const bool ge = t.end_a >= (int64_t)t.begin_a + w;
            const int jA = ge ? 2 * w : (int)((int64_t)2 * w - ((int64_t)t.begin_a + w - t.end_a));
            x = t.iA + (bkey - Y);
            y = jA - (bkey - Y);
        }
        int64_t pos64 = (int64_t)t.begin_a + x + y - w;
        if (pos64 >= t.alen) {
            res.flags = ST_OUT_OF_RANGE << 8;  // reference: a.at(pos) throws in the traceback
        } else {
            // ---- phase D: traceback (:217-311) ------------------------------------------------------------
            int pos = (int)pos64;
            const int end_pos = pos, end_x = x;
            const bool want_ops = dt.flags & TF_WANT_OPS;
            uint8_t* ops = p.ops_buf + dt.ops_off;
            u32 len = 0, nm = 0;
            bool have_last = false, have_first = false;
            int la = 0, lb = 0, fa = 0, fb = 0;
            int l = y / C, c = y - l * C;
            // cached direction word
            int cw_blk = -1, cw_l = -1, cw_c = -1;
            u32 cw = 0;
            while (x >= 0 && y >= 0 && pos >= 0) {
                if (x == 0 || pos == 0 || want_ops) {
                    // single step with the reference's exact rules
                    const int pa = HASN ? code_at(t.a2, t.an, t.a_base + pos) : (int)((t.a2[(t.a_base + pos) >> 4] >> (((t.a_base + pos) & 15) * 2)) & 3);
                    const int64_t bi = t.b_base + t.begin_b + x;
                    const int pb = HASN ? code_at(t.b2, t.bn, bi) : (int)((t.b2[bi >> 4] >> ((bi & 15) * 2)) & 3);
                    const bool is_match = (pa == pb) || pa == 4 || pb == 4;
                    int op;  // 0 GAP_A, 1 GAP_B, 2 diag
                    if (pos == 0) {
                        const int s = score_of(pa, pb);
                        const int h = (x == 0) ? t.h0row[y] : t.pos0[x];
                        const bool left_ok = !(t.fs && x > FORCE_MAXGAP);
                        if (h == s) op = 2;
                        else if (y == Y - 1 || (left_ok && h == GAP)) op = 1;
                        else op = 0;
                    } else if (x == 0) {
                        const int s = score_of(pa, pb);
                        const int h = t.h0row[y];
                        const bool up_ok = !(t.fs && pos > FORCE_MAXGAP);
                        if (h == s) op = 2;
                        else if (y < Y - 1 && y > 0 && up_ok && h == GAP) op = 0;
                        else if (y < Y - 1 && y > 0) op = 1;
                        else if (y < Y - 1) op = 0;
                        else op = 1;
                    } else {
                        const int tau = x + l, blk = tau >> 4;
                        if (blk != cw_blk || l != cw_l || c != cw_c) {
                            cw = t.dir[dir_index<C>(blk, l, c)];
                            cw_blk = blk; cw_l = l; cw_c = c;
                        }
                        const u32 tag = (cw >> ((tau & 15) * 2)) & 3u;
                        op = (tag == 2u) ? 2 : (tag == 1u ? 0 : 1);
                    }
                    if (op == 2) {
                        if (is_match) {
                            nm++;
                            if (!have_last) { have_last = true; la = pos; lb = t.begin_b + x; }
                            have_first = true; fa = pos; fb = t.begin_b + x;
                        }
                        if (want_ops && lane == 0 && len < dt.ops_cap) ops[len] = is_match ? 2 : 3;
                        x--; pos--;
                    } else if (op == 1) {  // GAP_B: consumes a
                        if (want_ops && lane == 0 && len < dt.ops_cap) ops[len] = 1;
                        y--; pos--;
                        if (--c < 0) { c = C - 1; l--; }
                    } else {  // GAP_A: consumes b
                        if (want_ops && lane == 0 && len < dt.ops_cap) ops[len] = 0;
                        x--; y++;
                        if (++c == C) { c = 0; l++; }
                    }
                    len++;
                } else {
                    // interior: consume a whole run of diagonal steps from one direction word
                    const int tau = x + l, blk = tau >> 4, r = tau & 15;
                    if (blk != cw_blk || l != cw_l || c != cw_c) {
                        cw = t.dir[dir_index<C>(blk, l, c)];
                        cw_blk = blk; cw_l = l; cw_c = c;
                    }
                    const u32 T = (cw ^ 0xAAAAAAAAu) << (30 - 2 * r);  // pair r on top; diag pairs are 00
                    int n = T ? (__builtin_clz(T) >> 1) : (r + 1);
                    n = min(n, min(x, pos));  // stay in x >= 1, pos >= 1
                    if (n > 0) {
                        const int64_t ia = t.a_base + pos - n + 1, ib = t.b_base + t.begin_b + x - n + 1;
                        const u32 xr = fetch16(t.a2, ia) ^ fetch16(t.b2, ib);
                        u32 ne = (xr | (xr >> 1)) & 0x55555555u;  // bit 2k set: bases k differ
                        if (HASN) {
                            u32 nn = fetch16n(t.an, ia) | fetch16n(t.bn, ib);  // either is N -> MATCH
                            // spread 16 bits to even positions
                            nn = (nn | (nn << 8)) & 0x00FF00FFu;
                            nn = (nn | (nn << 4)) & 0x0F0F0F0Fu;
                            nn = (nn | (nn << 2)) & 0x33333333u;
                            nn = (nn | (nn << 1)) & 0x55555555u;
                            ne &= ~nn;
                        }
                        const u32 msk = (n == 16) ? 0x55555555u : (((1u << (2 * n)) - 1u) & 0x55555555u);
                        const u32 eq = ~ne & msk;
                        if (eq) {
                            nm += (u32)__builtin_popcount(eq);
                            const int hi = (31 - __builtin_clz(eq)) >> 1, lo = __builtin_ctz(eq) >> 1;
                            if (!have_last) { have_last = true; la = pos - n + 1 + hi; lb = t.begin_b + x - n + 1 + hi; }
                            have_first = true; fa = pos - n + 1 + lo; fb = t.begin_b + x - n + 1 + lo;
                        }
                        x -= n; pos -= n; len += (u32)n;
                    } else {
                        const u32 tag = (cw >> (r * 2)) & 3u;
                        if (tag == 1u) {  // GAP_A
                            x--; y++;
                            if (++c == C) { c = 0; l++; }
                        } else {  // GAP_B
                            y--; pos--;
                            if (--c < 0) { c = C - 1; l--; }
                        }
                        len++;
                    }
                }
            }
            res.begin_a = pos + 1;
            res.begin_b = t.begin_b + x + 1;
            res.score = best;
            res.n_match = nm;
            res.length = len;
            // first_match_pos without a MATCH returns the end coordinates, last_match_pos the begin ones
            res.first_a = have_first ? fa : end_pos + 1;
            res.first_b = have_first ? fb : t.begin_b + end_x + 1;
            res.last_a = have_last ? la : res.begin_a;
            res.last_b = have_last ? lb : res.begin_b;
            res.flags = (have_first ? 1u : 0u) | (have_last ? 2u : 0u) | (ST_OK << 8);
        }
    }
    if (lane == 0) p.results[dt.res_idx] = res;
    __syncthreads();
}

template <int C, int CE, bool HASN>
__global__ __launch_bounds__(64, 4) void k_align(const LaunchParams p)
{
    const int lane = threadIdx.x;
    u32* slot = p.scratch + (u64)blockIdx.x * p.slot_words;
    for (;;) {
        u32 ti = 0;
        if (lane == 0) ti = atomicAdd(p.cursor, 1u);
        ti = __builtin_amdgcn_readfirstlane(ti);
        if (ti >= p.n_tasks) break;
        run_task<C, CE, HASN>(p.tasks[ti], p, slot, lane);
    }
}

}  // namespace

int kernel_cols(int kid)
{
    switch (kid) {
    case K_C17_CE4: case K_C17_CE4_N: case K_GEN_C17: return 17;
    case K_C5_CE0: case K_C5_CE0_N: case K_GEN_C5: return 5;
    case K_GEN_C2: return 2;
    case K_GEN_C3: return 3;
    case K_GEN_C9: return 9;
    default: return 0;
    }
}

int kernel_waves_per_cu(int) { return 16; }

int launch_align(int kid, const LaunchParams& p, unsigned n_slots, void* stream)
{
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 g(n_slots), b(64);
    switch (kid) {
    case K_C17_CE4:   hipLaunchKernelGGL((k_align<17, 4, false>), g, b, 0, s, p); break;
    case K_C17_CE4_N: hipLaunchKernelGGL((k_align<17, 4, true>), g, b, 0, s, p); break;
    case K_C5_CE0:    hipLaunchKernelGGL((k_align<5, 0, false>), g, b, 0, s, p); break;
    case K_C5_CE0_N:  hipLaunchKernelGGL((k_align<5, 0, true>), g, b, 0, s, p); break;
    case K_GEN_C2:    hipLaunchKernelGGL((k_align<2, -1, true>), g, b, 0, s, p); break;
    case K_GEN_C3:    hipLaunchKernelGGL((k_align<3, -1, true>), g, b, 0, s, p); break;
    case K_GEN_C5:    hipLaunchKernelGGL((k_align<5, -1, true>), g, b, 0, s, p); break;
    case K_GEN_C9:    hipLaunchKernelGGL((k_align<9, -1, true>), g, b, 0, s, p); break;
    case K_GEN_C17:   hipLaunchKernelGGL((k_align<17, -1, true>), g, b, 0, s, p); break;
    default: return (int)hipErrorInvalidValue;
    }
    return (int)hipGetLastError();
}

}  // namespace gamdp
