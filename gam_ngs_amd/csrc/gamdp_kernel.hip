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
//       v_dot4_u32_u8 (one-hot(a) . scorerow(b) + H_diag)   | v_dot8_u32_u4 over 5 letters in the N-aware kernels
//       v_or_b32      (tag the `up` source)
//       v_max3_i32
//       v_alignbit_b32 (append the 2-bit direction to the lane's 16-row direction word)
//       v_and_b32     (strip the tag for the left chain / next row)
//   Direction words (2 bit/cell) are the only per-cell HBM traffic: 16 B/lane coalesced stores.  The two
//   per-row sequence operands (one-hot of the incoming a base, score row of the b base) are expanded once per
//   block by 16 lanes into per-wave LDS rings and read back with two ds_reads per row (LDS is otherwise idle).
//
//   Direction-free fast blocks (band-512 kernels).  The walk reads the directions of the ~50 000 cells on the path
//   only, so the fast blocks of those kernels compute plain values -- v_dot4 + v_max3 per cell (do_block_df) -- and
//   store the live row of every 4th block plus the values that cross every 4th lane boundary; materialise()
//   re-enacts a 4-lane strip of the sweep around the path, with the tagged cell, whenever the walk needs directions
//   that are not there.  Same recurrences on the same inputs: bit-identical, ~45 % fewer vector instructions.
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

// All buffers live in HBM: tell the compiler (pointers read from DevTask/LaunchParams would otherwise be
// "flat" and every access would tie up both the vector-memory and the LDS counters).
typedef const __attribute__((address_space(1))) u32* gcptr;
typedef __attribute__((address_space(1))) u32* gptr;
typedef __attribute__((address_space(1))) int* giptr;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u32x4* g4ptr;
__device__ __forceinline__ gcptr as_global(const u32* p) { return (gcptr)p; }

// ---- packed sequence access ---------------------------------------------------------------------
__device__ __forceinline__ u32 fetch16(gcptr p2, int64_t idx)
{  // 16 bases starting at base idx (any alignment, idx may be negative: pads)
    const int64_t w = idx >> 4;
    const u32 sh = (u32)(idx & 15) * 2u;
    return __builtin_amdgcn_alignbit(p2[w + 1], p2[w], sh);
}
__device__ __forceinline__ u32 fetch16n(gcptr pn, int64_t idx)
{
    const int64_t w = idx >> 5;
    const u32 sh = (u32)(idx & 31);
    return __builtin_amdgcn_alignbit(pn[w + 1], pn[w], sh) & 0xFFFFu;
}
__device__ __forceinline__ int code_at(gcptr p2, gcptr pn, int64_t idx)
{
    const int n = (pn[idx >> 5] >> (idx & 31)) & 1;
    const int c = (p2[idx >> 4] >> ((idx & 15) * 2)) & 3;
    return n ? 4 : c;
}
__device__ __forceinline__ int score_of(int p, int q) { return p == q ? 5 : ((p == 4 || q == 4) ? 0 : -4); }

// DPP wave shifts.  The lane without a source keeps `keep` (pass the previous result: it was NEG at the start of
// the task and so stays NEG, without a v_mov to re-materialise the constant before every shift).
__device__ __forceinline__ int wave_shl1(int keep, int v)  // lane l <- lane l+1 ; lane 63 <- keep
{
    return __builtin_amdgcn_update_dpp(keep, v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ int wave_shr1(int keep, int v)  // lane l <- lane l-1 ; lane 0 <- keep
{
    return __builtin_amdgcn_update_dpp(keep, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }

// uniform per-task values
struct Tk {
    gcptr a2, an, b2, bn;
    int64_t a_base, b_base, end_a;
    int alen, blen, begin_a, begin_b, X, band, Y;
    bool fs, fe;
    int iA;       // first row of the pos==end_a anti-diagonal scan (:192)
    int eaRel;    // end_a - begin_a + band clamped to int: band column of pos==end_a in row 0
    gptr dir;
    giptr h0row, pos0, lastrow, adh;
    // direction-free fill: blocks [df_lo, df_hi) (multiples of 4) keep no directions; instead the live row of every
    // 4th block start goes to ckpt and the values crossing every 4th lane boundary to bnd (see do_block_df)
    gptr ckpt, bnd;
    int df_lo, df_hi;
};


// ---- per-wave LDS rings of pre-expanded sequence operands ---------------------------------------------------
// Every row of every lane needs two operands derived from the sequences: the one-hot byte of the base entering
// its a-window and the score row of its b base.  Expanding them costs 6 vector instructions per row per lane
// (20 issue cycles of ~320).  Instead 16 lanes expand the 16 new a and b bases of a block ONCE, one block
// ahead, into two LDS rings, and every lane fetches its operands with two ds_reads per row: the LDS pipe is
// otherwise idle, so this takes the work off the vector ALU.
//   ring A: entry k = one-hot (bytes for dot4, nibbles for dot8) of a[A0 + k], A0 = a_base + begin_a - band; lane l needs
//           k = tau + (C-1)*(l+1) at row-time tau.  With 17 columns per lane it is stored transposed,
//           pos = (k%16)*72 + (k/16)%72, so the 64 lanes of a read (k = k0 + 16*l) hit 64 consecutive dwords (no
//           bank conflict) and the 16 rows of a block are 16 compile-time offsets from one per-lane address.
//   ring B: entry k = score row of b[b_base + begin_b + k]; lane l needs k = tau - l.  128 entries + a copy
//           of the first 16 behind them so that the 16 rows of a block never wrap.
//   (Kernels with fewer columns per lane -- lane stride C-1 not a multiple of 16 -- use the same rings with a
//   plain layout pos = k % size plus a 16-entry copy behind the ring; their ring-A reads are 2..8-way bank
//   conflicted, which the otherwise idle LDS pipe absorbs.)
constexpr int RING_A = 16 * 72;  // 1152 entries: the transposed ring of the 17-column kernels (72 columns of 16)
constexpr int RING_B = 128;
__shared__ u32 s_ringA[RING_A + ROWS];
__shared__ u32 s_ringB[RING_B + ROWS];
__shared__ u32 s_bnd_full[512];  // boundary values of one block on their way out, see do_block_df
__shared__ int s_cap[2][2][20];  // side-capture scratch [pos==0 | pos==end_a][lane parity], see do_block

// ring A geometry per column count.  Plain layout: size = power of two >= 64*(C-1) + 32.  Transposed layout (C = 17):
// 16 rows of COLS = 72 dwords, entry k at (k%16)*COLS + (k/16)%COLS (72 >= 1056/16 + 2; a multiple of 8 keeps the
// 64 lanes of a read on 64 distinct banks).
template <int C>
struct RingA {
    static constexpr bool transposed = ((C - 1) % 16 == 0);
    static constexpr int span = 64 * (C - 1) + 32;
    static constexpr int size = span <= 128 ? 128 : span <= 256 ? 256 : span <= 512 ? 512 : 1024;
    static constexpr int COLS = 72;
    static_assert(transposed ? (span <= 16 * (COLS - 2)) : (span <= size && size + ROWS <= RING_A + ROWS), "ring A too small");
    // position of entry k
    static __device__ __forceinline__ int pos(int k)
    {
        if (transposed) return (k & 15) * COLS + (int)((u32)(k >> 4) % (u32)COLS);
        return k & (size - 1);
    }
    // write entry k (plain layout keeps a copy of the first 16 entries behind the ring so that the 16 rows of a
    // block can be read at immediate offsets without wrapping)
    static __device__ __forceinline__ void put(int k, u32 v)
    {
        const int p = pos(k);
        s_ringA[p] = v;
        if (!transposed && p < ROWS) s_ringA[p + size] = v;
    }
};

template <bool HASN>
__device__ __forceinline__ u32 enc_a(u32 code2, bool isn)
{
    // N-aware: one-hot NIBBLE (value 4) per letter A T C G N + a constant 10 in nibble 5, for v_dot8_u32_u4
    if (HASN) return (4u << ((isn ? 4u : code2) * 4u)) | (12u << 20) | (1u << 24);
    return 1u << (code2 * 8u);  // one-hot byte per base, for v_dot4_u32_u8
}
template <bool HASN>
__device__ __forceinline__ u32 enc_b(u32 code2, bool isn)
{
    // what a diag step adds: 4*(S(a,b)+16)+2 = 86 match (N-N included), 66 N-vs-base, 50 mismatch.
    // dot4 form: byte(a) = that value.  dot8 form: 48 = 12*4 from constant nibble 5 and the tag 2 = 1*2 from constant
    // nibble 6 (so that the direction-free blocks can drop the tag by clearing that nibble), + 4*9 = 36 on a match,
    // + 4*4 = 16 when exactly one side is N.
    if (HASN) return (isn ? (0x4444u | (9u << 16)) : ((9u << (code2 * 4u)) | (4u << 16))) | (4u << 20) | (2u << 24);
    return 0x32323232u + (0x24u << (code2 * 8u));
}

// the diag-step tag inside a ring-B entry: the direction-free kernels keep their ring untagged (their hot blocks use it
// as it is, their few tagged blocks add the tag per row); the other kernels keep it tagged
template <bool HASN>
constexpr u32 RING_TAG = HASN ? (2u << 24) : 0x02020202u;

// lanes 0..15 expand the 16 new entries of the block whose first row-time is T
template <int C, bool HASN, bool UNTAGGED>
__device__ __forceinline__ void ring_produce(const int T, const int lane, const u32 aw, const u32 anw, const u32 bw, const u32 bnw)
{
    if (lane < ROWS) {
        const int kA = T + (C - 1) * 64 + lane;
        RingA<C>::put(kA, enc_a<HASN>((aw >> (2 * lane)) & 3u, HASN && ((anw >> lane) & 1u)));
        const u32 brow = enc_b<HASN>((bw >> (2 * lane)) & 3u, HASN && ((bnw >> lane) & 1u)) - (UNTAGGED ? RING_TAG<HASN> : 0u);
        const int pb = (T + lane) & (RING_B - 1);
        s_ringB[pb] = brow;
        if (pb < ROWS) s_ringB[pb + RING_B] = brow;
    }
}

// Out-of-line device functions receive their arguments in vector registers, so the compiler has to assume
// they differ per lane.  Everything in Tk is wave-uniform: re-assert that (v_readfirstlane) once per call so
// that the callee computes addresses, loop counters and the whole traceback walk on the scalar unit.
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v)
{
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(u64)v);
    const u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)((u64)v >> 32));
    return (int64_t)(((u64)hi << 32) | lo);
}
template <class P>
__device__ __forceinline__ P unip(P p) { return (P)(u64)uni64((int64_t)(u64)p); }
__device__ __forceinline__ Tk load_uniform(const Tk* tp)
{
    Tk t = *tp;
    t.a2 = unip(t.a2); t.an = unip(t.an); t.b2 = unip(t.b2); t.bn = unip(t.bn);
    t.a_base = uni64(t.a_base); t.b_base = uni64(t.b_base); t.end_a = uni64(t.end_a);
    t.alen = uni(t.alen); t.blen = uni(t.blen); t.begin_a = uni(t.begin_a); t.begin_b = uni(t.begin_b);
    t.X = uni(t.X); t.band = uni(t.band); t.Y = uni(t.Y);
    t.fs = uni(t.fs) != 0; t.fe = uni(t.fe) != 0;
    t.iA = uni(t.iA); t.eaRel = uni(t.eaRel);
    t.dir = unip(t.dir); t.h0row = unip(t.h0row); t.pos0 = unip(t.pos0); t.lastrow = unip(t.lastrow); t.adh = unip(t.adh);
    t.ckpt = unip(t.ckpt); t.bnd = unip(t.bnd); t.df_lo = uni(t.df_lo); t.df_hi = uni(t.df_hi);
    return t;
}

template <int C>
__device__ __forceinline__ u64 dir_index(int blk, int lane, int c)
{
    constexpr int G = C / 4, REM = C % 4;
    const u64 base = (u64)blk * (u64)(C * 64);
    if (c < 4 * G) return base + (u64)((c >> 2) * 256 + lane * 4 + (c & 3));
    return base + (u64)(G * 256 + lane * REM + (c - 4 * G));
}

// ---- one block of 16 row-times --------------------------------------------------------------------
// MODE bit 0 (TOP): lanes may still be before row 1, cells with pos <= 0 exist (the reference's pos==0 rules).
// MODE bit 1 (END): rows of the pos==end_a anti-diagonal and/or the last row are in the block (side captures).
enum { M_FAST = 0, M_TOP = 1, M_END = 2, M_BOTH = 3 };

// which kernel variants run their fast blocks without directions (do_block_df / materialise): the tuned band-512 ones
// strip geometry of the direction-free kernels: SL lanes wide, boundary values stored for every SL-th lane,
// 64/SL groups re-enacted per materialise() call
#ifndef GAMDP_STRIP_LANES
#define GAMDP_STRIP_LANES 4
#endif
constexpr int SL = GAMDP_STRIP_LANES, SLOG = (SL == 4) ? 2 : 1, NB = 64 / SL;  // lanes per strip, log2, boundaries per row-time
static_assert(SL == 2 || SL == 4, "strips are 2 or 4 lanes wide");
constexpr u32 BND_WORDS = 2u * NB * 16u;  // boundary words per block: [received | handed][lane/SL][row-time 16]

template <int CE, int C, bool HASN>
constexpr bool DIRFREE_OK = CE >= 0 && CE < C - 1 && C - 1 <= 16 && C >= 9;  // (a loss with 5 columns per lane: 20-column strips, per-row work dominates)

template <int C, int CE, bool HASN, int MODE>
__device__ __forceinline__ void do_block(int (&Lp)[C], u32 (&acc)[C], u32 (&W)[C + 15], int& Lin, int& Lout, const Tk& t,
                                         const int blk, const int lane, const int LE, const int kill_c)
{
    const int tau0 = blk * ROWS;

    // operands come from the LDS rings: one per-lane address per ring and block, the 16 rows are immediate offsets
    const u32* ringA_lane;  // k = tau0 + r + (C-1)*(lane+1)
    if (RingA<C>::transposed) ringA_lane = s_ringA + (u32)((tau0 + (C - 1) * (lane + 1)) >> 4) % (u32)RingA<C>::COLS;
    else ringA_lane = s_ringA + ((tau0 + (C - 1) * (lane + 1)) & (RingA<C>::size - 1));
    const u32* ringB_lane = s_ringB + ((tau0 - lane) & (RING_B - 1));  // k = tau0 + r - lane

    int xkeep = NEG;  // lane 63's `up` hand-off source does not exist
    const u32 tagK = (CE >= 0 && CE < C - 1 && lane == LE) ? 0x80000001u : 1u;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        W[C - 1 + r] = ringA_lane[RingA<C>::transposed ? r * RingA<C>::COLS : r];
        const u32 brow = ringB_lane[r] + (DIRFREE_OK<CE, C, HASN> ? RING_TAG<HASN> : 0u);

        // per-row values of the special modes
        constexpr bool TOP = (MODE & M_TOP) != 0, END = (MODE & M_END) != 0;
        const int row = tau0 + r - lane;
        bool act = true;  // lanes before their row 1 (TOP) or past their last row (END) sit the row out
        int cm1 = 0, Zst = 0, ZL = 0;
        if (END) act = row <= t.X - 1;
        if (TOP) {
            act = act && row >= 1;
            cm1 = (t.band - t.begin_a - row - 1) - C * lane;  // column whose pos == -1
            Zst = 32 * row + 32 * (t.band - t.begin_a - 1);   // G4 of H = 0 at the pos == -1 cell
            ZL = (t.fs && row > FORCE_MAXGAP) ? NEG : Zst;    // ... as a `left` source (:150-155)
        }

        int L = Lin;
        int x;
        // (Computing all diag candidates of the row up front, ahead of the max3 chain, removes the s_nops the
        // compiler pads the dot4 -> VALU hazard with, but measured 3 % slower: the interleaved form below gives
        // each wave independent work between the dependent max3 -> and -> max3 steps.)
        auto cell = [&](const int c) __attribute__((always_inline)) {
            int D;
            if (HASN) D = (int)__builtin_amdgcn_udot8(W[r + c], brow, (u32)Lp[c], false);
            else D = (int)__builtin_amdgcn_udot4(W[r + c], brow, (u32)Lp[c], false);
            // tag the `up` source; the tuned kernels drop the `up` of the last band column (static position CE of
            // lane LE) in the same v_or: tagK also sets the sign bit there (all live G4 values are >= 0)
            int Uc = (c < C - 1) ? (int)((u32)Lp[(c < C - 1) ? c + 1 : c] | ((CE >= 0 && c == CE) ? tagK : 1u)) : (x | 1);
            if (CE < 0) Uc = (c == kill_c) ? NEG : Uc;
            const int R = imax3(D, Uc, L);
            acc[c] = __builtin_amdgcn_alignbit((u32)R, acc[c], 2);
            const int Lc = R & ~3;
            if (TOP) {
                const bool m1 = (cm1 == c);  // the pos == -1 cell holds H = 0 (the reference's zero-initialised matrix)
                Lp[c] = m1 ? Zst : Lc;
                L = m1 ? ZL : Lc;
            } else {
                Lp[c] = Lc;
                L = Lc;
            }
        };

        if (MODE == M_FAST || act) cell(0);
        x = wave_shl1(xkeep, Lp[0]);
        xkeep = x;
        // The last band column has no `up` source.  When it is the last column of lane LE the value arriving
        // from lane LE+1 must be dropped; otherwise (tuned kernels with CE < C-1) column C-1 of lane LE lies
        // outside the band and whatever arrives only feeds dead cells.
        if (CE < 0 || CE == C - 1) x = (lane >= LE) ? NEG : x;
        if (MODE == M_FAST || act) {
#pragma unroll
            for (int c = 1; c < C; ++c) cell(c);
            Lout = L;
        }
        // a lane that sits the row out hands over its last real chain value (its right neighbour is one row behind)
        Lin = wave_shr1(Lin, (MODE == M_FAST) ? L : Lout);

        // Side captures.  Every lane works on its own row, and the column of that row's pos==0 (pos==end_a) cell
        // moves by C-1 from lane to lane, so at any row-time at most two ADJACENT lanes hold such a cell.  Those
        // lanes drop their row values into a tiny LDS buffer (one per lane parity) and pick the one they need by
        // index (a dynamic register index would cost a compare + select per cell for every lane).
        if (TOP) {
            const int c0 = cm1 + 1;  // column of the pos == 0 cell
            if (act && row <= t.X - 1 && c0 >= 0 && c0 < C) {
#pragma unroll
                for (int c = 0; c < C; ++c) s_cap[0][lane & 1][c] = Lp[c];
                const int j0 = C * lane + c0;
                if (j0 < t.Y) t.pos0[row] = (s_cap[0][lane & 1][c0] >> 2) - 16 * row - 8 * j0;
            }
        }
        if (END) {
            const int cE = (t.eaRel - row) - C * lane;  // column whose pos == end_a
            if (row >= 1 && row <= t.X - 1 && cE >= 0 && cE < C) {
#pragma unroll
                for (int c = 0; c < C; ++c) s_cap[1][lane & 1][c] = Lp[c];
                const int jE = C * lane + cE;
                if (jE < t.Y) t.adh[row - t.iA] = (s_cap[1][lane & 1][cE] >> 2) - 16 * row - 8 * jE;
            }
        }
    }
    if (MODE & M_END) {
        // lanes stopped updating after their last row, so the registers still hold row X-1 when it fell in this block
        const int rX = t.X - 1 - (tau0 - lane);  // position of the lane's last row in this block
        if (t.X > 1 && rX >= 0 && rX < ROWS) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int j = C * lane + c;
                if (j < t.Y) t.lastrow[j] = (Lp[c] >> 2) - 16 * (t.X - 1) - 8 * j;
            }
            // the direction words were shifted in from the top one row at a time: a lane that stopped at row rX
            // has its rows 2*(15-rX) bits too high
            const u32 sh = 2u * (u32)(ROWS - 1 - rX);
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] >>= sh;
        }
    }

    // direction words of this block: 16 B / lane coalesced
    {
        constexpr int G = C / 4, REM = C % 4;
        gptr blkp = t.dir + (u64)blk * (u64)(C * 64);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            u32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            *(g4ptr)(blkp + g * 256 + lane * 4) = v;
        }
#pragma unroll
        for (int e = 0; e < REM; ++e) blkp[G * 256 + lane * REM + e] = acc[4 * G + e];
    }
    // slide the a window
#pragma unroll
    for (int k = 0; k < C - 1; ++k) W[k] = W[k + ROWS];
}

// ---- a fast block that keeps no directions ------------------------------------------------------------------
// The traceback reads the directions of the cells on the path only, so the fast blocks of the tuned kernels compute
// plain values -- v_dot4 + v_max3 per cell, no tag, no direction word -- and store what materialise() below needs to
// re-enact a 4-lane strip of the sweep later, with directions, around the path:
//   ckpt: at the start of every 4th block (a "group" = 64 row-times) the live row of every lane;
//   bnd:  every row-time, for every 4th lane, the chain value it received from its left neighbour and the value it
//         hands to that neighbour (its new column 0): [block][received | handed][lane/4 16][row-time 16], written 16 B
//         (four row-times) at a time so that a strip later reads its 16 row-times of a block as one 64 B line.
// Values stay multiples of 4 (what the tagged blocks keep after stripping), so fast and slow blocks mix freely.
template <int C, int CE, bool HASN>
__device__ __forceinline__ void do_block_df(int (&Lp)[C], u32 (&W)[C + 15], int& Lin, int& Lout, const Tk& t, const int blk, const int lane, const int LE)
{
    static_assert(CE >= 0 && CE < C - 1, "tuned kernels only");
    const int tau0 = blk * ROWS;
    const u32* ringA_lane;
    if (RingA<C>::transposed) ringA_lane = s_ringA + (u32)((tau0 + (C - 1) * (lane + 1)) >> 4) % (u32)RingA<C>::COLS;
    else ringA_lane = s_ringA + ((tau0 + (C - 1) * (lane + 1)) & (RingA<C>::size - 1));
    const u32* ringB_lane = s_ringB + ((tau0 - lane) & (RING_B - 1));
    if ((blk & 3) == 0) {
        constexpr int G = C / 4, REM = C % 4;
        gptr ck = t.ckpt + (u64)(blk >> 2) * (u64)(C * 64);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            u32x4 v = {(u32)Lp[4 * g], (u32)Lp[4 * g + 1], (u32)Lp[4 * g + 2], (u32)Lp[4 * g + 3]};
            *(g4ptr)(ck + g * 256 + lane * 4) = v;
        }
#pragma unroll
        for (int e = 0; e < REM; ++e) ck[G * 256 + lane * REM + e] = (u32)Lp[4 * G + e];
    }
    // boundary values go through LDS ([received | handed][lane/SL][row-time 16]) and leave as one coalesced 2 KB
    // store per block: scattered 16 B stores from the edge lanes cost the fill 8 %
    static_assert(BND_WORDS == 512, "s_bnd and the block-end store assume 4-lane strips");
    u32* const sb = s_bnd_full + (u32)(lane >> SLOG) * 16u;
    const bool edge = (lane & (SL - 1)) == 0;
    const u32 killK = (lane == LE) ? 0x80000000u : 0u;  // the last band column has no `up` source
    int xkeep = NEG;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        W[C - 1 + r] = ringA_lane[RingA<C>::transposed ? r * RingA<C>::COLS : r];
        const u32 brow = ringB_lane[r];  // untagged in these kernels
        const int Lrecv = Lin;
        int L = Lin, x = NEG;
        auto cell = [&](const int c) __attribute__((always_inline)) {
            const int D = HASN ? (int)__builtin_amdgcn_udot8(W[r + c], brow, (u32)Lp[c], false)
                               : (int)__builtin_amdgcn_udot4(W[r + c], brow, (u32)Lp[c], false);
            int Uc = (c < C - 1) ? Lp[(c < C - 1) ? c + 1 : c] : x;
            if (c == CE) Uc = (int)((u32)Uc | killK);
            L = imax3(D, Uc, L);
            Lp[c] = L;
        };
        cell(0);
        x = wave_shl1(xkeep, Lp[0]);
        xkeep = x;
        if (edge) { sb[r] = (u32)Lrecv; sb[NB * 16 + r] = (u32)Lp[0]; }
#pragma unroll
        for (int c = 1; c < C; ++c) cell(c);
        Lout = L;
        Lin = wave_shr1(Lin, L);
    }
    {
        const u32x4 v0 = *(const u32x4*)(s_bnd_full + lane * 8), v1 = *(const u32x4*)(s_bnd_full + lane * 8 + 4);
        gptr bp = t.bnd + (u64)blk * BND_WORDS + (u32)lane * 8u;
        *(g4ptr)bp = v0;
        *(g4ptr)(bp + 4) = v1;
    }
#pragma unroll
    for (int k = 0; k < C - 1; ++k) W[k] = W[k + ROWS];
}

template <int C>
struct BlockState {
    int Lp[C];
    u32 acc[C];
    u32 W[C + 15];
    int Lin;
    int Lout;  // last chain value of the lane's last column (what it hands to its right neighbour)
};

// The fill is driven through three out-of-line functions that hand the per-lane register state over in a
// BlockState (private memory): init_row0 (phase A), slow_block (one block of the top-left triangle / ramp-up /
// anti-diagonal and last-row capture region: 2 % of the blocks of a 50 kb pair) and fast_range (a run of
// consecutive plain blocks -- the hot loop).  Keeping them separate functions gives the hot loop its own
// register allocation: nothing live in the other phases can force a spill (and with it a full
// `s_waitcnt vmcnt(0)` drain of the outstanding direction stores) into it.
template <int C>
__device__ __forceinline__ void load_state(const BlockState<C>* st, int (&Lp)[C], u32 (&acc)[C], u32 (&W)[C + 15], int& Lin, int& Lout)
{
#pragma unroll
    for (int c = 0; c < C; ++c) { Lp[c] = st->Lp[c]; acc[c] = st->acc[c]; }
#pragma unroll
    for (int k = 0; k < C - 1; ++k) W[k] = st->W[k];
#pragma unroll
    for (int k = C - 1; k < C + 15; ++k) W[k] = 0;
    Lin = st->Lin;
    Lout = st->Lout;
}
template <int C>
__device__ __forceinline__ void store_state(BlockState<C>* st, const int (&Lp)[C], const u32 (&acc)[C], const u32 (&W)[C + 15], const int Lin, const int Lout)
{
#pragma unroll
    for (int c = 0; c < C; ++c) { st->Lp[c] = Lp[c]; st->acc[c] = acc[c]; }
#pragma unroll
    for (int k = 0; k < C - 1; ++k) st->W[k] = W[k];
    st->Lin = Lin;
    st->Lout = Lout;
}

template <int C, int CE, bool HASN, int MODE>
__device__ __noinline__ void slow_block(BlockState<C>* st, const Tk* tp, const int blk_, const int lane)
{
    const Tk t = load_uniform(tp);
    const int blk = uni(blk_);
    const int LE = (t.Y - 1) / C;
    const int kill_c = (CE < 0 && lane == LE) ? (t.Y - 1) % C : -1;
    int Lp[C];
    u32 acc[C];
    u32 W[C + 15];
    int Lin, Lout;
    load_state<C>(st, Lp, acc, W, Lin, Lout);
    {
        // operands of this block are already in the LDS rings; expand the next block's 16 new bases
        const int T = (blk + 1) * ROWS;
        const int64_t ia = t.a_base + t.begin_a - t.band + T + (C - 1) * 64, ib = t.b_base + t.begin_b + T;
        ring_produce<C, HASN, DIRFREE_OK<CE, C, HASN>>(T, lane, fetch16(t.a2, ia), HASN ? fetch16n(t.an, ia) : 0u, fetch16(t.b2, ib),
                              HASN ? fetch16n(t.bn, ib) : 0u);
    }
    do_block<C, CE, HASN, MODE>(Lp, acc, W, Lin, Lout, t, blk, lane, LE, kill_c);
    store_state<C>(st, Lp, acc, W, Lin, Lout);
}

// blocks [blk_begin, blk_end) are all "fast": every lane is past row 0, no pos <= 0 cell, no capture row
template <int C, int CE, bool HASN, bool DF>
__device__ __noinline__ void fast_range(BlockState<C>* st, const Tk* tp, const int blk_begin_, const int blk_end_, const int lane)
{
    const Tk t = load_uniform(tp);
    const int blk_begin = uni(blk_begin_), blk_end = uni(blk_end_);
    const int LE = (t.Y - 1) / C;
    const int kill_c = (CE < 0 && lane == LE) ? (t.Y - 1) % C : -1;
    int Lp[C];
    u32 acc[C];
    u32 W[C + 15];
    int Lin, Lout;
    load_state<C>(st, Lp, acc, W, Lin, Lout);
    // Invariant: on entry of block T the LDS rings hold everything block T reads; the top of block T expands the 16
    // new bases of block T+16 from packed words that were requested one block earlier (so the hot loop never waits
    // on a load it just issued, and never drains the direction stores in flight).
    const int64_t iA0 = t.a_base + t.begin_a - t.band + (C - 1) * 64, iB0 = t.b_base + t.begin_b;
    gcptr pa = t.a2 + (iA0 >> 4), pb = t.b2 + (iB0 >> 4);  // wave-uniform streams, one word per block
    const u32 sha = (u32)(iA0 & 15) * 2u, shb = (u32)(iB0 & 15) * 2u;
    u32 a_lo = pa[blk_begin + 1], a_hi = pa[blk_begin + 2], b_lo = pb[blk_begin + 1], b_hi = pb[blk_begin + 2];
    u32 an_lo = 0, an_hi = 0, bn_lo = 0, bn_hi = 0;
    if (HASN) {
        const int64_t ia = iA0 + (int64_t)(blk_begin + 1) * ROWS, ib = iB0 + (int64_t)(blk_begin + 1) * ROWS;
        an_lo = t.an[ia >> 5]; an_hi = t.an[(ia >> 5) + 1];
        bn_lo = t.bn[ib >> 5]; bn_hi = t.bn[(ib >> 5) + 1];
    }
    asm volatile("" : "+v"(a_lo), "+v"(a_hi), "+v"(b_lo), "+v"(b_hi), "+v"(Lin));
    if (HASN) asm volatile("" : "+v"(an_lo), "+v"(an_hi), "+v"(bn_lo), "+v"(bn_hi));
#pragma unroll
    for (int c = 0; c < C; ++c) asm volatile("" : "+v"(Lp[c]), "+v"(acc[c]));
#pragma unroll
    for (int k = 0; k < C - 1; ++k) asm volatile("" : "+v"(W[k]));
    for (int blk = blk_begin; blk < blk_end; ++blk) {
        const int T = (blk + 1) * ROWS;
        const u32 a_nx = pa[blk + 3], b_nx = pb[blk + 3];  // words of block blk+2, used at the next iteration
        u32 an_lo_nx = 0, an_hi_nx = 0, bn_lo_nx = 0, bn_hi_nx = 0;
        if (HASN) {
            const int64_t ia = iA0 + T + ROWS, ib = iB0 + T + ROWS;
            an_lo_nx = t.an[ia >> 5]; an_hi_nx = t.an[(ia >> 5) + 1];
            bn_lo_nx = t.bn[ib >> 5]; bn_hi_nx = t.bn[(ib >> 5) + 1];
        }
        u32 anw = 0, bnw = 0;
        if (HASN) {
            anw = __builtin_amdgcn_alignbit(an_hi, an_lo, (u32)((iA0 + T) & 31)) & 0xFFFFu;
            bnw = __builtin_amdgcn_alignbit(bn_hi, bn_lo, (u32)((iB0 + T) & 31)) & 0xFFFFu;
        }
        ring_produce<C, HASN, DIRFREE_OK<CE, C, HASN>>(T, lane, __builtin_amdgcn_alignbit(a_hi, a_lo, sha), anw,
                              __builtin_amdgcn_alignbit(b_hi, b_lo, shb), bnw);
        if constexpr (DF) do_block_df<C, CE, HASN>(Lp, W, Lin, Lout, t, blk, lane, LE);
        else do_block<C, CE, HASN, M_FAST>(Lp, acc, W, Lin, Lout, t, blk, lane, LE, kill_c);
        a_lo = a_hi; a_hi = a_nx; b_lo = b_hi; b_hi = b_nx;
        if (HASN) { an_lo = an_lo_nx; an_hi = an_hi_nx; bn_lo = bn_lo_nx; bn_hi = bn_hi_nx; }
    }
    store_state<C>(st, Lp, acc, W, Lin, Lout);
}

// ---- directions of a 4-lane strip, on demand --------------------------------------------------------------
// Re-enacts lanes 4q..4q+3 of the sweep over 16 groups (64 row-times each) at once -- lanes 4k..4k+3 of this
// wavefront do group g_hi-k -- from what do_block_df stored, this time with the tagged cell, and writes the direction
// words of those lanes exactly where the tagged fill would have put them.  The walk in finish_task calls it when it
// enters direction-free blocks whose strip is not there yet; an alignment path drifts sideways only by its net indel
// count, so nearly every call serves 1 000 rows of path.
template <int C, int CE, bool HASN>
__device__ __noinline__ void materialise(const Tk* tp, const int q_, const int g_hi_, const int lane)
{
    static_assert(CE >= 0 && CE < C - 1 && C - 1 <= 16, "tuned kernels only");
    constexpr int HR = 8;  // row-times per unrolled chunk
    const Tk t = load_uniform(tp);
    const int q = uni(q_), g_hi = uni(g_hi_);
    const int lam = lane & (SL - 1);
    const int R = SL * q + lam;  // the lane of the fill this lane re-enacts
    const int g_first = t.df_lo >> 2;
    const int g = g_hi - (lane >> SLOG);
    const bool live = g >= g_first;
    const int gg = live ? g : g_first;  // lanes beyond the range redo the first group and store nothing
    const int LE = (t.Y - 1) / C;

    int Lp[C];
    u32 acc[C];
    u32 W[C - 1 + HR];
    u32 sv[HR];
    {
        constexpr int G = C / 4, REM = C % 4;
        gptr ck = t.ckpt + (u64)gg * (u64)(C * 64);
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const u32x4 v = *(g4ptr)(ck + k * 256 + R * 4);
            Lp[4 * k] = (int)v.x; Lp[4 * k + 1] = (int)v.y; Lp[4 * k + 2] = (int)v.z; Lp[4 * k + 3] = (int)v.w;
        }
#pragma unroll
        for (int e = 0; e < REM; ++e) Lp[4 * G + e] = (int)ck[G * 256 + R * REM + e];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] = 0;
    }
    // quad shifts: lane lam <- lam-1 / lam+1 of the same strip (the strip's outer lanes take the stored values)
    auto from_left = [](int v) { return __builtin_amdgcn_update_dpp(v, v, SL == 4 ? 0x90 : 0xA0, 0xf, 0xf, false); };   // quad_perm:[0,0,1,2] | [0,0,2,2]
    auto from_right = [](int v) { return __builtin_amdgcn_update_dpp(v, v, SL == 4 ? 0xF9 : 0xF5, 0xf, 0xf, false); };  // quad_perm:[1,2,3,3] | [1,1,3,3]
    int Lin = from_left(Lp[C - 1]);  // what the left neighbour handed over at the end of the previous row-time
    const int tau_g = gg * 64;
    const int64_t iaW = t.a_base + t.begin_a - t.band + (int64_t)(C - 1) * R + tau_g;  // W[k] <-> a[iaW + k]
    {
        const u32 aw = fetch16(t.a2, iaW), awn = HASN ? fetch16n(t.an, iaW) : 0u;
#pragma unroll
        for (int k = 0; k < C - 1; ++k) W[k] = enc_a<HASN>((aw >> (2 * k)) & 3u, HASN && ((awn >> k) & 1u));
    }
    const u32 tagK = (R == LE) ? 0x80000001u : 1u;
    // the stored boundary values this lane consumes: lam 0 the chain value entering the strip from the left, lam 3 the
    // `up` hand-off entering from the right (none right of lane 63); lam 1, 2 load the left one and ignore it
    const bool right_edge = lam == SL - 1;
    const bool has_right = q < NB - 1;
    gptr sp = t.bnd + (u64)gg * (4u * BND_WORDS) + (u32)((right_edge && has_right) ? NB * 16 + (q + 1) * 16 : q * 16);
    static_assert(HR == 8, "the stream is loaded 4 row-times (16 B) at a time");
    auto load_stream4 = [&](const int chunk, const int half) {  // row-times 4*half .. +3 of a chunk
        const u32x4 v = *(g4ptr)(sp + (u32)(chunk >> 1) * BND_WORDS + (u32)(chunk & 1) * 8u + (u32)half * 4u);
        sv[4 * half] = v.x; sv[4 * half + 1] = v.y; sv[4 * half + 2] = v.z; sv[4 * half + 3] = v.w;
    };
    load_stream4(0, 0);
    load_stream4(0, 1);

    // packed words of the chunk after the current one are requested a chunk ahead (nothing here waits on a load it
    // has just issued)
    const int64_t ia_new = iaW + (C - 1), ib_new = t.b_base + t.begin_b + tau_g - R;
    u32 an_nx = fetch16(t.a2, ia_new), bw_nx = fetch16(t.b2, ib_new);
    u32 ann_nx = HASN ? fetch16n(t.an, ia_new) : 0u, bwn_nx = HASN ? fetch16n(t.bn, ib_new) : 0u;
    for (int ch = 0; ch < 64 / HR; ++ch) {
        const u32 an = an_nx, bw = bw_nx, ann = ann_nx, bwn = bwn_nx;
        const int nx = min(ch + 1, 64 / HR - 1) * HR;  // the last chunk re-reads itself
        an_nx = fetch16(t.a2, ia_new + nx);
        bw_nx = fetch16(t.b2, ib_new + nx);
        if (HASN) { ann_nx = fetch16n(t.an, ia_new + nx); bwn_nx = fetch16n(t.bn, ib_new + nx); }
#pragma unroll
        for (int r = 0; r < HR; ++r) {
            W[C - 1 + r] = enc_a<HASN>((an >> (2 * r)) & 3u, HASN && ((ann >> r) & 1u));
            const u32 brow = enc_b<HASN>((bw >> (2 * r)) & 3u, HASN && ((bwn >> r) & 1u));
            const int s = (int)sv[r];
            int L = (lam == 0) ? s : Lin;
            int x = NEG;
            auto cell = [&](const int c) __attribute__((always_inline)) {
                const int D = HASN ? (int)__builtin_amdgcn_udot8(W[r + c], brow, (u32)Lp[c], false)
                                   : (int)__builtin_amdgcn_udot4(W[r + c], brow, (u32)Lp[c], false);
                const int Uc = (c < C - 1) ? (int)((u32)Lp[(c < C - 1) ? c + 1 : c] | ((c == CE) ? tagK : 1u)) : (x | 1);
                const int Rv = imax3(D, Uc, L);
                acc[c] = __builtin_amdgcn_alignbit((u32)Rv, acc[c], 2);
                L = Rv & ~3;
                Lp[c] = L;
            };
            cell(0);
            x = from_right(Lp[0]);
            if (right_edge) x = has_right ? s : NEG;
#pragma unroll
            for (int c = 1; c < C; ++c) cell(c);
            Lin = from_left(L);
            // the same rows of the next chunk: in flight while this chunk computes
            if (r == 3) load_stream4(nx / HR, 0);
            if (r == 7) load_stream4(nx / HR, 1);
        }
#pragma unroll
        for (int k = 0; k < C - 1; ++k) W[k] = W[k + HR];
        if ((ch & 1) && live) {
            constexpr int G = C / 4, REM = C % 4;
            gptr blkp = t.dir + (u64)(gg * 4 + (ch >> 1)) * (u64)(C * 64);
#pragma unroll
            for (int k = 0; k < G; ++k) {
                u32x4 v = {acc[4 * k], acc[4 * k + 1], acc[4 * k + 2], acc[4 * k + 3]};
                *(g4ptr)(blkp + k * 256 + R * 4) = v;
            }
#pragma unroll
            for (int e = 0; e < REM; ++e) blkp[G * 256 + R * REM + e] = acc[4 * G + e];
        }
    }
}

// ---- phases C + D: end-cell search and traceback ------------------------------------------------------
// Kept out of line (like the slow block) so that its registers -- and the scalar registers holding its many
// compare masks -- are allocated separately from the hot fill loop.
template <int C, int CE, bool HASN>
__device__ __noinline__ void finish_task(const Tk* tp, const DevTask* dtp_, const LaunchParams* pp, const int lane)
{
    const Tk t = load_uniform(tp);
    const DevTask* dtp = unip(dtp_);
    const u32 dt_flags = (u32)uni((int)dtp->flags), dt_res_idx = (u32)uni((int)dtp->res_idx);
    const u64 dt_ops_off = (u64)uni64((int64_t)dtp->ops_off), dt_ops_cap = (u64)uni64((int64_t)dtp->ops_cap);
    uint8_t* const p_ops_buf = unip(pp->ops_buf);
    DevResult* const p_results = unip(pp->results);
    const int X = t.X, Y = t.Y, w = t.band;
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop L1 lines cached by the slot's previous task

    // ---- phase C: end cell (:174-212), first maximum in scan order wins --------------------------------
    int best = NEG, bkey = 0x7fffffff;  // key = scan position
    {
        const giptr lr = (X == 1) ? t.h0row : t.lastrow;
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
    // after the wave reduction every lane holds the same (best, bkey): make that explicit so that the whole
    // walk below is scalar code (SALU) and does not take vector-issue slots from the waves still filling
    best = __builtin_amdgcn_readfirstlane(best);
    bkey = __builtin_amdgcn_readfirstlane(bkey);
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
            const bool want_ops = dt_flags & TF_WANT_OPS;
            uint8_t* ops = p_ops_buf + dt_ops_off;
            u32 len = 0, nm = 0;
            bool have_last = false, have_first = false;
            int la = 0, lb = 0, fa = 0, fb = 0;
            int l = y / C, c = y - l * C;
            // Direction-word cache for the walk: lane k (< TB_DEPTH) holds, for block (cblk0 - k), the 4-column
            // group `cg` of lane-row `cl`.  A diagonal run stays in one (lane-row, column) and walks down the
            // blocks, a gap moves one column sideways (3 times out of 4 inside the same group), so one refill
            // (one 16 B load per lane, all in flight together) serves several dependent steps of the walk.
            constexpr int TB_DEPTH = 16;
            int cblk0 = -1, cl = -1, cg = -1;
            u32 cw = 0;  // lane 4*k + e holds the direction word of block (cblk0 - k), column 4*cg + e
            // Packed-sequence window for the match counting: lane k holds 2-bit word (w0 - k); the walk moves towards
            // lower indices, so one refill (one coalesced 256 B load) covers the next ~1000 bases.
            int64_t sa_w0 = INT64_MIN, sb_w0 = INT64_MIN, san_w0 = INT64_MIN, sbn_w0 = INT64_MIN;
            u32 sa_v = 0, sb_v = 0, san_v = 0, sbn_v = 0;  // (the N-plane windows only exist in the N-aware kernels)
            // strip of direction-free blocks whose directions materialise() has produced: lanes 4*mat_q .. +3, blocks
            // mat_lo .. mat_hi
            int mat_q = -1, mat_lo = 0, mat_hi = -1, mat_calls = 0;
            int old_q = -1, old_lo = 0, old_hi = -1;  // the strip materialised before that one
            long long mat_ticks = 0;
            int dg_iters = 0, dg_refills = 0;
            const long long walk_t0 = (dt_flags & TF_DIAG_COUNT_MAT) ? wall_clock64() : 0;
            int cvalid_lo = 0;  // cached blocks below this one are not usable (direction-free, not materialised)
            auto get_word = [&](const int blk, const int l_, const int c_) -> u32 {
                const int g = c_ >> 2;
                int k = cblk0 - blk;
                if (l_ != cl || g != cg || k < 0 || k >= TB_DEPTH || blk < cvalid_lo) {
                    cvalid_lo = 0;
                    dg_refills++;
                    if constexpr (DIRFREE_OK<CE, C, HASN>) {
                        if (blk >= t.df_hi) {
                            cvalid_lo = t.df_hi;  // the refill may reach down into direction-free blocks: not usable
                        } else if (blk >= t.df_lo) {
                            if ((l_ >> SLOG) == old_q && blk >= old_lo && blk <= old_hi) {
                                // back in the strip the walk came from (a path sitting on a strip border): its words
                                // are still in memory
                                const int tq = mat_q, tl = mat_lo, th = mat_hi;
                                mat_q = old_q; mat_lo = old_lo; mat_hi = old_hi;
                                old_q = tq; old_lo = tl; old_hi = th;
                            } else if (!((l_ >> SLOG) == mat_q && blk >= mat_lo && blk <= mat_hi)) {
                                const int g_hi = blk >> 2;
                                const long long tm0 = (dt_flags & TF_DIAG_COUNT_MAT) ? wall_clock64() : 0;
                                materialise<C, (DIRFREE_OK<CE, C, HASN> ? CE : 0), HASN>(tp, l_ >> SLOG, g_hi, lane);
                                if (dt_flags & TF_DIAG_COUNT_MAT) mat_ticks += wall_clock64() - tm0;
                                // the loads below must see those stores: wait until L2 has them, then drop this CU's L1
                                // lines (an agent-scope release would write back the whole L2 of the XCD, far too much)
                                __builtin_amdgcn_s_waitcnt(0);
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                                old_q = mat_q; old_lo = mat_lo; old_hi = mat_hi;
                                mat_q = l_ >> SLOG; mat_hi = 4 * g_hi + 3; mat_lo = max(4 * (g_hi - (NB - 1)), t.df_lo);
                                mat_calls++;
                            }
                            cvalid_lo = (mat_lo == t.df_lo) ? 0 : mat_lo;  // below df_lo the tagged blocks are all there
                        }
                    }
                    cblk0 = blk; cl = l_; cg = g; k = 0;
                    // one coalesced load: each group of 4 lanes reads the 16 B of one block.  No lane-dependent branch
                    // (it would make the compiler treat the whole walk as divergent): blocks below 0 and columns past
                    // the lane's last one just re-read a valid word
                    const int myblk = max(blk - (lane >> 2), 0);
                    const int myc = min(4 * g + (lane & 3), C - 1);
                    cw = t.dir[dir_index<C>(myblk, l_, myc)];
                }
                return (u32)__builtin_amdgcn_readlane((int)cw, __builtin_amdgcn_readfirstlane(4 * k + (c_ & 3)));
            };
            if (dt_flags & TF_DIAG_SKIP_TRACEBACK) {
                if constexpr (DIRFREE_OK<CE, C, HASN>) {
                    // timing diagnostics: with GAMDP_DIAG_COUNT_MAT as well, do the strip materialisations a walk down
                    // the middle of the band would ask for, and nothing else
                    if (dt_flags & TF_DIAG_COUNT_MAT)
                        for (int g_hi = (t.df_hi >> 2) - 1; g_hi >= (t.df_lo >> 2); g_hi -= NB)
                            materialise<C, (DIRFREE_OK<CE, C, HASN> ? CE : 0), HASN>(tp, (Y / 2) / (SL * C), g_hi, lane);
                }
                x = -1;
            }
            while (x >= 0 && y >= 0 && pos >= 0) {
                // the walk state is wave-uniform by construction; pin it to scalar registers every iteration so
                // the body is selected as SALU code whatever the divergence analysis concluded about the loop
                x = uni(x); y = uni(y); pos = uni(pos); l = uni(l); c = uni(c);
#ifdef GAMDP_DIAG_COUNTERS
                dg_iters = uni(dg_iters) + 1; dg_refills = uni(dg_refills); mat_calls = uni(mat_calls);
#endif
                cblk0 = uni(cblk0); cl = uni(cl); cg = uni(cg);
                mat_q = uni(mat_q); mat_lo = uni(mat_lo); mat_hi = uni(mat_hi); cvalid_lo = uni(cvalid_lo);
                old_q = uni(old_q); old_lo = uni(old_lo); old_hi = uni(old_hi);
                sa_w0 = uni64(sa_w0); sb_w0 = uni64(sb_w0);
                if (HASN) { san_w0 = uni64(san_w0); sbn_w0 = uni64(sbn_w0); }
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
                        const u32 cw = get_word(blk, l, c);
                        const u32 tag = (cw >> ((tau & 15) * 2)) & 3u;
                        op = (tag == 2u) ? 2 : (tag == 1u ? 0 : 1);
                    }
                    if (op == 2) {
                        if (is_match) {
                            nm++;
                            if (!have_last) { have_last = true; la = pos; lb = t.begin_b + x; }
                            have_first = true; fa = pos; fb = t.begin_b + x;
                        }
                        if (want_ops && len < dt_ops_cap) ops[len] = is_match ? 2 : 3;
                        x--; pos--;
                    } else if (op == 1) {  // GAP_B: consumes a
                        if (want_ops && len < dt_ops_cap) ops[len] = 1;
                        y--; pos--;
                        if (--c < 0) { c = C - 1; l--; }
                    } else {  // GAP_A: consumes b
                        if (want_ops && len < dt_ops_cap) ops[len] = 0;
                        x--; y++;
                        if (++c == C) { c = 0; l++; }
                    }
                    len++;
                } else {
                    // interior: consume a whole run of diagonal steps in one go, with all 64 lanes: the lanes that hold
                    // this column's words of the cached blocks find where the run ends (ballot), then 16 bases per
                    // lane are compared straight out of the two packed-sequence windows.  A single wavefront issues
                    // about one instruction per 4 cycles, so the walk is bound by its instruction count, not by the
                    // vector ALU: one iteration per run instead of one per direction word.
                    const int tau = x + l, blk = tau >> 4, r = tau & 15;
                    const u32 w_here = get_word(blk, l, c);  // (re)fills the cache / materialises the strip if needed
                    const int k0 = cblk0 - blk, e = c & 3;
                    const int myk = lane >> 2;
                    const bool mine = ((lane & 3) == e) && myk >= k0 && (cblk0 - myk) >= max(cvalid_lo, 0);
                    u32 T = cw ^ 0xAAAAAAAAu;  // a diagonal step reads 00
                    if (myk == k0) T <<= (30 - 2 * r);  // the block the walk stands in: row r on top
                    const int avail = T ? (__builtin_clz(T) >> 1) : ((myk == k0) ? r + 1 : 16);
                    const u64 m_mine = __ballot(mine), m_stop = __ballot(mine && T != 0u);
                    int n, Ls = -1;
                    if (m_stop == 0) {
                        n = (r + 1) + 16 * (((63 - __builtin_clzll(m_mine)) >> 2) - k0);
                    } else {
                        Ls = __builtin_ctzll(m_stop);
                        const int ks = Ls >> 2;
                        const int lead = __builtin_amdgcn_readlane(avail, Ls);
                        n = (ks == k0) ? lead : (r + 1) + 16 * (ks - k0 - 1) + lead;
                    }
                    const int n_run = n;
                    n = min(n, min(x, pos));  // stay in x >= 1, pos >= 1
                    if (n > 0) {
                        // chunk j (from the top of the run) = bases [hi - 16j - 15, hi - 16j] of both sequences
                        const int64_t ia_hi = t.a_base + pos, ib_hi = t.b_base + t.begin_b + x;
                        const int64_t wa = (ia_hi - 15) >> 4, wb = (ib_hi - 15) >> 4;
                        const int64_t wa_bot = (ia_hi - n - 15) >> 4, wb_bot = (ib_hi - n - 15) >> 4;
                        if (wa + 1 > sa_w0 || wa_bot < sa_w0 - 63) { sa_w0 = wa + 1; sa_v = t.a2[sa_w0 - lane]; }
                        if (wb + 1 > sb_w0 || wb_bot < sb_w0 - 63) { sb_w0 = wb + 1; sb_v = t.b2[sb_w0 - lane]; }
                        const int da = (int)(sa_w0 - wa), db = (int)(sb_w0 - wb);  // lane of chunk 0's low word (>= 1)
                        const u32 a16 = __builtin_amdgcn_alignbit((u32)wave_shr1(0, (int)sa_v), sa_v, (u32)((ia_hi - 15) & 15) * 2u);
                        const u32 b16s = __builtin_amdgcn_alignbit((u32)wave_shr1(0, (int)sb_v), sb_v, (u32)((ib_hi - 15) & 15) * 2u);
                        const u32 b16 = (u32)__builtin_amdgcn_ds_bpermute(4 * (lane + db - da), (int)b16s);
                        const int j = lane - da;
                        const u32 xr = a16 ^ b16;
                        u32 ne = (xr | (xr >> 1)) & 0x55555555u;  // bit 2p set: bases p differ
                        if constexpr (HASN) {
                            // a base that is N on either side counts as a MATCH: the 16 mask bits of this lane's chunk
                            // come out of two 64-word windows of the N planes (32 bases per word), fetched per lane
                            const int64_t na_lo = ia_hi - 15 - 16 * (int64_t)j, nb_lo = ib_hi - 15 - 16 * (int64_t)j;
                            const int64_t wna = (ia_hi >> 5) + 1, wnb = (ib_hi >> 5) + 1;  // highest word needed
                            const int64_t wna_bot = (ia_hi - n - 31) >> 5, wnb_bot = (ib_hi - n - 31) >> 5;
                            if (wna > san_w0 || wna_bot < san_w0 - 63) { san_w0 = wna; san_v = t.an[san_w0 - lane]; }
                            if (wnb > sbn_w0 || wnb_bot < sbn_w0 - 63) { sbn_w0 = wnb; sbn_v = t.bn[sbn_w0 - lane]; }
                            auto nbits = [&](const u32 win, const int64_t w0, const int64_t lo) -> u32 {
                                const int src = (int)(w0 - (lo >> 5));  // lane holding the word with base `lo`
                                const u32 wl = (u32)__builtin_amdgcn_ds_bpermute(4 * src, (int)win);
                                const u32 wh = (u32)__builtin_amdgcn_ds_bpermute(4 * (src - 1), (int)win);
                                return __builtin_amdgcn_alignbit(wh, wl, (u32)(lo & 31)) & 0xFFFFu;
                            };
                            u32 nn = nbits(san_v, san_w0, na_lo) | nbits(sbn_v, sbn_w0, nb_lo);
                            nn = (nn | (nn << 8)) & 0x00FF00FFu;  // spread the 16 bits to the even positions
                            nn = (nn | (nn << 4)) & 0x0F0F0F0Fu;
                            nn = (nn | (nn << 2)) & 0x33333333u;
                            nn = (nn | (nn << 1)) & 0x55555555u;
                            ne &= ~nn;
                        }
                        const int p0 = 16 * j + 16 - n;                 // first base of the chunk that still belongs to the run
                        u32 msk = (p0 <= 0) ? 0x55555555u : ((p0 >= 16) ? 0u : (0x55555555u << (2 * p0)));
                        if (j < 0) msk = 0u;
                        u32 eq = ~ne & msk;
                        const u64 m_eq = __ballot(eq != 0u);
                        if (m_eq) {
                            // sum over the wavefront with DPP adds (a butterfly through ds_bpermute costs six LDS round trips)
                            int cnt = __builtin_popcount(eq);
                            cnt += __builtin_amdgcn_update_dpp(0, cnt, 0xB1, 0xf, 0xf, true);   // quad_perm:[1,0,3,2]
                            cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x4E, 0xf, 0xf, true);   // quad_perm:[2,3,0,1]
                            cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x141, 0xf, 0xf, true);  // row_half_mirror
                            cnt += __builtin_amdgcn_update_dpp(0, cnt, 0x140, 0xf, 0xf, true);  // row_mirror: every lane = its row's sum
                            nm += (u32)(__builtin_amdgcn_readlane(cnt, 0) + __builtin_amdgcn_readlane(cnt, 16) +
                                        __builtin_amdgcn_readlane(cnt, 32) + __builtin_amdgcn_readlane(cnt, 48));
                            const int Ltop = __builtin_ctzll(m_eq), Lbot = 63 - __builtin_clzll(m_eq);
                            const u32 eq_top = (u32)__builtin_amdgcn_readlane((int)eq, Ltop), eq_bot = (u32)__builtin_amdgcn_readlane((int)eq, Lbot);
                            const int off_hi = 16 * (Ltop - da) + 15 - ((31 - __builtin_clz(eq_top)) >> 1);  // steps below the top of the run
                            const int off_lo = 16 * (Lbot - da) + 15 - (__builtin_ctz(eq_bot) >> 1);
                            if (!have_last) { have_last = true; la = pos - off_hi; lb = t.begin_b + x - off_hi; }
                            have_first = true; fa = pos - off_lo; fb = t.begin_b + x - off_lo;
                        }
                        x -= n; pos -= n; len += (u32)n;
                        // the run ended at a gap whose direction word is in the cache: take that step right away
                        if (Ls >= 0 && n == n_run && x >= 1 && pos >= 1) {
                            const u32 w2 = (u32)__builtin_amdgcn_readlane((int)cw, Ls);
                            const u32 tag2 = (w2 >> (((x + l) & 15) * 2)) & 3u;
                            if (tag2 == 1u) {  // GAP_A
                                x--; y++;
                                if (++c == C) { c = 0; l++; }
                            } else {  // GAP_B
                                y--; pos--;
                                if (--c < 0) { c = C - 1; l--; }
                            }
                            len++;
                        }
                    } else {
                        const u32 tag = (w_here >> (r * 2)) & 3u;
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
            if (dt_flags & TF_DIAG_COUNT_MAT) {  // diagnostics only: result unusable
                res.n_match = (u32)mat_calls;
                res.first_a = (int)mat_ticks;
                res.first_b = (int)(wall_clock64() - walk_t0);
                res.last_a = dg_iters; res.last_b = dg_refills;
            }
        }
    }
    if (lane == 0) p_results[dt_res_idx] = res;
    __syncthreads();
}

// ---- phase A: row 0 -----------------------------------------------------------------------------------
template <int C, bool HASN, bool UNTAGGED>
__device__ __noinline__ void init_row0(BlockState<C>* st, const Tk* tp, const int lane)
{
    const Tk t = load_uniform(tp);
    const int Y = t.Y, w = t.band;
    const int LE = (Y - 1) / C;
    // row 0 (:112-132): running max without gap penalty along j
    int Lp[C];
    u32 acc[C];
    u32 W[C + 15];
    int Lin = NEG, Lout = NEG;
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
            W[k] = enc_a<HASN>((ab >> (2 * k)) & 3u, HASN && ((an >> k) & 1u));
        }
#pragma unroll
        for (int k = C - 1; k < C + 15; ++k) W[k] = 0;
    }

    {
        // everything block 0 reads: ring A entries k < 16 + 64*(C-1), ring B entries k < 16
        const int64_t A0 = t.a_base + t.begin_a - w, BB = t.b_base + t.begin_b;
        for (int k = lane; k < ROWS + (C - 1) * 64; k += 64) {
            const int64_t ia = A0 + k;
            const u32 code = (t.a2[ia >> 4] >> ((ia & 15) * 2)) & 3u;
            const bool isn = HASN && ((t.an[ia >> 5] >> (ia & 31)) & 1u);
            RingA<C>::put(k, enc_a<HASN>(code, isn));
        }
        if (lane < ROWS) {
            const int64_t ib = BB + lane;
            const u32 brow = enc_b<HASN>((t.b2[ib >> 4] >> ((ib & 15) * 2)) & 3u, HASN && ((t.bn[ib >> 5] >> (ib & 31)) & 1u)) -
                             (UNTAGGED ? RING_TAG<HASN> : 0u);
            s_ringB[lane] = brow;
            s_ringB[lane + RING_B] = brow;
        }
    }
    store_state<C>(st, Lp, acc, W, Lin, Lout);
}

// ---- the whole task -------------------------------------------------------------------------------
template <int C, int CE, bool HASN>
__device__ __forceinline__ void run_task(const DevTask& dt, const LaunchParams& p, u32* slot, const int lane)
{
    Tk t;
    t.a2 = as_global(dt.a2); t.an = as_global(dt.an); t.b2 = as_global(dt.b2); t.bn = as_global(dt.bn);
    t.a_base = dt.a_base; t.b_base = dt.b_base; t.end_a = dt.end_a;
    t.alen = dt.alen; t.blen = dt.blen; t.begin_a = dt.begin_a; t.begin_b = dt.begin_b;
    t.X = dt.X; t.band = dt.band; t.Y = 2 * dt.band + 1;
    t.fs = dt.flags & TF_FORCE_START; t.fe = dt.flags & TF_FORCE_END;
    t.dir = (gptr)slot;
    t.h0row = (giptr)(slot + p.dir_words);
    t.pos0 = t.h0row + p.ypad;
    t.lastrow = t.pos0 + p.ypad;
    t.adh = t.lastrow + p.ypad;
    t.ckpt = (gptr)(slot + p.ckpt_off);
    t.bnd = (gptr)(slot + p.bnd_off);
    t.df_lo = t.df_hi = 0;
    {
        const int64_t rel = t.end_a - t.begin_a + t.band;  // may be negative
        t.eaRel = (int)min(max(rel, (int64_t)-(1 << 30)), (int64_t)(1 << 30));
        const bool ge = t.end_a >= (int64_t)t.begin_a + t.band;
        const int64_t ia = ge ? t.end_a - ((int64_t)t.begin_a + t.band) : 0;
        t.iA = (int)min(ia, (int64_t)(1 << 30));
    }
    const int X = t.X, w = t.band;
    const int LE = (t.Y - 1) / C;  // lane holding the last band column
    BlockState<C> st;
    init_row0<C, HASN, DIRFREE_OK<CE, C, HASN>>(&st, &t, lane);

    // ---- phase B: rows 1..X-1 in blocks of 16 row-times ----------------------------------------------------
    const int nblk = (X - 1 + LE) / ROWS + 1;
    const int64_t iE0 = t.end_a - t.begin_a - w, iE1 = t.end_a - t.begin_a + w;
    auto mode_of = [&](const int blk) {
        const int tau0 = blk * ROWS;
        const bool top = !((tau0 - LE >= 1) && (t.begin_a - w + tau0 >= 1));
        const bool end = !((tau0 + ROWS - 1 < X - 1) && ((int64_t)(tau0 + ROWS - 1) < iE0 || (int64_t)(tau0 - LE) > iE1));
        return (top ? M_TOP : 0) | (end ? M_END : 0);
    };
    if constexpr (DIRFREE_OK<CE, C, HASN>) {
        // the first run of fast blocks goes direction-free, in whole groups of 4 blocks, leaving at least one tagged
        // fast block in front (what a lane receives at a group start must be its neighbour's plain last column)
        if (p.ckpt_off != 0 && !(dt.flags & TF_NO_DIRFREE)) {
            int b0 = 0;
            while (b0 < nblk && mode_of(b0) != M_FAST) ++b0;
            int b1 = b0;
            while (b1 < nblk && mode_of(b1) == M_FAST) ++b1;
            const int lo = (b0 + 1 + 3) & ~3, hi = b1 & ~3;
            if (hi - lo >= 8) { t.df_lo = lo; t.df_hi = hi; }
        }
    }
    for (int blk = 0; blk < nblk;) {
        const int m = mode_of(blk);
        if (m == M_FAST) {
            int e = blk + 1;
            while (e < nblk && mode_of(e) == M_FAST) ++e;
            if constexpr (DIRFREE_OK<CE, C, HASN>) {
                if (blk < t.df_hi && e > t.df_lo && t.df_hi > t.df_lo) {  // this is the run that holds the direction-free groups
                    if (blk < t.df_lo) fast_range<C, CE, HASN, false>(&st, &t, blk, t.df_lo, lane);
                    fast_range<C, CE, HASN, true>(&st, &t, t.df_lo, t.df_hi, lane);
                    if (t.df_hi < e) fast_range<C, CE, HASN, false>(&st, &t, t.df_hi, e, lane);
                    blk = e;
                    continue;
                }
            }
            fast_range<C, CE, HASN, false>(&st, &t, blk, e, lane);
            blk = e;
        } else {
            if (m == M_TOP) slow_block<C, CE, HASN, M_TOP>(&st, &t, blk, lane);
            else if (m == M_END) slow_block<C, CE, HASN, M_END>(&st, &t, blk, lane);
            else slow_block<C, CE, HASN, M_BOTH>(&st, &t, blk, lane);
            ++blk;
        }
    }
    finish_task<C, CE, HASN>(&t, &dt, &p, lane);
}

template <int C, int CE, bool HASN>
__global__ __launch_bounds__(64, GAMDP_WAVES_PER_SIMD) void k_align(const LaunchParams p)
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

int kernel_waves_per_cu(int) { return 4 * GAMDP_WAVES_PER_SIMD; }
int kernel_bnd_words() { return (int)BND_WORDS; }

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
