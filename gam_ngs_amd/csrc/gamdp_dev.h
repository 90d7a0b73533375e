// Structures shared between the host side of libgamdp (gamdp_host.cpp) and the gfx950 kernels
// (gamdp_kernel.hip).  Internal -- the public boundary is include/gamdp.h.
#pragma once
#include <stdint.h>

namespace gamdp {

typedef uint32_t u32;
typedef uint64_t u64;

// Sequence planes in HBM (one buffer per seqset):
//   2-bit plane: 16 bases per u32, base i at bits [2*(i&15)+1 : 2*(i&15)], codes A=0 T=1 C=2 G=3,
//                N stored as 0;
//   N plane:     32 bases per u32, bit (i&31) set when base i is N.
// Every sequence is preceded and followed by SEQ_PAD_BASES zero bases so that the kernels' window
// fetches (which run up to band+64*17 bases outside the contig) never leave the allocation.
constexpr int SEQ_PAD_BASES = 4096;

// resident waves per SIMD the kernels are register-budgeted for (4 SIMDs per CU)
#ifndef GAMDP_WAVES_PER_SIMD
#define GAMDP_WAVES_PER_SIMD 5
#endif

// one find_alignment call, pre-validated and pre-sized on the host
struct DevTask {
    const u32* a2;  // word holding base 0 of sequence a (2-bit plane)
    const u32* an;  // word holding base 0 of sequence a (N plane)
    const u32* b2;
    const u32* bn;
    int64_t a_base;  // view offset (chop_begin) in bases
    int64_t b_base;
    int64_t end_a;   // clamped to [0, 2^40]
    int32_t alen, blen;  // view lengths
    int32_t begin_a, begin_b;
    int32_t X;       // rows of the band matrix (x_size)
    int32_t band;
    u32 flags;       // TF_*
    u32 res_idx;     // slot in the result array
    u64 ops_off;     // where this task's ops go in the ops buffer (traceback order, i.e. reversed)
    u64 ops_cap;
};

enum : u32 { TF_FORCE_START = 1, TF_FORCE_END = 2, TF_WANT_OPS = 4,
             TF_DIAG_SKIP_TRACEBACK = 8 /* timing diagnostics only (GAMDP_DIAG_SKIP_TRACEBACK=1): results invalid */,
             TF_NO_DIRFREE = 16 /* GAMDP_DIAG_NO_DIRFREE=1: keep directions in every block (A/B measurements) */,
             TF_DIAG_COUNT_MAT = 32 /* GAMDP_DIAG_COUNT_MAT=1: n_match := number of materialise() calls (results invalid) */,
             TF_PADDING = 64 /* a copy of a companion that fills a wavefront of the multi-task kernels: filled along, no end cell, no walk, no result */,
             TF_CALL_N = 128 /* chain kernels only: the bases this call touches may hold an N (chain_filler tells the walker which cell it filled with) */ };

// the diagnostics flags only exist in the -DGAMDP_DIAG build: the product kernels mask them off at compile time, so a
// stray flag (or environment variable) can neither skip the traceback nor overwrite result fields with counters
#ifdef GAMDP_DIAG
constexpr u32 TF_LIVE_MASK = ~0u;
#else
constexpr u32 TF_LIVE_MASK = ~(u32)(TF_DIAG_SKIP_TRACEBACK | TF_NO_DIRFREE | TF_DIAG_COUNT_MAT);
#endif

struct DevResult {
    int32_t begin_a, begin_b;
    int32_t score;
    u32 n_match, length;
    int32_t first_a, first_b, last_a, last_b;
    u32 flags;  // bit0 first_found, bit1 last_found, bits 8.. status (GAMDP_ST_*)
};

// one kernel launch = one group of tasks that share (band, N-awareness) and a scratch slot size
struct LaunchParams {
    const DevTask* tasks;  // sorted by decreasing cell count
    u32 n_tasks;
    u32* cursor;           // work-queue head (zeroed before the launch)
    DevResult* results;
    uint8_t* ops_buf;
    u32* scratch;          // n_slots * slot_words
    u64 slot_words;        // u32 words per slot
    u64 dir_words;         // direction words at the start of a slot; side buffers follow
    u32 ypad;              // side-buffer stride in words (>= 2*band+2)
    u64 ckpt_off;          // word offsets inside a slot of the direction-free fill's row / boundary stores
    u64 bnd_off;           // (0 = this launch keeps directions everywhere)
    u32 flags;             // LP_*
    // Issue priority by remaining work (kernel_common.inc, set_prio_by_remaining): the units (tasks / pairs / quads / octets)
    // from index prio_from on raise their wavefront's s_setprio level while they have many blocks left, measured against
    // prio_R = the block count of the launch's largest task (0 = off).  The host switches it on for the units that are in
    // flight when the queue runs dry: all of a launch of at most two rounds, the last n_slots of a longer one.
    u32 prio_R, prio_from;
    // What the wavefronts did, counted on the device (LS_* below; lane 0 of a unit adds to it): the library's account of a launch
    // (gamdp_ctx_launch_info).  nullptr = not counted (the chain kernels).
    u32* stats;
};
enum : int { LS_UNITS = 0, LS_DIRFREE, LS_PACKED_TOP, LS_PACKED_TOP_MIXED, LS_STRIPS, LS_TOP_WANTED, LS_COUNT = 8 };
// the two-task kernel walks its two tasks side by side (kernel_walk.inc) instead of one after the other: set by the host for
// launches of at most two rounds, where the wavefronts of a SIMD walk at the same time and the scalar unit is the bottleneck
constexpr u32 LP_WALK_SIDE_BY_SIDE = 1;
// the packed kernels keep the int32 tagged code for their top blocks (cells with pos <= 0), as before round 4: GAMDP_NO_PACKED_TOP=1 (A/B,
// and a second way through every test)
constexpr u32 LP_NO_PACKED_TOP = 2;
// the strips of the direction-free ranges begin at multiples of the strip width, as before round 4 (Tk::sshift = 0): GAMDP_NO_STRIP_SHIFT=1
// (A/B, and a second way through the tests)
constexpr u32 LP_NO_STRIP_SHIFT = 4;
// the packed top blocks only for wavefronts whose calls share begin_a and hold no force_start call, as in round 4 (the others keep the
// int32 code): GAMDP_NO_PACKED_TOP_MIXED=1 (A/B, and a second way through the tests)
constexpr u32 LP_NO_PACKED_TOP_MIXED = 16;
// Issue priority of a wavefront of the two- / eight-task kernels while it is in its end-cell / strip / walk phase: bits 8-9.  That phase
// is latency (the walk: a memory round trip per diagonal run) and short dependent chains (a strip's tagged cells); at the level of the
// fills beside it, it waits for issue slots on top of its own latencies and holds its slot -- and the three fills' only partner that
// does something else -- for longer.  One level above a steady-state fill (the host sets 1 for launches of more than two rounds;
// the units in flight when the queue runs dry are at 1..3 by remaining work and still go first): 100 000 x 50 kb at band 150
// 170.7 -> 163.2 ms on one box, 170.0 -> 168.0 on another; band 512 359.5 -> 355.3; 200 000 x 5 kb at band 512 +4 %; level 2 and 3 gain
// less, a level of its own for the walk proper changes nothing; launches of one or two rounds lose 1 % and stay at 0.  GAMDP_WALK_PRIO=0..3
// overrides (A/B).  (The four-task int32 kernel, whose walks run one task at a time on the scalar unit, loses 3 % that way and the
// one-task kernels gain nothing: they keep level 0.)
constexpr u32 LP_WALK_PRIO_SHIFT = 8;

// Kernel variants.  C = band columns per lane; CE = (2*band) % C is the in-lane position of the
// last band column (compile-time for the tuned variants, -1 = runtime for the generic ones).
enum KernelId : int {
    K_C17_CE4 = 0,   // band 512 (benchmark workload), ACGT only
    K_C17_CE4_N,     // band 512, N-aware
    K_C5_CE0,        // band 150 (gam-merge's live default), ACGT only
    K_C5_CE0_N,      // band 150, N-aware
    K_P17_CE4,       // band 512, ACGT only, TWO tasks per wavefront, fast blocks in packed f16 (kernel_pair.inc)
    K_O19_CE15,      // band 150, ACGT only, EIGHT tasks per wavefront: two quads, fast blocks in packed f16
    K_Q19_CE15,      // band 150, ACGT only, FOUR tasks per wavefront (16 lanes x 19 columns each): big batches
    K_Q19_CE15_N,    // the same, N-aware
    K_GEN_C2, K_GEN_C3, K_GEN_C5, K_GEN_C9, K_GEN_C17,  // any band up to 543, N-aware, runtime edge column
    K_WIDE,          // any band beyond that (gamdp_wide.hip): a workgroup per task, the band matrix as int32 in the scratch slot, no throughput target
    K_COUNT
};

// ---- the pre-checks of find_alignment on plain numbers (banded_smith_waterman.cc:90-132), shared by the host (every call is
// validated and sized before it is launched) and the merge-block chain kernel (which derives the next call of a chain on
// the device).  Returns ST_OK (0) when the call has to run (then *X_out = rows of the band matrix), otherwise the final
// status the reference's behaviour maps to (1 EMPTY, 2 OUT_OF_RANGE, 3 INVALID = GAMDP_ST_*); *cells_out = x_size * y_size
// whenever the reference got as far as sizing its matrix.
#if defined(__HIPCC__)
#define GAMDP_HD __host__ __device__
#else
#define GAMDP_HD
#endif
GAMDP_HD inline int preflight_hd(u64 alen, u64 blen, u64 band, u64 begin_a, u64 end_a, u64 begin_b, u64 end_b, bool fs, bool fe,
                                 u64* X_out, u64* cells_out)
{
    constexpr int64_t MAXGAP = 10;   // FORCE_MAXGAP_LEN, banded_smith_waterman.hpp:37
    enum { S_OK = 0, S_EMPTY = 1, S_OOR = 2, S_INVALID = 3 };
    auto mn = [](int64_t x, int64_t y) { return x < y ? x : y; };
    *X_out = 0;
    *cells_out = 0;
    if (end_b < begin_b) return S_EMPTY;                              // :90
    if (begin_a >= (1ull << 31) - 65536) return S_INVALID;            // beyond any contig this code addresses
    const int64_t lo = ((int64_t)begin_a - (int64_t)band) > 0 ? ((int64_t)begin_a - (int64_t)band) : 0, hi = (int64_t)begin_a + (int64_t)band;
    if (begin_b >= blen) {
        // b.at(begin_b) / a.at(pos) throws in the row-0 loop once a column qualifies (:116-131)
        bool any;
        if (!fs) any = lo < (int64_t)alen;                            // some 0 <= pos < |a| in the band
        else any = (lo <= mn(hi, MAXGAP)) || lo < (int64_t)alen;
        return any ? S_OOR : S_INVALID;
    }
    if (end_b >= blen) end_b = blen - 1;                              // :91
    u64 X = end_b - begin_b + 1;                                      // :93-95
    const u64 lim = alen + band - begin_a;                            // unsigned wrap as in the reference
    if (lim < X) X = lim;
    if (X > 500000) X = 500000;
    if (X == 0) return S_INVALID;
    const u64 Y = 2 * band + 1;
    *X_out = X;
    *cells_out = X * Y;
    if (fs) {  // row 0 touches a.at(pos) for every 0 <= pos <= 10 in the band, even past |a| (:116)
        const int64_t up = mn(hi, MAXGAP);
        if (lo <= up && up >= (int64_t)alen) return S_OOR;
    }
    if (begin_a > alen + band) {
        // `lim` wrapped (the reference computes |a| + band - begin_a in unsigned long, :93-95): X is bounded by the b
        // window alone and EVERY cell has pos >= |a|, so the matrix keeps its zeros and the fill does nothing.  The
        // outcome follows from the end-cell scan (:174-212) alone: any eligible cell wins with value 0, lies outside a,
        // and the traceback throws from a.at(pos); no eligible cell -> MyAlignment().  Resolved here: the kernels
        // never see a window that starts past the padded contig.
        bool found = false;
        if (!fe) found = begin_a + (X - 1) - band <= end_a;                      // last row, column 0 has the smallest pos
        {
            const bool ge = end_a >= begin_a + band;
            const int64_t i0 = ge ? (int64_t)(end_a - (begin_a + band)) : 0;
            const int64_t j0 = ge ? (int64_t)(2 * band) : (int64_t)(2 * band) - (int64_t)(begin_a + band - end_a);
            if (j0 >= 0 && (u64)i0 < X) {
                const int64_t kmax = mn((int64_t)X - 1 - i0, j0);  // cells (i0 + k, j0 - k), k = 0..kmax
                if (!fe) found = true;
                else if (X >= (u64)MAXGAP + 1 && (u64)(i0 + kmax) >= X - 1 - (u64)MAXGAP) found = true;
            }
        }
        return found ? S_OOR : S_EMPTY;
    }
    return S_OK;
}

// May bases [lo, hi] of a view of a sequence (reverse complement or not, chopped by `off` bases) hold an N?  pre[k] = number of N among
// bases [0, 256 k) of the sequence in forward orientation (nullptr: none at all).  Positions outside the sequence do not count; the
// answer is by blocks of 256 bases: "yes" a little more often than the truth, never less.  Shared by the host (SeqSet::window_has_n)
// and the chain kernels, which pick the cell of every call by the window it touches.
GAMDP_HD inline bool npre_window_has_n(const u32* pre, int64_t len, bool rc, u64 off, int64_t lo, int64_t hi)
{
    if (pre == nullptr) return false;
    const int64_t o_lo = (int64_t)off + (lo > 0 ? lo : 0), o_hi = ((int64_t)off + hi < len - 1) ? (int64_t)off + hi : len - 1;
    if (o_lo > o_hi) return false;
    const int64_t f_lo = rc ? len - 1 - o_hi : o_lo, f_hi = rc ? len - 1 - o_lo : o_hi;
    return pre[(f_hi / 256) + 1] != pre[f_lo / 256];
}
// the bases a call touches (banded_smith_waterman.cc:135-171: pos = begin_a - band + x + y), `margin` more on either side
GAMDP_HD inline bool call_touches_n(const u32* pre_a, int64_t alen, bool a_rc, u64 a_off, const u32* pre_b, int64_t blen, bool b_rc, u64 b_off,
                                    int64_t band, int64_t begin_a, int64_t begin_b, int64_t X, int64_t margin)
{
    return npre_window_has_n(pre_a, alen, a_rc, a_off, begin_a - band - margin, begin_a + X - 1 + band + margin) ||
           npre_window_has_n(pre_b, blen, b_rc, b_off, begin_b - margin, begin_b + X - 1 + margin);
}

// ---- the main chain of a merge block on the device (k_chain, gamdp_kernel.hip) -----------------------------------------
// One wavefront takes a merge block through alignBlocks' serial chain (PctgBuilder.cc:1617-1708: block k starts where block
// k-1's last match ended, plus the gap between the blocks) and the orientation retry of findBestAlignment (:1420-1509)
// without returning to the host: the next window, the pre-checks, is_good(vector) (:1711-1724) are integer arithmetic.  Every
// find_alignment call leaves its result record in the audit list of its merge block, in call order; the host replays its
// own state machine over that list (gamdp_l1.cpp), so the decisions are taken twice and compared.
struct DevBlk { int32_t m_begin, m_end, s_begin, s_end; };   // in chain order (first block of the chain first)
struct DevMB {
    const u32 *a2, *an;            // master contig (the `a` sequence of every call of the chain)
    const u32 *b2, *bn;            // slave contig, forward
    const u32 *b2rc, *bnrc;        // slave contig, reverse complement
    u64 mlen, slen;
    u64 m_start, s_start, s_end;   // region (alignMergeBlock :741-744)
    u64 align_thr;
    u32 first_blk, n_blocks;       // its blocks in the DevBlk array
    u32 audit_first;               // its first record in the audit list (room for 2 * n_blocks)
    u32 rows;                      // sum of its slave frames: the rows one pass of the chain fills
    u32 try_rev;                   // orientation of the first attempt
    u32 has_n;                     // one of its two contigs holds an N: its calls MAY need the N-aware cells (12 % more instructions per row)
    // ... and which of them do is decided call by call, by the window the call touches (round 5: N by window for the chains too):
    // the N counts per 256 bases of the two contigs, forward orientation (SeqSet::npre on the device; nullptr: no N in that contig)
    const u32 *npre_a, *npre_b;
    // its scratch: slots sized for ITS longest call (x_size <= its longest slave frame), not the launch's
    u32 max_x;                     // the rows a slot has room for: a call that needs more ends the chain with state 3 (the host's round loop takes the merge block)
    u64 slot_off[2];               // word offset in ChainParams::scratch of the first slot of its workgroup / of its twin's (within the piece of the launch it is in)
    u64 slot_words, dir_words;     // words per slot; direction words at the start of a slot (side buffers follow)
    u64 ckpt_off, bnd_off;         // as in LaunchParams (0 = directions everywhere)
};
struct ChainOut { u32 n_dp; u32 state; u32 t_begin, t_end; u32 hw, hw_twin; u32 t_end_att[2]; u32 t_begin2; u32 pad; };   // t_*: the device's 100 MHz clock (low word) when the chain's workgroup started / ended (timing diagnostics)
//   // state: 0 main chain good (rev = orientation), 1 both attempts failed, 2 a call threw / was invalid, 3 a call did not fit the
//   // chain's scratch slots (nothing of the chain is used: the host takes the merge block through its round loop); bit 8: rev
// What a call of a chain was run on, next to its result record (same index): the window the device derived (PctgBuilder.cc:1652-1677),
// the orientation, the rows and the status of the pre-checks.  The host's replay derives the same call by itself and compares.
struct ChainWin { u64 begin_a, end_a, begin_b, end_b; u32 X; u32 info; };   // info: bit 0 = the slave reverse-complemented, bit 1 = the call ran the N-aware cells, bits 8.. = status of the pre-checks (0 = the DP ran)
// A long chain gets a twin: a second workgroup that runs the OTHER orientation (findBestAlignment's second attempt, :1463-1509)
// at the same time instead of after the first has failed -- the merge blocks that need it (a first guess that was wrong, a merge
// block that fails) are the ones a call waits for.  The two never wait for each other: each leaves its verdict here, and the one
// that finishes second puts the chain's record list together (attempt 0's records, then attempt 1's if attempt 0 failed --
// exactly where the one-after-the-other chain puts them) and hands it to the host.  A first attempt that settles the chain
// raises `cancel`; the twin looks at it between calls.
struct ChainSync { u32 verdict[2]; u32 n[2]; u32 fin; u32 cancel; u32 t_begin; u32 hw[2]; u32 t_end[2]; u32 t_begin2; };   // verdict: ChainOut::state of that attempt alone (1 = failed)
struct ChainParams {
    const DevMB* mbs; const DevBlk* blks; u32 n_mbs;
    u32 n_twins;                   // the launch's first n_twins workgroups are the twins of merge blocks 0 .. n_twins - 1 (only in a launch that takes all merge blocks at once)
    ChainSync* sync;               // [n_twins], zeroed before the launch
    u32 first_mb;                  // the launch takes merge blocks first_mb .. first_mb + grid - 1
    u32* cursor;
    DevResult* audit; ChainOut* out;
    ChainWin* win;                 // [2 * blocks], parallel to audit
    u32* scratch; u32 ypad; u32 band;   // (slot geometry: per merge block, DevMB)
    u32 max_rows;                  // the largest DevMB::rows of the call: chains with many rows left go first (set_prio_by_remaining)
    // the host's view while the launch runs (pinned, coherent host memory, device pointers): a chain that ends copies its
    // records and its ChainOut there and then raises its flag (done[mi] = epoch, system-scope release), so the host takes a
    // merge block on (replay, tail alignments) while longer chains are still going
    DevResult* host_audit; ChainOut* host_out; u32* host_done; u32 epoch;
    ChainWin* host_win;
    u32 skew_call;                 // diagnostics build only (GAMDP_DIAG_CHAIN_SKEW=k): the device starts call k of every first attempt one base late on the slave; ~0u = off
    int32_t n_margin;              // bases added on either side of a call's window when it is tested for N (64; the diagnostics build can shrink it below zero: GAMDP_DIAG_N_WINDOW_SHRINK, the replay must notice)
    u32 n_by_contig;               // 1: every call of a chain whose contigs hold N runs the N-aware cells (GAMDP_N_BY_CONTIG=1, GAMDP_DIAG_FORCE_N)
    u32 two_waves;                 // k_chain2: a workgroup of one filling and several walking wavefronts with chain_slots_per_workgroup() scratch slots of slot_words each
};
int launch_chain(const ChainParams& p, bool has_n, unsigned n_workgroups, void* stream);   // returns hipError_t as int
int chain_slots_per_workgroup();   // scratch slots a k_chain2 workgroup goes round (1 + its walker wavefronts)

int kernel_cols(int kid);
const char* kernel_name(int kid);      // the instantiation as rocprofv3 prints it, e.g. "k_align_o<19,15>"
bool kernel_n_aware(int kid);
bool kernel_dirfree(int kid);          // its fast blocks can run without directions (when the launch provides ckpt_off / bnd_off)
int kernel_dir_block_words(int kid);  // words per block (16 row-times) of a task's direction image
// launches on `stream`; returns hipError_t as int
int launch_align(int kid, const LaunchParams& p, unsigned n_slots, unsigned dyn_lds, void* stream);
int launch_wide(const LaunchParams& p, unsigned n_slots, void* stream);   // K_WIDE (gamdp_wide.hip): n_slots workgroups
unsigned wide_static_lds();
constexpr int WIDE_WORKGROUPS_PER_CU = 8;
unsigned kernel_static_lds(int kid);  // static LDS bytes of a variant
// occupancy hint: resident waves per CU for this variant
int kernel_waves_per_cu(int kid);
int kernel_bnd_words(int kid);  // boundary words per block of the direction-free kernels
int kernel_ckpt_words(int kid);      // words per group (4 blocks) of the live-row store of the direction-free kernels
int kernel_tasks_per_wave(int kid);

}  // namespace gamdp
