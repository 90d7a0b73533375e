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

#include <type_traits>
#include <utility>

#include "gamdp_dev.h"

namespace gamdp {
namespace {

#include "kernel_common.inc"
#include "kernel_fill.inc"
#include "kernel_pair.inc"
#include "kernel_strip.inc"
#include "kernel_finish.inc"

// Timing diagnostics of the chain kernels (when a merge block's workgroup began / ended, on which XCC / CU): the diagnostics build
// only.  The product kernels read neither the clock nor the hardware id; ChainOut's t_* / hw fields are 0 there.
#ifdef GAMDP_DIAG
__device__ __forceinline__ u32 diag_clock() { return (u32)wall_clock64(); }
__device__ __forceinline__ u32 diag_hw_id() { return ((u32)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16) | ((u32)__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xffffu); }   // XCC_ID | HW_ID
#else
__device__ __forceinline__ constexpr u32 diag_clock() { return 0u; }
__device__ __forceinline__ constexpr u32 diag_hw_id() { return 0u; }
#endif

// the device's side of gamdp_ctx_launch_info: lane 0 of a unit counts what the unit did
__device__ __forceinline__ void count_unit(const LaunchParams& p, const int lane, const bool dirfree, const bool packed_top, const bool mixed_top, const bool top_wanted = false)
{
    if (p.stats != nullptr && lane == 0) {
        atomicAdd(p.stats + LS_UNITS, 1u);
        if (dirfree) atomicAdd(p.stats + LS_DIRFREE, 1u);
        if (packed_top) atomicAdd(p.stats + LS_PACKED_TOP, 1u);
        if (mixed_top) atomicAdd(p.stats + LS_PACKED_TOP_MIXED, 1u);
        if (top_wanted) atomicAdd(p.stats + LS_TOP_WANTED, 1u);   // a unit with a packed range whose tasks hold blocks with pos <= 0 cells behind the ramp (what the packed top blocks are for)
    }
}

// block modes of one task (scalar) and its runs: b0 = first fast block, b1 = first block after that fast run
struct Plan { int nblk, b0, b1, b2; int64_t iE0, iE1; int X, LE, begin_a, w; };  // b2 = first block after the end run that follows
__device__ __forceinline__ int plan_mode(const Plan& pl, const int blk)
{
    const int tau0 = blk * ROWS;
    const bool top = !((tau0 - pl.LE >= 1) && ((int64_t)pl.begin_a - pl.w + tau0 >= 1));
    const bool end = !((tau0 + ROWS - 1 < pl.X - 1) && ((int64_t)(tau0 + ROWS - 1) < pl.iE0 || (int64_t)(tau0 - pl.LE) > pl.iE1));
    return (top ? M_TOP : 0) | (end ? M_END : 0);
}
template <int C>
__device__ __forceinline__ Plan make_plan(const Tk& t)
{
    Plan pl;
    pl.X = t.X; pl.w = t.band; pl.begin_a = t.begin_a; pl.LE = (t.Y - 1) / C;
    pl.nblk = (t.X - 1 + pl.LE) / ROWS + 1;
    pl.iE0 = t.end_a - t.begin_a - t.band; pl.iE1 = t.end_a - t.begin_a + t.band;
    // b0 / b1 / b2 in closed form (a scan over the blocks was 2 x 3 000 iterations of per-lane 64-bit compares for the eight 50 kb
    // tasks of a wavefront): blocks are top blocks below bt; not end blocks in [0, nA) (before the pos == end_a anti-diagonal
    // enters the band and before the last row) and in [sB, eB) (after it has left) -- nA <= sB, eB <= nblk.
    const int64_t nb = pl.nblk;
    const int64_t T0 = max((int64_t)pl.LE + 1, (int64_t)pl.w - pl.begin_a + 1);
    const int64_t bt = min((T0 + ROWS - 1) / ROWS, nb);
    const int64_t M = min((int64_t)pl.X - 1, pl.iE0);
    const int64_t nA = M > 0 ? min(M / ROWS, nb) : 0;
    const int64_t eB = pl.X - 1 > 0 ? (int64_t)((pl.X - 1) / ROWS) : 0;
    const int64_t S = pl.iE1 + pl.LE;
    const int64_t sB = S >= 0 ? min(S / ROWS + 1, nb) : 0;
    int64_t b0, b1, b2;
    if (bt < nA) {
        b0 = bt; b1 = nA;
        const int64_t s = max(sB, nA);
        b2 = s < eB ? s : nb;
    } else {
        const int64_t s = max(bt, sB);
        if (s < eB) { b0 = s; b1 = eB; b2 = nb; }
        else b0 = b1 = b2 = nb;
    }
    pl.b0 = (int)b0; pl.b1 = (int)b1; pl.b2 = (int)b2;
    return pl;
}

// The direction-free blocks of a GENERIC kernel (CE < 0: any band, the last band column a runtime (lane, column) pair) run through the
// instance of the tuned fast range whose compile-time edge column is this task's, (Y - 1) % C: there the missing `up` source of
// the band's last column rides in a constant (do_block_df) instead of a select per cell -- two instructions per cell instead of
// three.  C - 1 instances per kernel (the column C - 1 case keeps the runtime form); everything else of a generic kernel -- tagged
// blocks, strips, walk -- stays on its one runtime-edge instance, the data layout is the same.
template <int C, int CE, bool HASN, bool END>
__device__ __forceinline__ void df_range(BlockState<C>* st, const Tk* tp, const int from, const int to, const int lane, const int ce_rt_)
{
    if constexpr (CE >= 0) fast_range<C, CE, HASN, true, END>(st, tp, from, to, lane);
    else {
        const int ce_rt = uni(ce_rt_);
        bool done = false;
        static_for<C - 1>([&](auto k) __attribute__((always_inline)) {
            if (!done && ce_rt == decltype(k)::value) { fast_range<C, decltype(k)::value, HASN, true, END>(st, tp, from, to, lane); done = true; }
        });
        if (!done) fast_range<C, -1, HASN, true, END>(st, tp, from, to, lane);
    }
}

// ---- the whole task -------------------------------------------------------------------------------
// phases A and B: row 0 and the sweep; leaves the task's values in t for the end-cell search and the walk
template <int C, int CE, bool HASN>
__device__ __forceinline__ void fill_task(const DevTask& dt, const LaunchParams& p, u32* slot, const int lane, const bool lrpt, Tk& t)
{
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
    t.df_lo = t.df_hi = t.df_top = t.sshift = 0;
    if constexpr (DIRFREE_OK<CE, C, HASN>) t.sshift = (p.flags & LP_NO_STRIP_SHIFT) ? 0 : strip_shift<C, Strip<64>::SL>(dt.band);
    {
        const int64_t rel = t.end_a - t.begin_a + t.band;  // may be negative
        t.eaRel = (int)min(max(rel, (int64_t)-(1 << 30)), (int64_t)(1 << 30));
        const bool ge = t.end_a >= (int64_t)t.begin_a + t.band;
        const int64_t ia = ge ? t.end_a - ((int64_t)t.begin_a + t.band) : 0;
        t.iA = (int)min(ia, (int64_t)(1 << 30));
    }
    const int X = t.X, w = t.band;
    const int LE = (t.Y - 1) / C;  // lane holding the last band column
    const int nblk = (X - 1 + LE) / ROWS + 1;
    t.prio_R = lrpt ? (int)p.prio_R : 0; t.prio_nblk = nblk;
    if (t.prio_R != 0) set_prio_by_remaining(nblk, t.prio_R);
    BlockState<C> st;
    init_row0<C, HASN, DIRFREE_OK<CE, C, HASN>>(&st, &t, lane);

    // ---- phase B: rows 1..X-1 in blocks of 16 row-times ----------------------------------------------------
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
        if (p.ckpt_off != 0 && !(dt.flags & TF_LIVE_MASK & TF_NO_DIRFREE)) {
            const Plan pl = make_plan<C>(t);   // b2: the end blocks that follow the fast run go direction-free too, as far as whole groups go
            const int lo = (pl.b0 + 1 + 3) & ~3, hi = pl.b2 & ~3;
            if (hi - lo >= 8 && pl.b1 > lo) { t.df_lo = lo; t.df_hi = hi; }
        }
    }
    for (int blk = 0; blk < nblk;) {
        if (cancelled(t.cancel)) break;
        const int m = mode_of(blk);
        if (m == M_FAST) {
            int e = blk + 1;
            while (e < nblk && mode_of(e) == M_FAST) ++e;
            if constexpr (DIRFREE_OK<CE, C, HASN>) {
                if (blk < t.df_hi && e > t.df_lo && t.df_hi > t.df_lo) {  // this is the run that holds the direction-free groups
                    if (blk < t.df_lo) fast_range<C, CE, HASN, false>(&st, &t, blk, t.df_lo, lane);
                    const int f_hi = min(t.df_hi, e);  // e = first block after the fast run
                    df_range<C, CE, HASN, false>(&st, &t, t.df_lo, f_hi, lane, (t.Y - 1) % C);
                    if (f_hi < e) fast_range<C, CE, HASN, false>(&st, &t, f_hi, e, lane);
                    blk = e;
                    if (t.df_hi > e) {  // end blocks of the direction-free range
                        df_range<C, CE, HASN, true>(&st, &t, e, t.df_hi, lane, (t.Y - 1) % C);
                        blk = t.df_hi;
                    }
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
}

template <int C, int CE, bool HASN>
__device__ __forceinline__ void run_task(const DevTask& dt, const LaunchParams& p, u32* slot, const int lane, const bool lrpt)
{
    Tk t;
    t.cancel = nullptr;
    fill_task<C, CE, HASN>(dt, p, slot, lane, lrpt, t);
    count_unit(p, lane, t.df_hi > t.df_lo, false, false);
    if (t.prio_R != 0) __builtin_amdgcn_s_setprio(0);   // end cell + walk: few vector instructions, whoever still fills goes first
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
        run_task<C, CE, HASN>(p.tasks[ti], p, slot, lane, p.prio_R != 0 && ti >= p.prio_from);
    }
}

// ---- two tasks per wavefront, fast blocks in packed f16 (kernel_pair.inc) -------------------------------------------
__device__ __forceinline__ Tk make_tk(const DevTask& dt, const LaunchParams& p, u32* dir, u32* side, u32* slot)
{
    Tk t;
    t.a2 = as_global(dt.a2); t.an = as_global(dt.an); t.b2 = as_global(dt.b2); t.bn = as_global(dt.bn);
    t.a_base = dt.a_base; t.b_base = dt.b_base; t.end_a = dt.end_a;
    t.alen = dt.alen; t.blen = dt.blen; t.begin_a = dt.begin_a; t.begin_b = dt.begin_b;
    t.X = dt.X; t.band = dt.band; t.Y = 2 * dt.band + 1;
    t.fs = dt.flags & TF_FORCE_START; t.fe = dt.flags & TF_FORCE_END;
    t.dir = (gptr)dir;
    t.h0row = (giptr)side;
    t.pos0 = t.h0row + p.ypad;
    t.lastrow = t.pos0 + p.ypad;
    t.adh = t.lastrow + p.ypad;
    t.ckpt = (gptr)(slot + p.ckpt_off);
    t.bnd = (gptr)(slot + p.bnd_off);
    t.df_lo = t.df_hi = t.df_top = t.sshift = 0;
    t.prio_R = t.prio_nblk = 0;
    t.cancel = nullptr;
    const int64_t rel = t.end_a - t.begin_a + t.band;  // may be negative
    t.eaRel = (int)min(max(rel, (int64_t)-(1 << 30)), (int64_t)(1 << 30));
    const bool ge = t.end_a >= (int64_t)t.begin_a + t.band;
    const int64_t ia = ge ? t.end_a - ((int64_t)t.begin_a + t.band) : 0;
    t.iA = (int)min(ia, (int64_t)(1 << 30));
    return t;
}

// blocks [from, to) of one task with the int32 tagged code (directions for every cell)
template <int C, int CE, bool HASN>
__device__ __forceinline__ void tagged_blocks(BlockState<C>* st, const Tk* t, const Plan& pl, const int from, const int to, const int lane)
{
    for (int blk = from; blk < to;) {
        const int m = plan_mode(pl, blk);
        if (m == M_FAST) {
            int e = blk + 1;
            while (e < to && plan_mode(pl, e) == M_FAST) ++e;
            fast_range<C, CE, HASN, false>(st, t, blk, e, lane);
            blk = e;
        } else {
            if (m == M_TOP) slow_block<C, CE, HASN, M_TOP>(st, t, blk, lane);
            else if (m == M_END) slow_block<C, CE, HASN, M_END>(st, t, blk, lane);
            else slow_block<C, CE, HASN, M_BOTH>(st, t, blk, lane);
            ++blk;
        }
    }
}

// ---- four tasks per wavefront (throughput kernels of gam-merge's live band, 150) ---------------------------------
// Band 150 is 301 columns: with one task per wavefront a lane owns 5 of them and the per-row work (operand reads, the
// two hand-offs, the direction bookkeeping of the tagged cell) outweighs the cells.  Here a task takes one DPP row of 16
// lanes x 19 columns (304 >= 301), four tasks step through their blocks in lock-step, and everything the band-512
// kernels do applies: direction-free fast blocks (v_dot4 + v_max3 per cell), strips on demand, vectorised walk.  The four
// tasks of a wavefront are neighbours in the launch's longest-first order, so they finish within a block or two of each
// other; a task that is done sits masked through the slow END blocks of its companions.  One task per wavefront stays
// the better shape when a batch has fewer tasks than the chip has wave slots (latency: 5 columns per lane and row-time
// instead of 19), so the host picks per launch (gamdp_host.cpp).
__device__ __forceinline__ int quad_or(int v)
{
    return __builtin_amdgcn_readlane(v, 0) | __builtin_amdgcn_readlane(v, 16) | __builtin_amdgcn_readlane(v, 32) | __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int quad_max(int v)
{
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int quad_min(int v)
{
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
#include "kernel_walk.inc"

// end cell, walk and result of the NT tasks of a wavefront: the end cells one task at a time (the whole wavefront scans),
// the interior of all walks side by side (walk_many), then per task the steps the reference treats specially + the result
// (LPT = 64: the pair of the two-task kernel, whose Tk values are wave-uniform already; side_by_side = whether to use walk_many)
template <int C, int CE, bool HASN, bool PK, int NT, int LPT = QL>
__device__ __forceinline__ void finish_many(const LaunchParams& p, const u32 first_task, const Tk& ta, const Tk& tb, const int lane, const bool side_by_side = true)
{
    // The wave-uniform values of every task, for the out-of-line phases below (end cell, strips, walk): with one DPP row per task
    // they go to LDS ONCE per unit -- the boundary staging area, idle from here on -- instead of once per call into private
    // memory.  (The pair kernel's are wave-uniform to begin with and live in run_pair's frame.)
    Tk* const lds_tk = reinterpret_cast<Tk*>(s_qbnd);
    static_assert(LPT == 64 || (size_t)NT * sizeof(Tk) <= QB_TK_WORDS * sizeof(u32), "the tasks' values fit the staging area in front of a strip call's windows (word 512 on)");
    if constexpr (LPT != 64) {
        // (the first lane of every task row copies its own: ta / tb live in private memory -- the ranges take them by address -- and a
        // broadcast field by field, v_readlane after a reload each, was ~160 serialised round trips per unit: 160 us of a 5 kb unit's 5.5 ms)
        if ((lane & (QL - 1)) == 0) {
            Tk ts = ta;
            ts.prio_R = ts.prio_nblk = 0; ts.cancel = nullptr;
            lds_tk[lane >> 4] = ts;
            if constexpr (PK && NT > QT) {
                Tk us = tb;
                us.prio_R = us.prio_nblk = 0; us.cancel = nullptr;
                lds_tk[QT + (lane >> 4)] = us;
            }
        }
        __builtin_amdgcn_s_waitcnt(0);   // the LDS writes have landed before anybody reads them through a generic pointer
    }
    auto task_tkp = [&](const int s) -> const Tk* {
        if constexpr (LPT == 64) return s ? &tb : &ta;
        else return lds_tk + s;
    };
    // ... and so do the carries of the walks (what the end-cell search hands to the walk and a walk that stops early to the one that
    // finishes it): wave-uniform, sizeof(WalkCarry) = 100 B per task, behind the windows a strip call keeps its bases in (kernel_strip.inc; the
    // layout's offsets are kernel_common.inc's QB_*)
    WalkCarry wcs_private[LPT == 64 ? NT : 1];
    static_assert(LPT == 64 || QB_CARRY_AT * sizeof(u32) + (size_t)NT * sizeof(WalkCarry) <= sizeof(s_qbnd), "the walks' carries fit the staging area");
    WalkCarry* const wcs = (LPT == 64) ? wcs_private : reinterpret_cast<WalkCarry*>(s_qbnd + QB_CARRY_AT);
    int skip = 0, padding = 0;
#pragma unroll 1
    for (int s = 0; s < NT; ++s) {
        const u32 fl = (u32)uni((int)p.tasks[first_task + (u32)s].flags);
        if (s > 0 && (fl & TF_PADDING)) {   // (never the first task of a wavefront: its end_cell() holds the barrier and the fence the others rely on)
            WalkCarry w = {};
            w.mat_q = w.old_q = -1; w.mat_hi = w.old_hi = -1;   // status 0: nothing to walk
            wcs[s] = w;
            padding |= 1 << s;
            continue;
        }
        end_cell<C>(task_tkp(s), lane, &wcs[s]);
        if (fl & TF_WANT_OPS) skip |= 1 << s;                                                   // the edit string takes one step at a time
        if (fl & TF_LIVE_MASK & (TF_DIAG_SKIP_TRACEBACK | TF_DIAG_COUNT_MAT)) skip |= (1 << NT) - 1;  // diagnostics: the one-task walk only
    }
    // (with four int32 tasks the side-by-side walk spends as many vector instructions per task as the one-task walk, which
    // keeps its bookkeeping on the scalar unit: measured 6 % slower on 98 304 x 50 kb; eight packed tasks: 4 % faster)
    if constexpr (NT == 8 || LPT == 64) {
        if (side_by_side && skip != (1 << NT) - 1) {
            for (;;) {
                const u32 need = (u32)uni((int)walk_many<C, CE, HASN, PK, NT, LPT>(&ta, &tb, wcs, skip, lane, lds_tk));
                if (need == 0) break;
                if constexpr (LPT != 64) {
                    // the strips the walks wait for, one task at a time with the whole wavefront on it -- inlined HERE, where nothing
                    // of the walks is in registers (walk_many)
#pragma unroll 1
                    for (u32 rest = need; rest != 0; rest &= rest - 1) {
                        const int s = __builtin_ctz(rest);
                        __builtin_amdgcn_s_waitcnt(0);   // (the carries walk_many has just written)
                        const int q_s = uni(wcs[s].need_q), ghi_s = uni(wcs[s].need_ghi);
                        materialise_auto<C, CE, HASN, LPT, PK, true>(lds_tk + s, q_s, ghi_s, lane, s >> 2, QL * (s & 3));
                        // the loads of the walk must see those stores: wait until L2 has them, then drop this CU's L1 lines
                        __builtin_amdgcn_s_waitcnt(0);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        if (lane == 0) {
                            WalkCarry& w = wcs[s];
                            w.old_q = w.mat_q; w.old_lo = w.mat_lo; w.old_hi = w.mat_hi;
                            w.mat_q = q_s; w.mat_hi = 4 * ghi_s + 3; w.mat_lo = max(4 * (ghi_s - (Strip<LPT, PK>::NB - 1)), lds_tk[s].df_lo);
                            w.mat_calls++;
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0);
                }
            }
        }
    }
#pragma unroll 1
    for (int s = 0; s < NT; ++s) {
        if ((padding >> s) & 1) continue;
        const int lb = (LPT == 64) ? 0 : QL * (s & 3), hs = (LPT == 64) ? s : s >> 2;
        finish_walk<C, CE, HASN, LPT, PK>(task_tkp(s), &p.tasks[first_task + (u32)s], &p, lane, lb, hs, &wcs[s]);
    }
}

// The packed END range runs on past a task's last row: up to END_OVERRUN_BLOCKS blocks (how far the tasks of a wavefront may end
// apart) plus a group of 4, and the operand rings look (C - 1) * LPT bases + one block ahead of that.  All of it must stay inside the
// zero padding every sequence carries (SEQ_PAD_BASES, gamdp_dev.h; padded_bases() in gamdp_host.cpp pads reverse complements too).
constexpr int END_OVERRUN_BLOCKS = 64;
template <int C, int LPT_>
constexpr bool end_overrun_fits() { return (END_OVERRUN_BLOCKS + 4) * ROWS + (C - 1) * LPT_ + 512 + 64 <= SEQ_PAD_BASES; }

template <int C, int CE>
__device__ __forceinline__ void run_pair(const LaunchParams& p, const u32 qi, u32* slot, const int lane)
{
    constexpr bool HASN = false;
    const DevTask& da = p.tasks[2 * qi];
    const DevTask& db = p.tasks[2 * qi + 1];
    // slot: [dir A][dir B][side buffers A][side buffers B][packed rows][packed boundaries]
    u32* const sideA = slot + 2 * p.dir_words;
    Tk ta = make_tk(da, p, slot, sideA, slot);
    Tk tb = make_tk(db, p, slot + p.dir_words, sideA + 4u * p.ypad, slot);
    ta.sshift = tb.sshift = (p.flags & LP_NO_STRIP_SHIFT) ? 0 : strip_shift<C, Strip<64, true>::SL>(max(ta.band, tb.band));
    const Plan pa = make_plan<C>(ta), pb = make_plan<C>(tb);
    // the packed range: fast blocks of BOTH tasks, whole groups of 4 blocks, at least one tagged fast block in front of it
    // for either task (what a lane receives at a group start must be its neighbour's plain last column)
    // ... followed, still packed, by the blocks up to where the first task leaves its fast + end run
    int lo = (max(pa.b0, pb.b0) + 1 + 3) & ~3, mid = min(pa.b1, pb.b1) & ~3, hi = min(pa.b2, pb.b2) & ~3;
    if (!(p.ckpt_off != 0 && mid - lo >= 8) || ((da.flags | db.flags) & TF_LIVE_MASK & TF_NO_DIRFREE)) lo = mid = hi = 0;
    // ... and on over the last rows of both tasks, to the end of the longer one rounded up to a group (run_octo has the argument)
    static_assert(end_overrun_fits<C, 64>(), "what a task reads past its end in the packed END range stays inside the sequences' padding");
    if (hi > lo && pa.b2 == pa.nblk && pb.b2 == pb.nblk && abs(pa.nblk - pb.nblk) <= END_OVERRUN_BLOCKS) hi = (max(pa.nblk, pb.nblk) + 3) & ~3;
    // The top blocks (cells with pos <= 0: rows < band + 1 - begin_a) go packed as well: from the first group start behind the ramp
    // (every lane past its row 1) and one tagged block, pair_top_range() up to the first plain block.  Tasks that differ in begin_a cost
    // nothing extra there (round 5: the pos == -1 cell is found by a marker in the ring, not by per-lane masks); force_start calls take the
    // FS instance; a task whose band has left the triangle earlier runs plain packed blocks there.  top_mixed is kept for the launch
    // record and for GAMDP_NO_PACKED_TOP_MIXED (round 4's rule: such wavefronts through the int32 code -- a second way through the tests).
    int top_from = 0, top_to = 0;
    const bool top_mixed = ta.begin_a != tb.begin_a || ta.fs || tb.fs, top_fs = ta.fs || tb.fs;
    if (GAMDP_PACKED_TOP && hi > lo && !(p.flags & LP_NO_PACKED_TOP) && !(top_mixed && (p.flags & LP_NO_PACKED_TOP_MIXED))) {
        const int after_ramp = (pa.LE + 1 + ROWS - 1) / ROWS;   // first block with tau0 - LE >= 1
        top_from = (after_ramp + 1 + 3) & ~3;
        top_to = max(pa.b0, pb.b0);                              // (the first block without pos <= 0 cells in either task)
        // ... top blocks and nothing else: a call whose end_a lies inside the band's first rows has its pos == end_a anti-diagonal
        // (an END capture) in the same blocks
        bool only_top = top_from < top_to && top_to <= lo;
        for (int b = top_from; only_top && b < top_to; ++b) only_top = !(plan_mode(pa, b) & M_END) && !(plan_mode(pb, b) & M_END);
        if (only_top) lo = top_from;
        else top_from = top_to = 0;
    }
    ta.df_lo = tb.df_lo = lo; ta.df_hi = tb.df_hi = hi; ta.df_top = tb.df_top = top_to;
    count_unit(p, lane, hi > lo, top_to > top_from, top_to > top_from && top_mixed, hi > 0 && max(pa.b0, pb.b0) > (((pa.LE + 1 + ROWS - 1) / ROWS + 1 + 3) & ~3));
    if (p.prio_R != 0 && qi >= p.prio_from) {
        ta.prio_R = tb.prio_R = (int)p.prio_R;
        ta.prio_nblk = tb.prio_nblk = max(pa.nblk, pb.nblk);
        set_prio_by_remaining(ta.prio_nblk, ta.prio_R);
    }
    BlockState<C> sta, stb;
    init_row0<C, HASN, true>(&sta, &ta, lane);
    tagged_blocks<C, CE, HASN>(&sta, &ta, pa, 0, hi > lo ? lo : pa.nblk, lane);
    init_row0<C, HASN, true>(&stb, &tb, lane);
    tagged_blocks<C, CE, HASN>(&stb, &tb, pb, 0, hi > lo ? lo : pb.nblk, lane);
    if (hi > lo) {
        int from = lo;
        if (top_to > top_from) {
            if (!top_fs) pair_top_range<C, CE, 64, false>(&sta, &stb, &ta, &tb, top_from, top_to, lane);
            else pair_top_range<C, CE, 64, true>(&sta, &stb, &ta, &tb, top_from, top_to, lane);
            from = top_to;
        }
        pair_range<C, CE, false>(&sta, &stb, &ta, &tb, from, mid, lane);
        if (hi > mid) pair_range<C, CE, true>(&sta, &stb, &ta, &tb, mid, hi, lane);
        if (hi < pa.nblk) { single_resume<C, HASN, true>(&sta, &ta, hi, lane); tagged_blocks<C, CE, HASN>(&sta, &ta, pa, hi, pa.nblk, lane); }
        if (hi < pb.nblk) { single_resume<C, HASN, true>(&stb, &tb, hi, lane); tagged_blocks<C, CE, HASN>(&stb, &tb, pb, hi, pb.nblk, lane); }
    }
    // (two tasks that share no usable run of fast blocks were each filled with directions: they are walked as they are)
    // end cells, walks, results: one task after the other in long launches, where the scalar walk of one wavefront hides
    // behind the fills of its SIMD's other wavefronts; side by side in launches of at most two rounds (LP_WALK_SIDE_BY_SIDE)
    if (ta.prio_R != 0 || (p.flags >> LP_WALK_PRIO_SHIFT) != 0) set_prio_level((p.flags >> LP_WALK_PRIO_SHIFT) & 3u);
    finish_many<C, CE, HASN, true, 2, 64>(p, 2 * qi, ta, tb, lane, (p.flags & LP_WALK_SIDE_BY_SIDE) != 0);
}

#ifndef GAMDP_PAIR_WAVES_PER_SIMD
#define GAMDP_PAIR_WAVES_PER_SIMD 4
#endif
template <int C, int CE>
__global__ __launch_bounds__(64, GAMDP_PAIR_WAVES_PER_SIMD) void k_align_p(const LaunchParams p)
{
    const int lane = threadIdx.x;
    u32* slot = p.scratch + (u64)blockIdx.x * p.slot_words;
    for (;;) {
        u32 qi = 0;
        if (lane == 0) qi = atomicAdd(p.cursor, 1u);
        qi = __builtin_amdgcn_readfirstlane(qi);
        if (2 * qi >= p.n_tasks) break;   // n_tasks is even (the host pads the last pair)
        if ((p.flags >> LP_WALK_PRIO_SHIFT) != 0) __builtin_amdgcn_s_setprio(0);
        run_pair<C, CE>(p, qi, slot, lane);
    }
}

template <int C, int CE, bool HASN>
__device__ __forceinline__ void run_quad(const LaunchParams& p, const u32 qi, u32* slot, const int lane)
{
    constexpr int LPT = QL;
    static_assert(DIRFREE_OK<CE, C, HASN>, "the four-task kernels are direction-free kernels");
    const int sub = lane >> 4;
    const DevTask& dt = p.tasks[4 * qi + (u32)sub];  // every lane: the task of its DPP row
    Tk t;
    t.a2 = as_global(dt.a2); t.an = as_global(dt.an); t.b2 = as_global(dt.b2); t.bn = as_global(dt.bn);
    t.a_base = dt.a_base; t.b_base = dt.b_base; t.end_a = dt.end_a;
    t.alen = dt.alen; t.blen = dt.blen; t.begin_a = dt.begin_a; t.begin_b = dt.begin_b;
    t.X = dt.X; t.band = dt.band; t.Y = 2 * dt.band + 1;
    t.fs = dt.flags & TF_FORCE_START; t.fe = dt.flags & TF_FORCE_END;
    t.dir = (gptr)slot;                                   // shared: [block][column group][wavefront lane]
    t.h0row = (giptr)(slot + p.dir_words + (u64)sub * 4u * p.ypad);   // side buffers: one set per task
    t.pos0 = t.h0row + p.ypad;
    t.lastrow = t.pos0 + p.ypad;
    t.adh = t.lastrow + p.ypad;
    t.ckpt = (gptr)(slot + p.ckpt_off);
    t.bnd = (gptr)(slot + p.bnd_off);
    t.df_lo = t.df_hi = t.df_top = 0;
    t.sshift = (p.flags & LP_NO_STRIP_SHIFT) ? 0 : strip_shift<C, Strip<LPT>::SL>(uni(quad_max(dt.band)));   // one for the wavefront: its lanes share the boundary slots
    {
        const int64_t rel = t.end_a - t.begin_a + t.band;  // may be negative
        t.eaRel = (int)min(max(rel, (int64_t)-(1 << 30)), (int64_t)(1 << 30));
        const bool ge = t.end_a >= (int64_t)t.begin_a + t.band;
        const int64_t ia = ge ? t.end_a - ((int64_t)t.begin_a + t.band) : 0;
        t.iA = (int)min(ia, (int64_t)(1 << 30));
    }
    const int X = t.X, w = t.band;
    const int LE = (t.Y - 1) / C;
    const int nblk = (X - 1 + LE) / ROWS + 1;      // of this lane's task
    const int nblk_max = quad_max(nblk);
    t.prio_R = (p.prio_R != 0 && qi >= p.prio_from) ? (int)p.prio_R : 0; t.prio_nblk = nblk_max;
    t.cancel = nullptr;
    if (t.prio_R != 0) set_prio_by_remaining(nblk_max, t.prio_R);
    BlockState<C> st;
    init_row0<C, HASN, true, LPT>(&st, &t, lane);

    const int64_t iE0 = t.end_a - t.begin_a - w, iE1 = t.end_a - t.begin_a + w;
    auto mode_of = [&](const int blk) {             // per lane: the mode its own task needs for this block
        const int tau0 = blk * ROWS;
        const bool top = !((tau0 - LE >= 1) && ((int64_t)t.begin_a - w + tau0 >= 1));
        const bool end = !((tau0 + ROWS - 1 < X - 1) && ((int64_t)(tau0 + ROWS - 1) < iE0 || (int64_t)(tau0 - LE) > iE1));
        return (top ? M_TOP : 0) | (end ? M_END : 0);
    };
    // direction-free range common to the four tasks: after every task's first fast block (+1 tagged one in front, see
    // run_task), whole groups of 4 blocks, up to where the first task leaves its fast + end run
    int e_min = 0;
    if (p.ckpt_off != 0 && !(dt.flags & TF_LIVE_MASK & TF_NO_DIRFREE)) {
        const Plan pl = make_plan<C>(t);   // per lane
        const int lo = quad_max((pl.b0 + 1 + 3) & ~3), hi = quad_min(pl.b2 & ~3);
        e_min = quad_min(pl.b1);
        if (hi - lo >= 8 && e_min > lo) { t.df_lo = lo; t.df_hi = hi; }
    }
    const int df_lo = t.df_lo, df_hi = t.df_hi;  // wave-uniform by construction
    count_unit(p, lane, df_hi > df_lo, false, false);
    for (int blk = 0; blk < nblk_max;) {
        if (df_hi > df_lo && blk == df_lo) {
            const int f_hi = min(df_hi, e_min);
            if (f_hi > df_lo) fast_range<C, CE, HASN, true, false, LPT>(&st, &t, df_lo, f_hi, lane);
            if (df_hi > f_hi) fast_range<C, CE, HASN, true, true, LPT>(&st, &t, max(f_hi, df_lo), df_hi, lane);
            blk = df_hi;
            continue;
        }
        const int m = quad_or(mode_of(blk));
        if (m == M_FAST) {
            const int limit = (df_hi > df_lo && blk < df_lo) ? df_lo : nblk_max;
            int e = blk + 1;
            while (e < limit && quad_or(mode_of(e)) == M_FAST) ++e;
            fast_range<C, CE, HASN, false, false, LPT>(&st, &t, blk, e, lane);
            blk = e;
        } else {
            if (m == M_TOP) slow_block<C, CE, HASN, M_TOP, LPT>(&st, &t, blk, lane);
            else if (m == M_END) slow_block<C, CE, HASN, M_END, LPT>(&st, &t, blk, lane);
            else slow_block<C, CE, HASN, M_BOTH, LPT>(&st, &t, blk, lane);
            ++blk;
        }
    }
    if (t.prio_R != 0) __builtin_amdgcn_s_setprio(0);   // (the four-task kernel's walks, one task at a time on the scalar unit, lose 3 % one level up: LP_WALK_PRIO_SHIFT is for the packed kernels)
    finish_many<C, CE, HASN, false, QT>(p, 4 * qi, t, t, lane);
}

// ---- eight tasks per wavefront: two quads of band-150 tasks, their common fast + end blocks in packed f16 -------------
// The band-150 twin of run_pair: quad A's four tasks in the low halves, quad B's in the high halves, one task per DPP row
// in either.  The int32 phases (top blocks, what is left after the packed range) run one quad after the other with the
// tagged four-task code; the eight walks run one task at a time.
template <int C, int CE, bool HASN>
__device__ __forceinline__ void quad_tagged_blocks(BlockState<C>* st, const Tk* t, const Plan& pl, const int from, const int to, const int lane)
{
    for (int blk = from; blk < to;) {
        const int m = quad_or(plan_mode(pl, blk));
        if (m == M_FAST) {
            int e = blk + 1;
            while (e < to && quad_or(plan_mode(pl, e)) == M_FAST) ++e;
            fast_range<C, CE, HASN, false, false, QL>(st, t, blk, e, lane);
            blk = e;
        } else {
            int e = blk + 1;   // a run of top blocks (the ramp): one call
            while (m == M_TOP && e < to && quad_or(plan_mode(pl, e)) == m) ++e;
            if (m == M_TOP) slow_block<C, CE, HASN, M_TOP, QL>(st, t, blk, lane, e);
            else if (m == M_END) slow_block<C, CE, HASN, M_END, QL>(st, t, blk, lane);
            else slow_block<C, CE, HASN, M_BOTH, QL>(st, t, blk, lane);
            blk = e;
        }
    }
}

template <int C, int CE>
__device__ __forceinline__ void run_octo(const LaunchParams& p, const u32 qi, u32* slot, const int lane)
{
    constexpr bool HASN = false;
    const int sub = lane >> 4;
    const DevTask& da = p.tasks[8 * qi + (u32)sub];       // every lane: the tasks of its DPP row
    const DevTask& db = p.tasks[8 * qi + 4u + (u32)sub];
    // slot: [dir quad A][dir quad B][side buffers of the 8 tasks][packed rows][packed boundaries]
    u32* const side = slot + 2 * p.dir_words;
    Tk ta = make_tk(da, p, slot, side + (u64)sub * 4u * p.ypad, slot);
    Tk tb = make_tk(db, p, slot + p.dir_words, side + (u64)(4 + sub) * 4u * p.ypad, slot);
    ta.sshift = tb.sshift = (p.flags & LP_NO_STRIP_SHIFT) ? 0 : strip_shift<C, Strip<QL, true>::SL>(uni(quad_max(max(ta.band, tb.band))));   // one for the wavefront
    const Plan pa = make_plan<C>(ta), pb = make_plan<C>(tb);   // per lane
    const int nA = quad_max(pa.nblk), nB = quad_max(pb.nblk);
    int lo = (quad_max(max(pa.b0, pb.b0)) + 1 + 3) & ~3, mid = quad_min(min(pa.b1, pb.b1)) & ~3, hi = quad_min(min(pa.b2, pb.b2)) & ~3;
    if (!(p.ckpt_off != 0 && mid - lo >= 8) || quad_or((int)((da.flags | db.flags) & TF_LIVE_MASK & TF_NO_DIRFREE))) lo = mid = hi = 0;
    // The end blocks of the packed range run on over the last rows of EVERY task of the wavefront (round 5): rows past a task's last
    // row compute on padding and only ever feed rows past it (pair_range<END>: nothing is masked, the last row and the pos == end_a
    // cells are taken as they pass), so the range ends behind the longest task, rounded up to a group -- not at the last group
    // boundary before the shortest task's end with the rest left to the int32 code, one quad after the other (3 - 5 blocks per quad:
    // 5 % of a 5 kb task).  When every task's end run reaches its own last block and the tasks end within 64 blocks of each other:
    // what a task reads past its end then stays inside the sequences' padding (SEQ_PAD_BASES), its direction image has three blocks
    // to spare (gamdp_host.cpp).
    if (hi > lo) {
        const int nmax = max(nA, nB), nmin = quad_min(min(pa.nblk, pb.nblk));
        static_assert(end_overrun_fits<C, QL>(), "what a task reads past its end in the packed END range stays inside the sequences' padding");
        if (!quad_or((pa.b2 != pa.nblk || pb.b2 != pb.nblk) ? 1 : 0) && nmax - nmin <= END_OVERRUN_BLOCKS) hi = (nmax + 3) & ~3;
    }
    // packed top blocks (see run_pair)
    int top_from = 0, top_to = 0;
    const bool top_fs = quad_or((int)(ta.fs || tb.fs)) != 0;
    const bool top_mixed = top_fs || quad_max(max(ta.begin_a, tb.begin_a)) != quad_min(min(ta.begin_a, tb.begin_a));   // (run_pair: for the launch record and the A/B switch)
    if (GAMDP_PACKED_TOP && hi > lo && !(p.flags & LP_NO_PACKED_TOP) && !(top_mixed && (p.flags & LP_NO_PACKED_TOP_MIXED))) {
        const int after_ramp = (uni(pa.LE) + 1 + ROWS - 1) / ROWS;
        top_from = (after_ramp + 1 + 3) & ~3;
        top_to = (quad_max(max(pa.b0, pb.b0)) + 3) & ~3;   // (to a group boundary: the plain packed range re-centres by groups, PairFmt::GRP; a block or two past the triangle run here as plain blocks)
        int only_top = (top_from < top_to && top_to <= lo) ? 1 : 0;
        for (int b = top_from; only_top && b < top_to; ++b) only_top = quad_or((plan_mode(pa, b) | plan_mode(pb, b)) & M_END) ? 0 : 1;
        if (only_top) lo = top_from;
        else top_from = top_to = 0;
    }
    ta.df_lo = tb.df_lo = lo; ta.df_hi = tb.df_hi = hi; ta.df_top = tb.df_top = top_to;
    count_unit(p, lane, hi > lo, top_to > top_from, top_to > top_from && top_mixed, hi > 0 && quad_max(max(pa.b0, pb.b0)) > (((uni(pa.LE) + 1 + ROWS - 1) / ROWS + 1 + 3) & ~3));
    if (p.prio_R != 0 && qi >= p.prio_from) {
        ta.prio_R = tb.prio_R = (int)p.prio_R;
        ta.prio_nblk = tb.prio_nblk = max(nA, nB);
        set_prio_by_remaining(ta.prio_nblk, ta.prio_R);
    }
    BlockState<C> sta, stb;
    init_row0<C, HASN, true, QL>(&sta, &ta, lane);
    quad_tagged_blocks<C, CE, HASN>(&sta, &ta, pa, 0, hi > lo ? lo : nA, lane);
    init_row0<C, HASN, true, QL>(&stb, &tb, lane);
    quad_tagged_blocks<C, CE, HASN>(&stb, &tb, pb, 0, hi > lo ? lo : nB, lane);
    if (hi > lo) {
        int from = lo;
        if (top_to > top_from) {
            if (!top_fs) pair_top_range<C, CE, QL, false>(&sta, &stb, &ta, &tb, top_from, top_to, lane);
            else pair_top_range<C, CE, QL, true>(&sta, &stb, &ta, &tb, top_from, top_to, lane);
            from = top_to;
        }
        pair_range<C, CE, false, QL>(&sta, &stb, &ta, &tb, from, mid, lane);
        if (hi > mid) pair_range<C, CE, true, QL>(&sta, &stb, &ta, &tb, mid, hi, lane);
        if (hi < nA) { single_resume<C, HASN, true, QL>(&sta, &ta, hi, lane); quad_tagged_blocks<C, CE, HASN>(&sta, &ta, pa, hi, nA, lane); }
        if (hi < nB) { single_resume<C, HASN, true, QL>(&stb, &tb, hi, lane); quad_tagged_blocks<C, CE, HASN>(&stb, &tb, pb, hi, nB, lane); }
    }
    if (uni(ta.prio_R) != 0 || (p.flags >> LP_WALK_PRIO_SHIFT) != 0) set_prio_level((p.flags >> LP_WALK_PRIO_SHIFT) & 3u);
    finish_many<C, CE, HASN, true, 2 * QT>(p, 8 * qi, ta, tb, lane);
}

template <int C, int CE>
__global__ __launch_bounds__(64, GAMDP_PAIR_WAVES_PER_SIMD) void k_align_o(const LaunchParams p)
{
    const int lane = threadIdx.x;
    u32* slot = p.scratch + (u64)blockIdx.x * p.slot_words;
    for (;;) {
        u32 qi = 0;
        if (lane == 0) qi = atomicAdd(p.cursor, 1u);
        qi = __builtin_amdgcn_readfirstlane(qi);
        if (8 * qi >= p.n_tasks) break;   // n_tasks is a multiple of 8 (the host pads the last wavefront)
        if ((p.flags >> LP_WALK_PRIO_SHIFT) != 0) __builtin_amdgcn_s_setprio(0);
        run_octo<C, CE>(p, qi, slot, lane);
    }
}

template <int C, int CE, bool HASN>
__global__ __launch_bounds__(64, GAMDP_WAVES_PER_SIMD) void k_align_q(const LaunchParams p)
{
    const int lane = threadIdx.x;
    u32* slot = p.scratch + (u64)blockIdx.x * p.slot_words;
    for (;;) {
        u32 qi = 0;
        if (lane == 0) qi = atomicAdd(p.cursor, 1u);
        qi = __builtin_amdgcn_readfirstlane(qi);
        if (4 * qi >= p.n_tasks) break;   // n_tasks is a multiple of 4 (the host pads the last quad)
        run_quad<C, CE, HASN>(p, qi, slot, lane);
    }
}

// ---- the main chain of a merge block, one wavefront per merge block (structures and rationale: gamdp_dev.h) --------------
// Band 150 only (gam-merge's live band): the one-task kernel shape k_align<5, 0, HASN> run call after call by the same
// wavefront.  Everything but the DP itself is wave-uniform integer arithmetic that restates gamdp_l1.cpp's Machine for the
// MAIN phase (PctgBuilder.cc:1420-1509, 1617-1724); the host replays that machine over the audit list afterwards.
template <bool HASN>
__device__ __forceinline__ void run_chain(const ChainParams& cp, const LaunchParams& p, const u32 mi, u32* slot, const int lane)
{
    const u32 t_begin = diag_clock();
    const u32 hw_me = diag_hw_id(), hw_other = 0, te0 = 0, te1 = 0, tb2 = 0;   // XCC_ID | HW_ID (timing diagnostics)
    const DevMB* mb = unip(cp.mbs + mi);
    const u64 mlen = (u64)uni64((int64_t)mb->mlen), slen = (u64)uni64((int64_t)mb->slen);
    const u64 m_start = (u64)uni64((int64_t)mb->m_start), s_start = (u64)uni64((int64_t)mb->s_start), s_end = (u64)uni64((int64_t)mb->s_end);
    const u64 align_thr = (u64)uni64((int64_t)mb->align_thr);
    const u32 first_blk = (u32)uni((int)mb->first_blk), n = (u32)uni((int)mb->n_blocks), audit_first = (u32)uni((int)mb->audit_first);
    const u32 band = cp.band, max_x = (u32)uni((int)mb->max_x);
    bool try_rev = uni((int)mb->try_rev) != 0;
    auto frame_len = [](const int32_t b, const int32_t e) -> int32_t { return e < b ? 0 : e - b + 1; };   // Frame.cc:124-127
    u32 n_dp = 0, state = 1;
    for (int attempt = 0;; ) {
        int64_t cur_ms = (int64_t)m_start;
        int64_t cur_ss = (int64_t)(try_rev ? slen - s_end - 1 : s_start);   // reverse_complement maps (start,end) -> (|s|-end-1, |s|-start-1), :1446-1448
        u64 last_a = 0, last_b = 0, sumlen = 0;
        bool all_good = true, thrown = false, overflow = false;
        int rows_left = uni((int)mb->rows);
        for (u32 k = 0; k < n; ++k) {
            const DevBlk* bk = unip(cp.blks + first_blk + k);
            const int32_t cm_b = uni(bk->m_begin), cm_e = uni(bk->m_end), cs_b = uni(bk->s_begin), cs_e = uni(bk->s_end);
            const int32_t ml = frame_len(cm_b, cm_e), sl = frame_len(cs_b, cs_e);
            if (k > 0) {  // :1660-1667
                const int32_t pm_b = uni(bk[-1].m_begin), pm_e = uni(bk[-1].m_end), ps_b = uni(bk[-1].s_begin), ps_e = uni(bk[-1].s_end);
                const int32_t mgap = pm_b <= cm_b ? (cm_b - pm_e - 1) : (pm_b - cm_e - 1);
                const int32_t sgap = ps_b <= cs_b ? (cs_b - ps_e - 1) : (ps_b - cs_e - 1);
                cur_ms = (int64_t)(last_a + (u64)(int64_t)mgap); if (cur_ms < 0) cur_ms = 0;
                cur_ss = (int64_t)(last_b + (u64)(int64_t)sgap); if (cur_ss < 0) cur_ss = 0;
            }
#ifdef GAMDP_DIAG
            if (attempt == 0 && k == cp.skew_call) ++cur_ss;   // fault injection (GAMDP_DIAG_CHAIN_SKEW): the host's replay must notice
#endif
            const u64 begin_a = (u64)cur_ms, end_a = (u64)(cur_ms + ml - 1), begin_b = (u64)cur_ss, end_b = (u64)(cur_ss + sl - 1);
            // the call ends when its longest chain does: wavefronts that share a SIMD yield to the one with the most rows left
            set_prio_by_remaining(rows_left, (int)cp.max_rows);
            rows_left -= sl;
            u64 X = 0, cells = 0;
            const int st = preflight_hd(mlen, slen, band, begin_a, end_a, begin_b, end_b, false, false, &X, &cells);
            const u32 idx = audit_first + n_dp;
            if (st == 0 && X > (u64)max_x) { overflow = true; break; }   // (the host sized the slot for the chain's longest slave frame: never, unless its arithmetic and this one differ)
            // N by window: the cell of THIS call (HASN: one of the chain's contigs holds an N somewhere)
            bool call_n = false;
            if constexpr (HASN)
                call_n = st == 0 && (cp.n_by_contig != 0 || call_touches_n(unip(mb->npre_a), (int64_t)mlen, false, 0, unip(mb->npre_b), (int64_t)slen, try_rev, 0, (int64_t)band,
                                                                            (int64_t)begin_a, (int64_t)begin_b, (int64_t)X, (int64_t)cp.n_margin));
            if (lane == 0) { ChainWin w; w.begin_a = begin_a; w.end_a = end_a; w.begin_b = begin_b; w.end_b = end_b; w.X = (u32)X; w.info = (try_rev ? 1u : 0u) | (call_n ? 2u : 0u) | ((u32)st << 8); cp.win[idx] = w; }
#ifdef GAMDP_DIAG
            if (lane == 0) { ChainOut o; o.n_dp = n_dp; o.state = 0x1000u | ((u32)st << 16) | (k << 20); cp.out[mi] = o; }   // progress marker (overwritten at the end)
#endif
            DevResult r;
            if (st != 0) {   // settled without a DP, exactly as the host settles it (Ctx::align)
                r.begin_a = r.begin_b = r.score = 0; r.n_match = r.length = 0;
                r.first_a = r.first_b = r.last_a = r.last_b = 0;
                r.flags = (u32)st << 8;
                if (lane == 0) cp.audit[idx] = r;
            } else {
                DevTask dt;
                dt.a2 = mb->a2; dt.an = mb->an;
                dt.b2 = try_rev ? mb->b2rc : mb->b2; dt.bn = try_rev ? mb->bnrc : mb->bn;
                dt.a_base = 0; dt.b_base = 0;
                dt.end_a = (int64_t)(end_a < (1ull << 40) ? end_a : (1ull << 40));
                dt.alen = (int32_t)mlen; dt.blen = (int32_t)slen;
                dt.begin_a = (int32_t)begin_a; dt.begin_b = (int32_t)begin_b;
                dt.X = (int32_t)X; dt.band = (int32_t)band;
                dt.flags = 0; dt.res_idx = idx; dt.ops_off = 0; dt.ops_cap = 0;
                if constexpr (HASN) {
                    if (call_n) run_task<5, 0, true>(dt, p, slot, lane, false);
                    else run_task<5, 0, false>(dt, p, slot, lane, false);
                } else run_task<5, 0, false>(dt, p, slot, lane, false);
#ifdef GAMDP_DIAG
                if (lane == 0) { ChainOut o; o.n_dp = n_dp; o.state = 0x2000u | (k << 20); cp.out[mi] = o; }
#endif
                // the record lane 0 just wrote: wait until L2 has it, drop this CU's L1 lines, read it back (wave-uniform)
                __builtin_amdgcn_s_waitcnt(0);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                const DevResult* rp = cp.audit + idx;
                r.flags = (u32)uni((int)rp->flags); r.n_match = (u32)uni((int)rp->n_match); r.length = (u32)uni((int)rp->length);
                r.last_a = uni(rp->last_a); r.last_b = uni(rp->last_b);
            }
            ++n_dp;
#ifdef GAMDP_DIAG
            if (lane == 0) { ChainOut o; o.n_dp = n_dp; o.state = 0x3000u | (k << 20) | (r.flags & 0xf00u); cp.out[mi] = o; }
#endif
            const u32 status = r.flags >> 8;
            if (status == ST_OUT_OF_RANGE || status == 3u) { thrown = true; break; }   // the reference throws / undefined: the machine stops (finish_bad)
            if (status == ST_OK) {
                // homology >= 95 <=> n_match * 100 >= 95 * length (the quotient the host compares is correctly rounded and the
                // distance of n_match * 100 / length from 95 is either 0 or at least 1 / length: no rounding across 95)
                if (r.length == 0 || (u64)r.n_match * 100u < 95ull * (u64)r.length) all_good = false;
                sumlen += r.length;
                last_a = (u64)(int64_t)r.last_a; last_b = (u64)(int64_t)r.last_b;
            } else {   // EMPTY: MyAlignment(), homology 0, last match (0, 0)
                all_good = false;
                last_a = last_b = 0;
            }
        }
#ifdef GAMDP_DIAG
        if (lane == 0) { ChainOut o; o.n_dp = n_dp; o.state = 0x4000u | (all_good ? 1u : 0u) | (thrown ? 2u : 0u) | ((sumlen >= align_thr) ? 4u : 0u); cp.out[mi] = o; }
#endif
        if (overflow) { state = 3; break; }
        if (thrown) { state = 2; break; }
        if (all_good && sumlen >= align_thr) { state = try_rev ? 0x100u : 0u; break; }   // is_good(vector), :1711-1724
        if (++attempt == 2) { state = 1; break; }                                          // :1512
        try_rev = !try_rev;
    }
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0) { ChainOut o; o.n_dp = n_dp; o.state = state; o.t_begin = t_begin; o.t_end = diag_clock(); o.hw = hw_me; o.hw_twin = hw_other; o.t_end_att[0] = te0; o.t_end_att[1] = te1; o.t_begin2 = tb2; o.pad = 0; cp.out[mi] = o; }
    // hand the chain to the host: records and ChainOut into its pinned mirror, then the flag
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    {
        const u32* src = reinterpret_cast<const u32*>(cp.audit + audit_first);
        u32* dst = reinterpret_cast<u32*>(cp.host_audit + audit_first);
        const u32 nw = n_dp * (u32)(sizeof(DevResult) / sizeof(u32));
        for (u32 w = (u32)lane; w < nw; w += 64) dst[w] = src[w];
        const u32* wsrc = reinterpret_cast<const u32*>(cp.win + audit_first);
        u32* wdst = reinterpret_cast<u32*>(cp.host_win + audit_first);
        const u32 nww = n_dp * (u32)(sizeof(ChainWin) / sizeof(u32));
        for (u32 w = (u32)lane; w < nww; w += 64) wdst[w] = wsrc[w];
        if (lane == 0) { ChainOut o; o.n_dp = n_dp; o.state = state; o.t_begin = t_begin; o.t_end = diag_clock(); o.hw = hw_me; o.hw_twin = hw_other; o.t_end_att[0] = te0; o.t_end_att[1] = te1; o.t_begin2 = tb2; o.pad = 0; cp.host_out[mi] = o; }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: every lane's stores above are out before the flag
    if (lane == 0) __hip_atomic_store(cp.host_done + mi, cp.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One workgroup (= one wavefront) per merge block, no work queue: the grid is the list (longest chains first), every workgroup
// owns the scratch slot of its index.  (A persistent-wavefront version with an atomic cursor hung on the device after the last
// call of every chain -- its loop exit had been compiled lane-wise; not pursued: a merge-block call has far fewer merge blocks
// than the scratch arena has room for slots, and the host launches in pieces when it does not.)
// HASN = false: a launch in which no contig holds an N; true: every chain picks its cells by its own two contigs.
template <bool HASN>
__global__ __launch_bounds__(64, GAMDP_WAVES_PER_SIMD) void k_chain(const ChainParams cp)
{
    const int lane = threadIdx.x;
    const u32 mi = cp.first_mb + blockIdx.x;
    const DevMB* const mbp = unip(cp.mbs + mi);
    u32* slot = cp.scratch + (u64)uni64((int64_t)mbp->slot_off[0]);
    LaunchParams p;
    p.tasks = nullptr; p.n_tasks = 0; p.cursor = nullptr; p.results = cp.audit; p.ops_buf = nullptr;
    p.scratch = cp.scratch; p.slot_words = (u64)uni64((int64_t)mbp->slot_words); p.dir_words = (u64)uni64((int64_t)mbp->dir_words); p.ypad = cp.ypad;
    p.ckpt_off = (u64)uni64((int64_t)mbp->ckpt_off); p.bnd_off = (u64)uni64((int64_t)mbp->bnd_off); p.flags = 0; p.prio_R = 0; p.prio_from = 0; p.stats = nullptr;
    if constexpr (HASN) {
        if (uni((int)cp.mbs[mi].has_n) != 0) run_chain<true>(cp, p, mi, slot, lane);
        else run_chain<false>(cp, p, mi, slot, lane);
    } else run_chain<false>(cp, p, mi, slot, lane);
}

// ---- the same chain by two wavefronts: one fills, one walks ------------------------------------------------------------------
// A lone wavefront issues one instruction every ~5 cycles whatever it does, and a fifth of a chain's instructions are the end-cell
// search and the walk (measured, tools/lab_r03/r03_probe8.sh: 50 kb band-150 calls one per CU, 3.81 ms fill + 1.00 ms walk).  The next
// call of a chain needs one thing from the walk of this one: the last match (PctgBuilder.cc:1660-1667), which is the FIRST match
// the walk meets.  So the workgroup has two wavefronts and two scratch slots: wavefront 0 fills call k + 1 into one slot while
// wavefront 1 walks call k in the other (ChainMail, kernel_finish.inc).  Same calls, same records, same order in the audit list.
// role: 0 = the whole chain (both attempts, one after the other), 1 = the first attempt of a chain that has a twin, 2 = the twin
template <bool HASN>
__device__ __forceinline__ void chain_filler(const ChainParams& cp, const LaunchParams& p, const u32 mi, u32* slots, const int lane, const int role)
{
    u32 t_begin = diag_clock();
    u32 hw_me = diag_hw_id(), hw_other = 0, te0 = 0, te1 = 0, tb2 = 0;   // XCC_ID | HW_ID (timing diagnostics)
    const DevMB* mb = unip(cp.mbs + mi);
    const u64 mlen = (u64)uni64((int64_t)mb->mlen), slen = (u64)uni64((int64_t)mb->slen);
    const u64 m_start = (u64)uni64((int64_t)mb->m_start), s_start = (u64)uni64((int64_t)mb->s_start), s_end = (u64)uni64((int64_t)mb->s_end);
    const u64 align_thr = (u64)uni64((int64_t)mb->align_thr);
    const u32 first_blk = (u32)uni((int)mb->first_blk), n = (u32)uni((int)mb->n_blocks), audit_first = (u32)uni((int)mb->audit_first);
    const u32 band = cp.band, max_x = (u32)uni((int)mb->max_x);
    bool try_rev = (uni((int)mb->try_rev) != 0) != (role == 2);
    auto frame_len = [](const int32_t b, const int32_t e) -> int32_t { return e < b ? 0 : e - b + 1; };   // Frame.cc:124-127
    ChainSync* const sy = role != 0 ? cp.sync + mi : nullptr;
    if (role == 1 && lane == 0) sy->t_begin = t_begin;
    u32 n_dp = role == 2 ? n : 0u, state = 1;   // (a second attempt follows a first that made all its n calls)
    int sent = 0;   // calls handed to the walker so far (the chain's calls that need a DP)
    for (int attempt = role == 2 ? 1 : 0;; ) {
        int64_t cur_ms = (int64_t)m_start;
        int64_t cur_ss = (int64_t)(try_rev ? slen - s_end - 1 : s_start);   // reverse_complement maps (start,end) -> (|s|-end-1, |s|-start-1), :1446-1448
        u64 last_a = 0, last_b = 0;
        bool settled_bad = false, thrown = false, overflow = false;
        int rows_left = uni((int)mb->rows);
        // (the walkers are idle here: every call handed over so far is done)
        if (lane == 0) { s_mail.bad = 0; s_mail.sum = 0; }
        bool cancelled = false;
        for (u32 k = 0; k < n; ++k) {
            if (role == 2 && uni((int)__hip_atomic_load(&sy->cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) != 0) { cancelled = true; break; }   // the first attempt has settled the chain
            const DevBlk* bk = unip(cp.blks + first_blk + k);
            const int32_t cm_b = uni(bk->m_begin), cm_e = uni(bk->m_end), cs_b = uni(bk->s_begin), cs_e = uni(bk->s_end);
            const int32_t ml = frame_len(cm_b, cm_e), sl = frame_len(cs_b, cs_e);
            if (k > 0) {  // :1660-1667
                const int32_t pm_b = uni(bk[-1].m_begin), pm_e = uni(bk[-1].m_end), ps_b = uni(bk[-1].s_begin), ps_e = uni(bk[-1].s_end);
                const int32_t mgap = pm_b <= cm_b ? (cm_b - pm_e - 1) : (pm_b - cm_e - 1);
                const int32_t sgap = ps_b <= cs_b ? (cs_b - ps_e - 1) : (ps_b - cs_e - 1);
                cur_ms = (int64_t)(last_a + (u64)(int64_t)mgap); if (cur_ms < 0) cur_ms = 0;
                cur_ss = (int64_t)(last_b + (u64)(int64_t)sgap); if (cur_ss < 0) cur_ss = 0;
            }
#ifdef GAMDP_DIAG
            if (attempt == 0 && k == cp.skew_call) ++cur_ss;   // fault injection (GAMDP_DIAG_CHAIN_SKEW): the host's replay must notice
#endif
            const u64 begin_a = (u64)cur_ms, end_a = (u64)(cur_ms + ml - 1), begin_b = (u64)cur_ss, end_b = (u64)(cur_ss + sl - 1);
            set_prio_by_remaining(rows_left, (int)cp.max_rows);
            rows_left -= sl;
            u64 X = 0, cells = 0;
            const int st = preflight_hd(mlen, slen, band, begin_a, end_a, begin_b, end_b, false, false, &X, &cells);
            const u32 idx = audit_first + n_dp;
            if (st == 0 && X > (u64)max_x) { overflow = true; break; }   // (the host sized the slots for the chain's longest slave frame: never, unless its arithmetic and this one differ)
            // N by window: the cell of THIS call (HASN: one of the chain's contigs holds an N somewhere; most frames on a scaffold
            // do not touch its runs of N).  Same window as the batch path's (gamdp_host.cpp prepare_task); the host's replay derives
            // the answer once more and compares.
            bool call_n = false;
            if constexpr (HASN)
                call_n = st == 0 && (cp.n_by_contig != 0 || call_touches_n(unip(mb->npre_a), (int64_t)mlen, false, 0, unip(mb->npre_b), (int64_t)slen, try_rev, 0, (int64_t)band,
                                                                            (int64_t)begin_a, (int64_t)begin_b, (int64_t)X, (int64_t)cp.n_margin));
            if (lane == 0) { ChainWin w; w.begin_a = begin_a; w.end_a = end_a; w.begin_b = begin_b; w.end_b = end_b; w.X = (u32)X; w.info = (try_rev ? 1u : 0u) | (call_n ? 2u : 0u) | ((u32)st << 8); cp.win[idx] = w; }
            u32 status;
            if (st != 0) {   // settled without a DP, exactly as the host settles it (Ctx::align)
                DevResult r;
                r.begin_a = r.begin_b = r.score = 0; r.n_match = r.length = 0;
                r.first_a = r.first_b = r.last_a = r.last_b = 0;
                r.flags = (u32)st << 8;
                if (lane == 0) cp.audit[idx] = r;
                status = (u32)st;
                last_a = last_b = 0;
                settled_bad = true;   // (never ST_OK: an empty alignment, or one of the two that stop the machine)
            } else {
                const int par = sent % CH_NS;
                mail_wait_ge(&s_mail.done_of[par], sent - CH_NS + 1);   // the slot's last call (CH_NS calls back) has been walked
                DevTask dt;
                dt.a2 = mb->a2; dt.an = mb->an;
                dt.b2 = try_rev ? mb->b2rc : mb->b2; dt.bn = try_rev ? mb->bnrc : mb->bn;
                dt.a_base = 0; dt.b_base = 0;
                dt.end_a = (int64_t)(end_a < (1ull << 40) ? end_a : (1ull << 40));
                dt.alen = (int32_t)mlen; dt.blen = (int32_t)slen;
                dt.begin_a = (int32_t)begin_a; dt.begin_b = (int32_t)begin_b;
                dt.X = (int32_t)X; dt.band = (int32_t)band;
                dt.flags = call_n ? TF_CALL_N : 0u; dt.res_idx = idx; dt.ops_off = 0; dt.ops_cap = 0;
                Tk t;
                t.cancel = role == 2 ? &sy->cancel : nullptr;
                if constexpr (HASN) {
                    if (call_n) fill_task<5, 0, true>(dt, p, slots + (u64)par * p.slot_words, lane, false, t);
                    else fill_task<5, 0, false>(dt, p, slots + (u64)par * p.slot_words, lane, false, t);
                } else fill_task<5, 0, false>(dt, p, slots + (u64)par * p.slot_words, lane, false, t);
                if (lane == 0) { s_mail.tk[par] = t; s_mail.dt[par] = dt; }
                ++sent;
                mail_post(&s_mail.filled, sent, lane);   // (after the wavefront's stores: rows, directions, side buffers)
                mail_wait_ge(&s_mail.early_of[par], sent);
                const u32 fl = (u32)uni(s_mail.e_flags[par]);
                status = fl >> 8;
                if (status == ST_OK) { last_a = (u64)(int64_t)uni(s_mail.e_last_a[par]); last_b = (u64)(int64_t)uni(s_mail.e_last_b[par]); }
                else last_a = last_b = 0;   // EMPTY: MyAlignment(), last match (0, 0)
            }
            ++n_dp;
            if (status == ST_OUT_OF_RANGE || status == 3u) { thrown = true; break; }   // the reference throws / undefined: the machine stops (finish_bad)
        }
        mail_wait_ge(&s_mail.n_done, sent);   // every walk of this attempt has ended: is_good(vector), :1711-1724
        const bool all_good = !settled_bad && uni(s_mail.bad) == 0;
        const u64 sumlen = (u64)uni64((int64_t)s_mail.sum);
        if (cancelled) { state = 1; break; }   // (whatever: nobody looks at a cancelled twin's verdict)
        if (overflow) { state = 3; break; }
        if (thrown) { state = 2; break; }
        if (all_good && sumlen >= align_thr) { state = try_rev ? 0x100u : 0u; break; }
        if (++attempt == 2 || role == 1) { state = 1; break; }                             // :1512 (role 1: the second attempt is the twin's)
        try_rev = !try_rev;
    }
    if (lane == 0) { s_mail.total = sent; s_mail.quit = 1; }
    mail_post(&s_mail.filled, sent + CH_NW, lane);   // (wakes every walker, whichever call it waits for; nothing behind those counts)
    __builtin_amdgcn_s_waitcnt(0);
    if (role != 0) {
        // the verdict of this attempt; whoever is second puts the chain together
        const int me = role - 1;
        if (lane == 0) {
            sy->verdict[me] = state; sy->n[me] = n_dp;
            sy->hw[me] = hw_me; sy->t_end[me] = diag_clock(); if (role == 2) sy->t_begin2 = t_begin;
            if (role == 1 && state != 1u) __hip_atomic_store(&sy->cancel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0);
        u32 before = 0;
        if (lane == 0) before = __hip_atomic_fetch_add(&sy->fin, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);   // (release: this attempt's records and verdict; acquire: the other's)
        before = (u32)__builtin_amdgcn_readfirstlane((int)before);
        // a first attempt that settles the chain hands it over at once (the twin is winding down and never does); one that
        // fails leaves it to whoever is second
        const bool mine = role == 1 ? (state != 1u || before != 0u) : (before != 0u && (u32)uni((int)__hip_atomic_load(&sy->verdict[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 1u);
        if (!mine) {
            if (role == 2 && lane == 0) { cp.host_out[mi].t_begin2 = t_begin; cp.host_out[mi].t_end_att[1] = diag_clock(); }   // (timing diagnostics: a twin that wound down after the chain was handed over)
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const u32 v0 = (u32)uni((int)sy->verdict[0]), n0 = (u32)uni((int)sy->n[0]), v1 = (u32)uni((int)sy->verdict[1]), n1 = (u32)uni((int)sy->n[1]);
        if (v0 != 1u) { state = v0; n_dp = n0; }   // the first attempt settled the chain (good, or a call threw)
        else { state = v1; n_dp = n1; }             // it failed after all its n calls: the second attempt's records follow them
        t_begin = (u32)uni((int)sy->t_begin);
        hw_me = (u32)uni((int)sy->hw[0]); hw_other = (u32)uni((int)sy->hw[1]);
        te0 = (u32)uni((int)sy->t_end[0]); te1 = (u32)uni((int)sy->t_end[1]); tb2 = (u32)uni((int)sy->t_begin2);
    }
    if (lane == 0) { ChainOut o; o.n_dp = n_dp; o.state = state; o.t_begin = t_begin; o.t_end = diag_clock(); o.hw = hw_me; o.hw_twin = hw_other; o.t_end_att[0] = te0; o.t_end_att[1] = te1; o.t_begin2 = tb2; o.pad = 0; cp.out[mi] = o; }
    // hand the chain to the host: records and ChainOut into its pinned mirror, then the flag
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    {
        const u32* src = reinterpret_cast<const u32*>(cp.audit + audit_first);
        u32* dst = reinterpret_cast<u32*>(cp.host_audit + audit_first);
        const u32 nw = n_dp * (u32)(sizeof(DevResult) / sizeof(u32));
        for (u32 w = (u32)lane; w < nw; w += 64) dst[w] = src[w];
        const u32* wsrc = reinterpret_cast<const u32*>(cp.win + audit_first);
        u32* wdst = reinterpret_cast<u32*>(cp.host_win + audit_first);
        const u32 nww = n_dp * (u32)(sizeof(ChainWin) / sizeof(u32));
        for (u32 w = (u32)lane; w < nww; w += 64) wdst[w] = wsrc[w];
        if (lane == 0) { ChainOut o; o.n_dp = n_dp; o.state = state; o.t_begin = t_begin; o.t_end = diag_clock(); o.hw = hw_me; o.hw_twin = hw_other; o.t_end_att[0] = te0; o.t_end_att[1] = te1; o.t_begin2 = tb2; o.pad = 0; cp.host_out[mi] = o; }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // system scope: every lane's stores above are out before the flag
    if (lane == 0) __hip_atomic_store(cp.host_done + mi, cp.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <bool HASN>
__device__ __forceinline__ void chain_walker(const LaunchParams& p, const int lane)
{
    __builtin_amdgcn_s_setprio(3);   // few instructions, and the filler waits for the first of them
    for (;;) {
        int w = 0;
        if (lane == 0) w = __hip_atomic_fetch_add(&s_mail.claim, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        w = __builtin_amdgcn_readfirstlane(w);
        mail_wait_ge(&s_mail.filled, w + 1);
        if (uni(s_mail.quit) != 0 && w >= uni(s_mail.total)) break;   // (quit is posted with a count past every waiting walker's call)
        const int par = w % CH_NS;
        WalkCarry wc;
        end_cell<5, true>(&s_mail.tk[par], lane, &wc);
        wc.early_seq = w + 1;
        if constexpr (HASN) {   // (the walk of a call filled with the N-aware cell counts N as a match; chain_filler says which it was)
            if ((u32)uni((int)s_mail.dt[par].flags) & TF_CALL_N) finish_walk<5, 0, true, 64, false, true>(&s_mail.tk[par], &s_mail.dt[par], &p, lane, 0, 0, &wc);
            else finish_walk<5, 0, false, 64, false, true>(&s_mail.tk[par], &s_mail.dt[par], &p, lane, 0, 0, &wc);
        } else finish_walk<5, 0, false, 64, false, true>(&s_mail.tk[par], &s_mail.dt[par], &p, lane, 0, 0, &wc);
    }
}

template <bool HASN>
__global__ __launch_bounds__(64 * (1 + CH_NW), GAMDP_WAVES_PER_SIMD) void k_chain2(const ChainParams cp)
{
    const int lane = threadIdx.x & 63;
    const int wave = uni((int)(threadIdx.x >> 6));
    const bool twin = blockIdx.x < cp.n_twins;
    const u32 mi = twin ? blockIdx.x : cp.first_mb + blockIdx.x - cp.n_twins;
    const int role = twin ? 2 : (mi < cp.n_twins ? 1 : 0);
    const DevMB* const mbp = unip(cp.mbs + mi);
    u32* slots = cp.scratch + (u64)uni64((int64_t)mbp->slot_off[twin ? 1 : 0]);
    LaunchParams p;
    p.tasks = nullptr; p.n_tasks = 0; p.cursor = nullptr; p.results = cp.audit; p.ops_buf = nullptr;
    p.scratch = cp.scratch; p.slot_words = (u64)uni64((int64_t)mbp->slot_words); p.dir_words = (u64)uni64((int64_t)mbp->dir_words); p.ypad = cp.ypad;
    p.ckpt_off = (u64)uni64((int64_t)mbp->ckpt_off); p.bnd_off = (u64)uni64((int64_t)mbp->bnd_off); p.flags = 0; p.prio_R = 0; p.prio_from = 0; p.stats = nullptr;
    if (threadIdx.x == 0) {
        s_mail.filled = 0; s_mail.quit = 0; s_mail.total = 0; s_mail.claim = 0; s_mail.n_done = 0;
        for (int k = 0; k < CH_NS; ++k) { s_mail.early_of[k] = 0; s_mail.done_of[k] = 0; }
    }
    __syncthreads();
    const bool mb_n = HASN && uni((int)cp.mbs[mi].has_n) != 0;
    if (wave == 0) {
        if constexpr (HASN) {
            if (mb_n) chain_filler<true>(cp, p, mi, slots, lane, role);
            else chain_filler<false>(cp, p, mi, slots, lane, role);
        } else chain_filler<false>(cp, p, mi, slots, lane, role);
    } else {
        if constexpr (HASN) {
            if (mb_n) chain_walker<true>(p, lane);
            else chain_walker<false>(p, lane);
        } else chain_walker<false>(p, lane);
    }
}

}  // namespace

int chain_slots_per_workgroup() { return CH_NS; }

int launch_chain(const ChainParams& p, bool has_n, unsigned n_slots, void* stream)
{
    ChainParams cp = p;
    void* args[] = {&cp};
    if (cp.two_waves) {
        const void* f = has_n ? (const void*)k_chain2<true> : (const void*)k_chain2<false>;
        return (int)hipLaunchKernel(f, dim3(n_slots), dim3(64 * (1 + CH_NW)), args, 0, static_cast<hipStream_t>(stream));
    }
    const void* f = has_n ? (const void*)k_chain<true> : (const void*)k_chain<false>;
    return (int)hipLaunchKernel(f, dim3(n_slots), dim3(64), args, 0, static_cast<hipStream_t>(stream));
}

int kernel_cols(int kid)
{
    switch (kid) {
    case K_C17_CE4: case K_C17_CE4_N: case K_GEN_C17: return 17;
    case K_C5_CE0: case K_C5_CE0_N: case K_GEN_C5: return 5;
    case K_Q19_CE15: case K_Q19_CE15_N: return 19;
    case K_P17_CE4: return 17;
    case K_O19_CE15: return 19;
    case K_GEN_C2: return 2;
    case K_GEN_C3: return 3;
    case K_GEN_C9: return 9;
    default: return 0;
    }
}

const char* kernel_name(int kid)
{
    switch (kid) {
    case K_C17_CE4:    return "k_align<17,4,false>";
    case K_C17_CE4_N:  return "k_align<17,4,true>";
    case K_C5_CE0:     return "k_align<5,0,false>";
    case K_C5_CE0_N:   return "k_align<5,0,true>";
    case K_P17_CE4:    return "k_align_p<17,4>";
    case K_O19_CE15:   return "k_align_o<19,15>";
    case K_Q19_CE15:   return "k_align_q<19,15,false>";
    case K_Q19_CE15_N: return "k_align_q<19,15,true>";
    case K_GEN_C2:     return "k_align<2,-1,true>";
    case K_GEN_C3:     return "k_align<3,-1,true>";
    case K_GEN_C5:     return "k_align<5,-1,true>";
    case K_GEN_C9:     return "k_align<9,-1,true>";
    case K_GEN_C17:    return "k_align<17,-1,true>";
    case K_WIDE:       return "k_align_w";
    default:           return "?";
    }
}
bool kernel_n_aware(int kid)
{
    switch (kid) {
    case K_C17_CE4: case K_C5_CE0: case K_P17_CE4: case K_O19_CE15: case K_Q19_CE15: return false;
    default: return true;
    }
}

int kernel_waves_per_cu(int kid) { return kid == K_WIDE ? WIDE_WORKGROUPS_PER_CU : 4 * ((kid == K_P17_CE4 || kid == K_O19_CE15) ? GAMDP_PAIR_WAVES_PER_SIMD : GAMDP_WAVES_PER_SIMD); }
int kernel_tasks_per_wave(int kid) { return (kid == K_Q19_CE15 || kid == K_Q19_CE15_N) ? QT : (kid == K_P17_CE4 ? 2 : (kid == K_O19_CE15 ? 2 * QT : 1)); }
// words per block of the direction image: lane major (LANE_WORDS per lane) in the direction-free kernels
static_assert(DIRFREE_OK<4, 17, false> && DIRFREE_OK<4, 17, true> && DIRFREE_OK<15, 19, false> && DIRFREE_OK<15, 19, true> &&
                  !DIRFREE_OK<0, 5, false> && !DIRFREE_OK<0, 5, true> && DIRFREE_OK<-1, 17, true> && DIRFREE_OK<-1, 9, true> && !DIRFREE_OK<-1, 5, true> && !DIRFREE_OK<-1, 3, true>,
              "kernel_dir_block_words() below lists the direction-free kernels by id: keep it in step with DIRFREE_OK");
int kernel_dir_block_words(int kid)
{
    switch (kid) {
    case K_C17_CE4: case K_C17_CE4_N: case K_P17_CE4: return IMG_WORDS<17, true>;
    case K_Q19_CE15: case K_Q19_CE15_N: case K_O19_CE15: return IMG_WORDS<19, true>;
    case K_GEN_C17: return IMG_WORDS<17, true>;
    case K_GEN_C9: return IMG_WORDS<9, true>;
    case K_WIDE: return 1;   // (no direction image: the slot holds the band matrix itself, sized by the host)
    default: return kernel_cols(kid) * 64;
    }
}
int kernel_ckpt_words(int kid)
{
    if (kid == K_P17_CE4) return (int)PairFmt<17, 64>::CK_WORDS;
    if (kid == K_O19_CE15) return (int)PairFmt<19, QL>::CK_WORDS;
    return kernel_dir_block_words(kid);
}

bool kernel_dirfree(int kid)
{
    switch (kid) {
    case K_C17_CE4: case K_C17_CE4_N: case K_P17_CE4: case K_O19_CE15: case K_Q19_CE15: case K_Q19_CE15_N: case K_GEN_C9: case K_GEN_C17: return true;
    default: return false;
    }
}

int kernel_bnd_words(int kid)
{
    if (kid == K_P17_CE4) return (int)PairFmt<17, 64>::BND_WORDS;
    if (kid == K_O19_CE15) return (int)PairFmt<19, QL>::BND_WORDS;
    return kernel_tasks_per_wave(kid) > 1 ? (int)Strip<QL>::BND_WORDS : (int)Strip<64>::BND_WORDS;
}

namespace {
// the kernel behind a variant id, as a host-side function pointer
const void* kernel_ptr(int kid)
{
    switch (kid) {
    case K_C17_CE4:    return (const void*)k_align<17, 4, false>;
    case K_C17_CE4_N:  return (const void*)k_align<17, 4, true>;
    case K_C5_CE0:     return (const void*)k_align<5, 0, false>;
    case K_C5_CE0_N:   return (const void*)k_align<5, 0, true>;
    case K_P17_CE4:    return (const void*)k_align_p<17, 4>;
    case K_O19_CE15:   return (const void*)k_align_o<19, 15>;
    case K_Q19_CE15:   return (const void*)k_align_q<19, 15, false>;
    case K_Q19_CE15_N: return (const void*)k_align_q<19, 15, true>;
    case K_GEN_C2:     return (const void*)k_align<2, -1, true>;
    case K_GEN_C3:     return (const void*)k_align<3, -1, true>;
    case K_GEN_C5:     return (const void*)k_align<5, -1, true>;
    case K_GEN_C9:     return (const void*)k_align<9, -1, true>;
    case K_GEN_C17:    return (const void*)k_align<17, -1, true>;
    default:           return nullptr;
    }
}
}  // namespace

// static LDS bytes of a variant (0 if unknown)
unsigned kernel_static_lds(int kid)
{
    static unsigned cache[K_COUNT] = {0};
    if (kid < 0 || kid >= K_COUNT) return 0;
    if (kid == K_WIDE) return wide_static_lds();
    if (cache[kid] == 0) {
        hipFuncAttributes a;
        const void* f = kernel_ptr(kid);
        if (f && hipFuncGetAttributes(&a, f) == hipSuccess) cache[kid] = (unsigned)a.sharedSizeBytes;
    }
    return cache[kid];
}

// dyn_lds: unused dynamic LDS that only limits how many workgroups share a CU (see the launch planner in gamdp_host.cpp)
int launch_align(int kid, const LaunchParams& p, unsigned n_slots, unsigned dyn_lds, void* stream)
{
    if (kid == K_WIDE) return launch_wide(p, n_slots, stream);
    const void* f = kernel_ptr(kid);
    if (!f) return (int)hipErrorInvalidValue;
    LaunchParams lp = p;
    void* args[] = {&lp};
    return (int)hipLaunchKernel(f, dim3(n_slots), dim3(64), args, dyn_lds, static_cast<hipStream_t>(stream));
}

}  // namespace gamdp
