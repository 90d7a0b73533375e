// L1: the merge-block chain driver -- a batched, round-based re-design of
//   PctgBuilder::alignMergeBlock    lib/src/pctg/PctgBuilder.cc:726-844
//   PctgBuilder::findBestAlignment  :1361-1614
//   PctgBuilder::alignBlocks        :1617-1708
//   PctgBuilder::is_good            :1711-1730
// and ABlast::findHits (lib/src/alignment/ablast.cc:41-76).
//
// The reference walks one merge block at a time and blocks on every find_alignment call.  Inside a
// merge block the DP calls form a serial chain (block k starts where block k-1's last match ended,
// then up to one orientation retry and two tail alignments), but different merge blocks are
// independent (BuildPctgFunctions.cc:82-84).  Here every merge block is a small state machine; each
// round collects the next pending DP call of every unfinished merge block into ONE gamdp L0 batch on
// the GPU, feeds the results back and advances the machines.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "gamdp.h"
#include "gamdp_internal.h"

namespace gamdp {
static bool chain_n_by_contig() { static const bool v = std::getenv("GAMDP_N_BY_CONTIG") != nullptr; return v; }


// ---- ABlast::findHits ---------------------------------------------------------------------------
void find_hits(const uint8_t* a, u64 alen, u64 a_start, u64 a_end, const uint8_t* b, u64 blen, u64 b_start, u64 b_end,
               u64 word, std::vector<uint32_t>& hits)
{
    hits.clear();
    if (alen == 0 || blen == 0) return;                                   // ablast.cc:47
    if (a_end >= alen) a_end = alen - 1;                                  // :49-50
    if (b_end >= blen) b_end = blen - 1;
    if (a_start > a_end || b_start > b_end) return;                       // :52
    if (a_end + 1 < word + a_start || b_end + 1 < word + b_start) return; // :53
    if (word == 0) return;

    // k-mer codes are base-4 numbers with digits 0..4 (ablast.hpp:53-59), rolled along the window
    u64 top = 1;  // 4^(word-1) mod 2^64
    for (u64 i = 1; i < word; i++) top *= 4;
    auto roll = [&](const uint8_t* s, u64 first, u64 last, auto&& emit) {
        u64 code = 0;
        for (u64 i = first; i < first + word; i++) code = 4 * code + s[i];
        for (u64 p = first;; p++) {
            emit(code, p);
            if (p == last) break;
            code = 4 * (code - top * s[p]) + s[p + word];
        }
    };
    // index of the a k-mers: open-addressing table code -> chain of positions (the reference keeps a std::map of
    // position lists, ablast.hpp:61-69; only the multiset of (code, position) pairs matters for the vote)
    const u64 n_a = a_end - word + 2 - a_start;
    u64 cap = 16;
    while (cap < 2 * n_a) cap *= 2;
    static thread_local std::vector<u64> keys;
    static thread_local std::vector<u32> head, next;
    keys.assign(cap, 0);
    head.assign(cap, 0);  // 0 = empty slot, else 1 + index of the newest position in the chain
    next.assign(n_a, 0);
    const u64 mask = cap - 1;
    auto slot_of = [&](u64 code) {
        u64 h = (code * 0x9E3779B97F4A7C15ull) >> 17 & mask;
        while (head[h] != 0 && keys[h] != code) h = (h + 1) & mask;
        return h;
    };
    roll(a, a_start, a_end - word + 1, [&](u64 code, u64 p) {
        const u64 h = slot_of(code);
        const u32 me = (u32)(p - a_start);
        keys[h] = code;
        next[me] = head[h];
        head[h] = me + 1;
    });

    std::vector<u64> f(a_end - a_start + 1, 0);
    roll(b, b_start, b_end - word + 1, [&](u64 code, u64 bp) {
        const u64 ib = bp - b_start;
        for (u32 e = head[slot_of(code)]; e != 0; e = next[e - 1]) {
            const u64 ia = e - 1;
            if (ia >= ib) f[ia - ib]++;  // mark_found, ablast.hpp:71-78
        }
    });
    u64 best = 0;
    for (u64 v : f) best = std::max(best, v);
    if (best == 0) return;
    for (u64 i = 0; i < f.size(); i++)
        if (f[i] == best) hits.push_back((uint32_t)(a_start + i));
}

// ---- merge-block state machine ------------------------------------------------------------------
namespace {

constexpr double MIN_HOMOLOGY = 95.0;  // PctgBuilder.hpp:63

inline int32_t frame_len(int32_t b, int32_t e) { return e < b ? 0 : e - b + 1; }  // Frame.cc:124-127
inline u64 umin(u64 x, u64 y) { return x < y ? x : y; }

struct Machine {
    enum Phase { MAIN, LEFT, RIGHT, DONE };
    const gamdp_mb_in* in = nullptr;
    gamdp_mb_out* out = nullptr;
    const SeqSet *ms = nullptr, *ss = nullptr;
    u64 mlen = 0, slen = 0;
    u32 band = GAMDP_DEFAULT_BAND;
    Phase phase = DONE;
    // region (alignMergeBlock :741-744) and orientation evidence (findBestAlignment :1380-1408)
    u64 m_start = 0, s_start = 0, s_end = 0;
    double con_prob = 0;
    u64 mt = 0, st = 0, align_thr = 0, thr = 0;
    bool forward = true;
    // main chain
    int attempt = 0;
    bool try_rev = false;
    u32 k = 0;
    int64_t cur_ms = 0, cur_ss = 0;
    u64 last_a = 0, last_b = 0;
    std::vector<gamdp_result> A;
    bool rev = false;
    // tails
    u64 sa = 0, sb = 0, ea = 0, eb = 0, i1 = 0, i2 = 0, j1 = 0, j2 = 0;
    gamdp_result left{}, right{};
    bool left_rev = false, right_rev = false;  // the reference leaves these uninitialised when a tail is skipped
    // audit
    gamdp_result* audit = nullptr;
    u32 audit_cap = 0;
    bool on_device = false;   // its main chain is part of the chain launch of this call (launch_main_chains)

    const gamdp_block& blk(u32 i) const { return forward ? in->blocks[i] : in->blocks[in->n_blocks - 1 - i]; }

    void finish_bad(int status)
    {
        out->status = (uint8_t)status;
        out->align_ok = 0;
        phase = DONE;
    }

    void init()
    {
        out->align_ok = 1;  // :757
        out->align_rev = 0; out->status = GAMDP_ST_OK; out->coords_set = 0;
        out->m_start = out->m_end = out->s_start = out->s_end = 0;
        out->n_dp = 0; out->cells = 0;
        const u32 n = in->n_blocks;
        if (n == 0 || !in->blocks) { finish_bad(GAMDP_ST_INVALID); return; }  // front() of an empty list: UB
        const gamdp_block &fb = in->blocks[0], &lb = in->blocks[n - 1];
        m_start = (u64)(int64_t)std::min(fb.m_begin, lb.m_begin);
        s_start = (u64)(int64_t)std::min(fb.s_begin, lb.s_begin);
        s_end = (u64)(int64_t)std::max(fb.s_end, lb.s_end);
        forward = fb.m_begin <= lb.m_begin;  // :1650
        u64 con = 0, dis = 0;
        int32_t min_frame_len = 100;
        for (u32 i = 0; i < n; i++) {
            const gamdp_block& b = in->blocks[i];
            const int32_t mn = std::min(frame_len(b.m_begin, b.m_end), frame_len(b.s_begin, b.s_end));
            if (i == 0 || min_frame_len > mn) min_frame_len = mn;
            if (b.m_strand != b.s_strand) dis += (u64)b.n_reads; else con += (u64)b.n_reads;
        }
        con_prob = (double)con / (double)(con + dis);
        mt = (u64)(0.3 * (double)mlen);
        st = (u64)(0.3 * (double)slen);
        align_thr = (u64)(int64_t)(int32_t)(0.7 * min_frame_len);
        thr = (u64)(int64_t)(int32_t)umin(200, umin(mt, st));
        A.assign(n, gamdp_result{});
        attempt = 0;
        if (con_prob >= 0.5) try_rev = false;
        else if (con_prob < 0.5) try_rev = true;
        else { finish_bad(GAMDP_ST_OK); return; }  // NaN: neither branch of :1420/:1463 runs
        start_attempt();
    }

    void start_attempt()
    {
        phase = MAIN;
        k = 0;
        cur_ms = (int64_t)m_start;
        // reverse_complement maps (start,end) -> (|s|-end-1, |s|-start-1), :1446-1448
        cur_ss = (int64_t)(try_rev ? slen - s_end - 1 : s_start);
        last_a = last_b = 0;
    }

    // the next find_alignment call of this merge block
    void pending(ITask& t, std::unordered_map<u32, std::vector<uint8_t>>& rc_cache, std::mutex& rc_mu)
    {
        t = ITask{};
        t.band = band;
        if (phase == MAIN) {
            const gamdp_block& cur = blk(k);
            const int32_t ml = frame_len(cur.m_begin, cur.m_end), sl = frame_len(cur.s_begin, cur.s_end);
            if (k > 0) {  // :1660-1667
                const gamdp_block& prev = blk(k - 1);
                const int32_t mgap = prev.m_begin <= cur.m_begin ? (cur.m_begin - prev.m_end - 1) : (prev.m_begin - cur.m_end - 1);
                const int32_t sgap = prev.s_begin <= cur.s_begin ? (cur.s_begin - prev.s_end - 1) : (prev.s_begin - cur.s_end - 1);
                cur_ms = (int64_t)(last_a + (u64)(int64_t)mgap); if (cur_ms < 0) cur_ms = 0;
                cur_ss = (int64_t)(last_b + (u64)(int64_t)sgap); if (cur_ss < 0) cur_ss = 0;
            }
            t.sa = ms; t.a_id = (u32)in->m_id; t.sb = ss; t.b_id = (u32)in->s_id; t.b_rc = try_rev;
            t.begin_a = (u64)cur_ms; t.end_a = (u64)(cur_ms + ml - 1);
            t.begin_b = (u64)cur_ss; t.end_b = (u64)(cur_ss + sl - 1);
            return;
        }
        const uint8_t* mc = ms->codes[in->m_id].data();
        const uint8_t* sc;
        if (rev) {
            std::lock_guard<std::mutex> g(rc_mu);  // node-based map: the data pointer stays valid after unlock
            auto it = rc_cache.find((u32)in->s_id);
            if (it == rc_cache.end()) {
                std::vector<uint8_t> r = ss->codes[in->s_id];
                gamdp_revcomp(r.data(), r.size());
                it = rc_cache.emplace((u32)in->s_id, std::move(r)).first;
            }
            sc = it->second.data();
        } else sc = ss->codes[in->s_id].data();
        std::vector<uint32_t> hits;
        if (phase == LEFT) {  // :1535-1569, force_end
            t.force_end = true;
            if (i1 < j1) {
                find_hits(sc, slen, 0, sb - 1, mc, mlen, 0, sa - 1, 20, hits);
                t.sa = ss; t.a_id = (u32)in->s_id; t.a_rc = rev; t.sb = ms; t.b_id = (u32)in->m_id;
                t.begin_a = hits.empty() ? sb - sa : hits.back(); t.end_a = sb - 1; t.begin_b = 0; t.end_b = sa - 1;
                left_rev = true;
            } else {
                find_hits(mc, mlen, 0, sa - 1, sc, slen, 0, sb - 1, 20, hits);
                t.sa = ms; t.a_id = (u32)in->m_id; t.sb = ss; t.b_id = (u32)in->s_id; t.b_rc = rev;
                t.begin_a = hits.empty() ? sa - sb : hits.back(); t.end_a = sa - 1; t.begin_b = 0; t.end_b = sb - 1;
                left_rev = false;
            }
        } else {  // RIGHT :1573-1611, force_start; the chop_begin copy becomes a suffix view
            t.force_start = true;
            if (i2 < j2) {
                const u64 tl = slen - (eb + 1);
                find_hits(sc + eb + 1, tl, 0, tl - 1, mc, mlen, ea + 1, mlen - 1, 20, hits);
                t.sa = ss; t.a_id = (u32)in->s_id; t.a_rc = rev; t.a_off = eb + 1; t.sb = ms; t.b_id = (u32)in->m_id;
                t.begin_a = hits.empty() ? 0 : hits.front(); t.end_a = tl - 1; t.begin_b = ea + 1; t.end_b = mlen - 1;
                right_rev = true;
            } else {
                const u64 tl = mlen - (ea + 1);
                find_hits(mc + ea + 1, tl, 0, tl - 1, sc, slen, eb + 1, slen - 1, 20, hits);
                t.sa = ms; t.a_id = (u32)in->m_id; t.a_off = ea + 1; t.sb = ss; t.b_id = (u32)in->s_id; t.b_rc = rev;
                t.begin_a = hits.empty() ? 0 : hits.front(); t.end_a = tl - 1; t.begin_b = eb + 1; t.end_b = slen - 1;
                right_rev = false;
            }
        }
    }

    bool good_vec() const
    {  // is_good(vector), :1711-1724
        u64 len = 0;
        for (const gamdp_result& r : A) { if (r.homology < MIN_HOMOLOGY) return false; len += r.length; }
        return len >= align_thr;
    }
    static bool good_one(const gamdp_result& r, u64 min_len) { return r.homology >= MIN_HOMOLOGY && r.length >= min_len; }

    // LEFT phase only: will the right tail be aligned once the left one is fed?  (It depends on the main chain alone -- the
    // two tail alignments of findBestAlignment, :1535-1611, do not use each other's result -- so the round loop issues both in
    // one round; they are fed left first, as the reference computes them.)
    bool right_follows_left() const
    {
        if (!(umin(i2, j2) >= thr)) return false;
        return !((i2 < j2 && slen <= eb + 1) || (!(i2 < j2) && mlen <= ea + 1));   // (else chop_borders throws: enter_right_or_finalize)
    }

    void enter_right_or_finalize()
    {
        if (umin(i2, j2) >= thr) {
            // chop_borders throws std::domain_error when nothing is left to keep (contig.code.hpp:236-238)
            if ((i2 < j2 && slen <= eb + 1) || (!(i2 < j2) && mlen <= ea + 1)) { finish_bad(GAMDP_ST_OUT_OF_RANGE); return; }
            phase = RIGHT;
        } else finalize();
    }

    void after_main_good()
    {
        sa = A.front().first_a; sb = A.front().first_b;     // :1515-1517
        ea = A.back().last_a; eb = A.back().last_b;
        i1 = sa; i2 = mlen - ea - 1; j1 = sb; j2 = slen - eb - 1;
        std::memset(&left, 0, sizeof(left)); std::memset(&right, 0, sizeof(right));
        left.homology = right.homology = 100.0;             // MyAlignment(100)
        left_rev = right_rev = false;
        if (umin(i1, j1) < thr && umin(i2, j2) < thr) { finalize(); return; }  // :1526
        if (umin(i1, j1) >= thr) phase = LEFT;
        else enter_right_or_finalize();
    }

    void feed(const gamdp_result& r)
    {
        if (audit && out->n_dp < audit_cap) audit[out->n_dp] = r;
        out->n_dp++;
        out->cells += r.cells;
        if (r.status == GAMDP_ST_OUT_OF_RANGE || r.status == GAMDP_ST_INVALID) { finish_bad(r.status); return; }
        if (phase == MAIN) {
            A[k] = r;
            last_a = r.last_a; last_b = r.last_b;  // last_match_pos; (0,0) for an empty alignment
            if (++k < in->n_blocks) return;
            if (good_vec()) { rev = try_rev; after_main_good(); return; }
            if (++attempt == 2) { finish_bad(GAMDP_ST_OK); return; }  // :1512 -> :825-829, coords untouched
            try_rev = !try_rev;
            start_attempt();
        } else if (phase == LEFT) {
            left = r;
            enter_right_or_finalize();
        } else if (phase == RIGHT) {
            right = r;
            finalize();
        }
    }

    void finalize()
    {  // alignMergeBlock :759-843 (main_homology() >= 95 is implied by good_vec())
        const u64 thr2 = umin(100, umin(mt, st));
        const u64 left_min = (u64)(0.7 * (double)umin(i1, j1));
        const u64 right_min = (u64)(0.7 * (double)umin(i2, j2));
        const bool s_lt = rev ? in->s_rtail : in->s_ltail;
        const bool s_rt = rev ? in->s_ltail : in->s_rtail;
        u64 Sa = sa, Sb = sb, Ea = ea, Eb = eb;
        if (in->m_ltail && s_lt && umin(i1, j1) >= thr2) {
            if (good_one(left, left_min)) {
                Sa = left.first_a; Sb = left.first_b;
                if (left_rev) std::swap(Sa, Sb);
            } else out->align_ok = 0;
        }
        if (in->m_rtail && s_rt && umin(i2, j2) >= thr2) {
            if (good_one(right, right_min)) {
                u64 ta = right.last_a, tb = right.last_b;
                if (right_rev) { std::swap(ta, tb); Ea = ta; Eb += tb + 1; }
                else { Ea += ta + 1; Eb = tb; }
            } else out->align_ok = 0;
        }
        if (rev) { const u64 t = Sb; Sb = slen - Eb - 1; Eb = slen - t - 1; }  // :831-836
        out->align_rev = rev;
        out->m_start = (int32_t)Sa; out->m_end = (int32_t)Ea;
        out->s_start = (int32_t)Sb; out->s_end = (int32_t)Eb;
        out->coords_set = 1;
        phase = DONE;
    }
};

}  // namespace
}  // namespace gamdp

using namespace gamdp;

namespace gamdp {
namespace {

// ---- the main chains on the device (k_chain, gamdp_dev.h) ------------------------------------------------------------
// One launch takes every merge block through alignBlocks' chain and the orientation retry.  It runs on its own stream beside
// the round loop: a chain that ends copies its result records into a pinned mirror and raises a flag; the cohort that owns the
// merge block then replays its own machine over those records (so every decision is taken twice: a difference is an internal
// error, not a wrong answer) and sends what is left -- the tail alignments, at most two more calls -- through its next round,
// while longer chains are still running.  Band 150 (the only band gam-merge runs); anything else, and GAMDP_L1_ROUNDS=1, keeps
// the round loop for the whole call.
struct ChainRun {
    bool launched = false;
    size_t n_mb = 0;
    u32 band = 0;
    std::vector<u32> q_of;     // machine -> its index in the launch (~0u: not part of it)
    const DevMB* hmb = nullptr;
    const ChainOut* hout = nullptr;
    const DevResult* haud = nullptr;
    const ChainWin* hwin = nullptr;
    const volatile u32* done = nullptr;
    u32 epoch = 0;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const ChainOut* dout = nullptr;   // device copy of the ChainOut list (progress markers in the diagnostics build)
    bool n_by_contig = false;         // the launch ran every call of a chain with N in its contigs on the N-aware cells
    const SeqSet *ms = nullptr, *ss = nullptr;
    std::chrono::steady_clock::time_point t_launch;

    ChainRun() = default;
    ChainRun(const ChainRun&) = delete;
    ChainRun& operator=(const ChainRun&) = delete;
    ~ChainRun()
    {   // (only on a path that skipped finish(): an exception between the launch and the join)
        if (!launched) return;
        (void)hipStreamSynchronize(stream);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }

    bool ended(u32 q) const
    {
        if (done[q] != epoch) return false;
        std::atomic_thread_fence(std::memory_order_acquire);
        return true;
    }
    // waits for the launch (all paths out of the call, errors included: its buffers belong to the context)
    int finish(Ctx* c, float* from_ref_ms, float* ms)
    {
        if (!launched) return 0;
        launched = false;
        bool ok = hipStreamSynchronize(stream) == hipSuccess;
        float k = 0, f = 0;
        if (ok) ok = hipEventElapsedTime(&k, e0, e1) == hipSuccess && hipEventElapsedTime(&f, c->ref_event, e0) == hipSuccess;
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        if (!ok) { c->set_error(std::string("chain kernel: ") + hipGetErrorString(hipGetLastError())); return GAMDP_EHIP; }
        *ms = k; *from_ref_ms = f;
        c->kernel_ms += k; c->kernel_launches++;
        if (diag().timing) {
            std::fprintf(stderr, "gamdp chain: kernel %.3f ms, launched %.3f ms after the call began\n", k, f);
        }
        if (diag().timing && diag().build) {   // (the product kernels read no clock: ChainOut's t_* / hw fields are 0 there)
            // the chains that ended last (the device's 100 MHz clock, relative to the first workgroup's start)
            std::vector<u32> idx(n_mb);
            u32 t0 = hout[0].t_begin;
            for (size_t q = 0; q < n_mb; q++) { idx[q] = (u32)q; if ((int32_t)(hout[q].t_begin - t0) < 0) t0 = hout[q].t_begin; }
            std::sort(idx.begin(), idx.end(), [&](u32 a, u32 b) { return (int32_t)(hout[a].t_end - hout[b].t_end) > 0; });
            {   // the twin that ended last
                size_t qm = 0; int32_t lm = -1;
                for (size_t q = 0; q < n_mb; q++) if (hout[q].t_end_att[1] && (int32_t)(hout[q].t_end_att[1] - t0) > lm) { lm = (int32_t)(hout[q].t_end_att[1] - t0); qm = q; }
                if (lm >= 0) std::fprintf(stderr, "gamdp chain: last twin to end: #%zu's, at %.3f ms (its chain was handed over at %.3f)\n", qm, lm * 1e-5, (hout[qm].t_end - t0) * 1e-5);
            }
            for (size_t i = 0; i < std::min<size_t>(6, n_mb); i++) {
                const u32 q = idx[i];
                std::fprintf(stderr, "gamdp chain: #%u ended at %.3f ms (began %.3f): %u blocks, %u rows, %u calls, has_n %u, state 0x%x\n", q, (hout[q].t_end - t0) * 1e-5,
                             (hout[q].t_begin - t0) * 1e-5, hmb[q].n_blocks, hmb[q].rows, hout[q].n_dp, hmb[q].has_n, hout[q].state);
                auto place = [](u32 h) { char b[64]; std::snprintf(b, sizeof b, "xcc %u se %u sh %u cu %u simd %u", h >> 16, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 15, (h >> 4) & 3); return std::string(b); };
                if (hout[q].t_end_att[1]) std::fprintf(stderr, "gamdp chain:     first attempt ended at %.3f ms, the twin began at %.3f and ended at %.3f ms\n", (hout[q].t_end_att[0] - t0) * 1e-5, (hout[q].t_begin2 - t0) * 1e-5, (hout[q].t_end_att[1] - t0) * 1e-5);
                std::fprintf(stderr, "gamdp chain:     filler on %s%s%s\n", place(hout[q].hw).c_str(), hout[q].hw_twin ? "; twin's on " : "", hout[q].hw_twin ? place(hout[q].hw_twin).c_str() : "");
            }
        }
        return 0;
    }
    // diagnostics build: the progress markers of the chains
    void dump(FILE* f) const
    {
        hipStream_t s2;
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return;
        std::vector<ChainOut> o(n_mb);
        (void)hipMemcpyAsync(o.data(), dout, n_mb * sizeof(ChainOut), hipMemcpyDeviceToHost, s2);
        (void)hipStreamSynchronize(s2);
        for (size_t q = 0; q < n_mb; q++) std::fprintf(f, "gamdp chain watchdog: mb %zu n_blocks %u: n_dp %u state 0x%x flag %u\n", q, hmb[q].n_blocks, o[q].n_dp, o[q].state, (unsigned)(done[q] == epoch));
        std::fflush(f);
        (void)hipStreamDestroy(s2);
    }
};

// Starts the launch (asynchronous).  `arena` = bytes its scratch slots may take.  Returns 0 (run.launched says whether there is
// one: not for other bands, GAMDP_L1_ROUNDS=1, no merge block with a chain, or a frame too long for the arena) or an error code.
int launch_main_chains(Ctx* c, std::vector<Machine>& M, const SeqSet* ms, const SeqSet* ss, const u32 band, const u64 arena, ChainRun& run)
{
    run.launched = false;
    static const bool rounds_only = std::getenv("GAMDP_L1_ROUNDS") != nullptr;
    if (rounds_only || band != 150) return 0;
    // A merge block with an empty slave frame (s_end < s_begin) stays with the round loop: the call of such a block that
    // starts at slave base 0 has end_b = begin_b - 1 wrapped around (the reference computes it in unsigned long,
    // PctgBuilder.cc:1669-1677), so its rows are bounded by the contig, not by the frame the scratch slots below are sized for.
    std::vector<u32> act;
    for (u32 i = 0; i < (u32)M.size(); i++) {
        if (M[i].phase != Machine::MAIN) continue;
        bool empty_frame = false;
        for (u32 k = 0; k < M[i].in->n_blocks; k++) empty_frame = empty_frame || M[i].in->blocks[k].s_end < M[i].in->blocks[k].s_begin;
        if (!empty_frame) act.push_back(i);
    }
    if (act.empty()) return 0;
    // longest chains first
    std::vector<u64> w(act.size(), 0);
    std::vector<u32> longest(act.size(), 1);   // its longest slave frame: no call of the chain has more rows
    u64 n_blk = 0;
    bool has_n = false;
    std::vector<u32> need_rc;
    for (size_t q = 0; q < act.size(); q++) {
        const gamdp_mb_in& in = *M[act[q]].in;
        for (u32 k = 0; k < in.n_blocks; k++) {
            const int32_t sl = frame_len(in.blocks[k].s_begin, in.blocks[k].s_end);
            w[q] += (u64)sl;
            longest[q] = std::max<u32>(longest[q], (u32)sl);
        }
        n_blk += in.n_blocks;
        has_n = has_n || ms->has_n[in.m_id] || ss->has_n[in.s_id];
        need_rc.push_back((u32)in.s_id);
    }
    { int rc_ = ss->ensure_rc(need_rc, c); if (rc_) return rc_; }
    std::vector<u32> order(act.size());
    for (u32 q = 0; q < order.size(); q++) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return w[x] > w[y]; });

    // device: DevMB[] | DevBlk[] | ChainOut[] | DevResult audit[]      (the first two uploaded from h_chain)
    // mirror: ChainOut[] | done flags | DevResult audit[]               (pinned, coherent; written by the chains as they end)
    const u64 n_mb = act.size(), n_audit = 2 * n_blk;
    auto up = [](u64 v) { return (v + 255) & ~255ull; };
    // the longest chains get a twin workgroup for their second orientation (ChainSync, gamdp_dev.h): those within 1/8 of the
    // longest, at most 256 of them; GAMDP_L1_NO_TWINS=1: none
    static const bool no_twins = std::getenv("GAMDP_L1_NO_TWINS") != nullptr;
    u64 n_tw = 0;
    const u64 tw_cap = n_mb < 256 ? std::max<u64>(16, 256 - n_mb) : 256;   // (a small call: one workgroup per CU as long as that leaves room for a few)
    if (!no_twins)
        while (n_tw < n_mb && n_tw < tw_cap && w[order[n_tw]] * 8 >= w[order[0]] && w[order[n_tw]] >= 1024) n_tw++;
    const u64 off_mb = 0, off_blk = up(off_mb + n_mb * sizeof(DevMB)), off_sync = up(off_blk + n_blk * sizeof(DevBlk)), off_out = up(off_sync + (n_tw + 1) * sizeof(ChainSync)),
              off_aud = up(off_out + n_mb * sizeof(ChainOut)), off_win = up(off_aud + n_audit * sizeof(DevResult)), total = up(off_win + n_audit * sizeof(ChainWin));
    const u64 mo_out = 0, mo_done = up(mo_out + n_mb * sizeof(ChainOut)), mo_aud = up(mo_done + n_mb * sizeof(u32)), mo_win = up(mo_aud + n_audit * sizeof(DevResult)),
              mtotal = up(mo_win + n_audit * sizeof(ChainWin));
    if (total > c->cap_chain) {
        if (c->d_chain) (void)hipFree(c->d_chain);
        c->d_chain = nullptr; c->cap_chain = 0;
        if (hipMalloc(&c->d_chain, total + total / 4) != hipSuccess) { c->set_error("hipMalloc of the chain buffers failed"); return GAMDP_ENOMEM; }
        c->cap_chain = total + total / 4;
    }
    if (off_out > c->cap_hchain) {
        if (c->h_chain) (void)hipHostFree(c->h_chain);
        c->h_chain = nullptr; c->cap_hchain = 0;
        if (hipHostMalloc(&c->h_chain, off_out + off_out / 4) != hipSuccess) { c->set_error("hipHostMalloc of the chain buffers failed"); return GAMDP_ENOMEM; }
        c->cap_hchain = off_out + off_out / 4;
    }
    if (mtotal > c->cap_mirror) {
        if (c->h_mirror) (void)hipHostFree(c->h_mirror);
        c->h_mirror = nullptr; c->cap_mirror = 0;
        const u64 want = mtotal + mtotal / 4;
        if (hipHostMalloc(&c->h_mirror, want, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { c->set_error("hipHostMalloc of the chain mirror failed"); return GAMDP_ENOMEM; }
        c->cap_mirror = want;
    }
    void* d_mirror = nullptr;
    if (hipHostGetDevicePointer(&d_mirror, c->h_mirror, 0) != hipSuccess) { c->set_error("hipHostGetDevicePointer failed"); return GAMDP_EHIP; }
    if (!c->chain_stream) {
        // A stream of its own priority class: the runtime multiplexes the streams of one class over a few hardware queues, and a
        // round loop whose stream shares the chain launch's queue waits for the whole launch (measured: two of ten cohorts sat
        // 27 ms behind it).  The lowest class: the launch is resident at once, the round loops' small launches go ahead of nothing.
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = 0; (void)hipGetLastError(); }
        if (hipStreamCreateWithPriority(&c->chain_stream, hipStreamNonBlocking, least) != hipSuccess) { c->set_error("hipStreamCreate failed"); return GAMDP_EHIP; }
    }
    // the flags: cleared for every launch (the layout moves with the call's sizes, so what lies there may be an earlier call's
    // records), and raised to a value that changes from launch to launch
    std::memset((uint8_t*)c->h_mirror + mo_done, 0, n_mb * sizeof(u32));
    if (++c->chain_epoch == 0) c->chain_epoch = 1;
    uint8_t* const h = (uint8_t*)c->h_chain;
    uint8_t* const d = (uint8_t*)c->d_chain;
    uint8_t* const hm = (uint8_t*)c->h_mirror;
    uint8_t* const dm = (uint8_t*)d_mirror;
    DevMB* hmb = (DevMB*)(h + off_mb);
    DevBlk* hbk = (DevBlk*)(h + off_blk);
    run.q_of.assign(M.size(), ~0u);
    u32 blk_at = 0;
    for (size_t q = 0; q < n_mb; q++) {
        const Machine& m = M[act[order[q]]];
        const gamdp_mb_in& in = *m.in;
        run.q_of[act[order[q]]] = (u32)q;
        DevMB& x = hmb[q];
        x.a2 = ms->fwd[in.m_id].p2; x.an = ms->fwd[in.m_id].pn;
        x.b2 = ss->fwd[in.s_id].p2; x.bn = ss->fwd[in.s_id].pn;
        x.b2rc = ss->rc[in.s_id].p2; x.bnrc = ss->rc[in.s_id].pn;
        x.mlen = m.mlen; x.slen = m.slen;
        x.m_start = m.m_start; x.s_start = m.s_start; x.s_end = m.s_end;
        x.align_thr = m.align_thr;
        x.first_blk = blk_at; x.n_blocks = in.n_blocks; x.audit_first = 2 * blk_at;
        x.rows = (u32)std::min<u64>(w[order[q]], 0x7fffffffu);
        x.try_rev = m.try_rev ? 1u : 0u;
        x.has_n = (ms->has_n[in.m_id] || ss->has_n[in.s_id] || diag().force_n) ? 1u : 0u;
        x.npre_a = (size_t)in.m_id < ms->dev_npre.size() ? ms->dev_npre[in.m_id] : nullptr;
        x.npre_b = (size_t)in.s_id < ss->dev_npre.size() ? ss->dev_npre[in.s_id] : nullptr;
        for (u32 k = 0; k < in.n_blocks; k++) {
            const gamdp_block& b = m.blk(k);
            hbk[blk_at + k] = DevBlk{b.m_begin, b.m_end, b.s_begin, b.s_end};
        }
        blk_at += in.n_blocks;
    }
    // Scratch: every workgroup goes round its own slots (chain_slots_per_workgroup(); one for the one-wavefront kernel), sized
    // for the longest call ITS chain can make (x_size <= its longest slave frame) -- a call of a 30 Mb genome has a few chains
    // with frames of 100 kb and two thousand with frames of a few kb.  What does not fit the arena at once goes in pieces, one
    // launch after the other over the same memory; twins only when everything fits at once.
    const u64 Y = 2ull * band + 1, LE = (Y - 1) / 5;
    const u32 ypad = (u32)(((2 * band + 2 + 63) / 64) * 64);
    const bool df = kernel_dirfree(K_C5_CE0_N);   // (the chain kernels' 5-column shape keeps a direction per cell: no checkpoint / boundary stores)
    // k_chain2: a filling and two walking wavefronts per merge block; GAMDP_L1_ONE_WAVE=1 keeps the one-wavefront kernel (A/B)
    static const bool one_wave = std::getenv("GAMDP_L1_ONE_WAVE") != nullptr;
    const u64 per_wg = one_wave ? 1 : (u64)chain_slots_per_workgroup();
    const u64 arena_words = arena / sizeof(u32);
    u64 words_all = 0, words_twins = 0, slotw_max = 0;
    for (size_t q = 0; q < n_mb; q++) {
        DevMB& x = hmb[q];
        const u64 nblk = ((u64)longest[order[q]] - 1 + LE) / 16 + 1;
        const u64 dirw = ((nblk * (u64)kernel_dir_block_words(K_C5_CE0_N) + 63) / 64) * 64;
        const u64 ckptw = df ? (nblk / 4 + 2) * (u64)kernel_ckpt_words(K_C5_CE0_N) : 0, bndw = df ? (nblk + 4) * (u64)kernel_bnd_words(K_C5_CE0_N) : 0;
        x.max_x = longest[order[q]];
        x.dir_words = dirw; x.slot_words = dirw + 4ull * ypad + ckptw + bndw;
        x.ckpt_off = df ? dirw + 4ull * ypad : 0; x.bnd_off = x.ckpt_off + ckptw;
        slotw_max = std::max(slotw_max, x.slot_words);
        words_all += per_wg * x.slot_words;
        if (q < n_tw) words_twins += per_wg * x.slot_words;
    }
    if (per_wg * slotw_max > arena_words) return 0;   // (a frame too long for the arena: the round loop peels such calls off by itself)
    if (one_wave || words_all + words_twins > arena_words) { n_tw = 0; words_twins = 0; }
    // pieces: [first, first + count) of the list, each within the arena; slot offsets are relative to the piece
    std::vector<std::pair<u32, u32>> pieces;
    u64 need_scratch = 0;
    {
        u64 at = 0;
        for (size_t q = 0; q < n_tw; q++) { hmb[q].slot_off[1] = at; at += per_wg * hmb[q].slot_words; }   // (the twins' workgroups come first in the grid)
        u32 first = 0;
        for (size_t q = 0; q < n_mb; q++) {
            const u64 mine = per_wg * hmb[q].slot_words;
            if (at + mine > arena_words) {   // (never with twins: then everything fits)
                pieces.emplace_back(first, (u32)q - first);
                need_scratch = std::max(need_scratch, at);
                first = (u32)q; at = 0;
            }
            hmb[q].slot_off[0] = at;
            if (q >= n_tw) hmb[q].slot_off[1] = 0;
            at += mine;
        }
        pieces.emplace_back(first, (u32)n_mb - first);
        need_scratch = std::max(need_scratch, at);
    }
    if (need_scratch > c->cap_chain_scratch) {
        if (c->d_chain_scratch) { (void)hipFree(c->d_chain_scratch); c->d_chain_scratch = nullptr; c->cap_chain_scratch = 0; }
        if (hipMalloc(&c->d_chain_scratch, need_scratch * sizeof(u32)) != hipSuccess) { c->d_chain_scratch = nullptr; c->set_error("hipMalloc of the chains' scratch slots failed"); return GAMDP_ENOMEM; }
        c->cap_chain_scratch = need_scratch;
    }
    ChainParams cp;
    cp.mbs = (const DevMB*)(d + off_mb); cp.blks = (const DevBlk*)(d + off_blk); cp.n_mbs = (u32)n_mb;
    cp.n_twins = (u32)n_tw; cp.sync = (ChainSync*)(d + off_sync);
    std::memset(h + off_sync, 0, (n_tw + 1) * sizeof(ChainSync));
    cp.cursor = nullptr; cp.audit = (DevResult*)(d + off_aud); cp.out = (ChainOut*)(d + off_out); cp.win = (ChainWin*)(d + off_win);
    cp.scratch = c->d_chain_scratch; cp.ypad = ypad; cp.band = band;
    cp.max_rows = (u32)std::min<u64>(std::max<u64>(1, w[order[0]]), 0x7fffffffu);
    cp.host_out = (ChainOut*)(dm + mo_out); cp.host_done = (u32*)(dm + mo_done); cp.host_audit = (DevResult*)(dm + mo_aud); cp.host_win = (ChainWin*)(dm + mo_win);
    cp.epoch = c->chain_epoch;
    cp.skew_call = ~0u;
    if (diag().build) { static const char* const e = std::getenv("GAMDP_DIAG_CHAIN_SKEW"); if (e) cp.skew_call = (u32)std::atoi(e); }
    cp.two_waves = one_wave ? 0u : 1u;
    // N by window: the chains pick the cell of every call by the bases it touches (+ 64 on either side, as the batch path does);
    // GAMDP_N_BY_CONTIG=1 / the diagnostics build's GAMDP_DIAG_FORCE_N: by the contigs' flags, as in rounds 3-4
    cp.n_margin = 64; cp.n_by_contig = (chain_n_by_contig() || diag().force_n) ? 1u : 0u;
    if (diag().build) { static const char* const e = std::getenv("GAMDP_DIAG_N_WINDOW_SHRINK"); if (e) cp.n_margin = 64 - std::atoi(e); }
    run.n_by_contig = cp.n_by_contig != 0;
    if (hipEventCreate(&run.e0) != hipSuccess) { c->set_error("hipEventCreate failed"); return GAMDP_EHIP; }
    if (hipEventCreate(&run.e1) != hipSuccess) { (void)hipEventDestroy(run.e0); c->set_error("hipEventCreate failed"); return GAMDP_EHIP; }
    if (diag().timing) std::fprintf(stderr, "gamdp chain: %zu merge blocks, %llu blocks, %zu piece(s), %.1f MB of scratch (slots of up to %llu words), %llu twins, has_n %d\n", (size_t)n_mb, (unsigned long long)n_blk, pieces.size(), need_scratch * 4e-6, (unsigned long long)slotw_max, (unsigned long long)n_tw, (int)has_n);
    run.n_mb = n_mb; run.band = band; run.hmb = hmb; run.ms = ms; run.ss = ss;
    run.hout = (const ChainOut*)(hm + mo_out); run.done = (const volatile u32*)(hm + mo_done); run.haud = (const DevResult*)(hm + mo_aud); run.hwin = (const ChainWin*)(hm + mo_win);
    run.epoch = cp.epoch; run.stream = c->chain_stream; run.dout = cp.out;
    run.t_launch = std::chrono::steady_clock::now();
    bool ok = hipMemcpyAsync(d, h, off_out, hipMemcpyHostToDevice, c->chain_stream) == hipSuccess;
    ok = ok && hipEventRecord(run.e0, c->chain_stream) == hipSuccess;
    for (size_t pc = 0; ok && pc < pieces.size(); pc++) {   // (one launch unless the arena is too small for all the slots at once)
        cp.first_mb = pieces[pc].first;
        ok = launch_chain(cp, has_n || diag().force_n, (unsigned)(pieces[pc].second + n_tw), c->chain_stream) == 0;
    }
    ok = ok && hipEventRecord(run.e1, c->chain_stream) == hipSuccess;
    if (!ok) {
        (void)hipStreamSynchronize(c->chain_stream);
        (void)hipEventDestroy(run.e0); (void)hipEventDestroy(run.e1);
        c->set_error(std::string("chain launch: ") + hipGetErrorString(hipGetLastError()));
        return GAMDP_EHIP;
    }
    run.launched = true;
    for (u32 i : act) M[i].on_device = true;
    return 0;
}

// The host's machine of merge block `mi` over the records its chain left (after run.ended()).  Every call of the chain is
// derived twice: the device left, next to each result record, the window it ran (ChainWin), the host's machine derives its own
// next call from the records so far (Machine::pending, PctgBuilder.cc:1652-1677) and runs its own pre-checks; the two must
// agree in every number -- window, orientation, rows, status -- or the call fails with GAMDP_EHIP naming the merge block and the
// call: a chain kernel that derived a different window can not hand back a plausible wrong answer.
int replay_chain(Ctx* cc, const ChainRun& run, Machine& m, const u32 mi)
{
    static std::unordered_map<u32, std::vector<uint8_t>> no_cache;   // (never touched: the MAIN phase does not look at the slave's codes)
    static std::mutex no_mu;
    const u32 q = run.q_of[mi];
    const DevMB& x = run.hmb[q];
    const u32 n_dp = run.hout[q].n_dp;
    if ((run.hout[q].state & 0xffu) == 3u) {   // a call did not fit the chain's scratch slots: nothing of the chain is used, the round loop takes the merge block
        m.on_device = false;
        return 0;
    }
    u32 used = 0;
    auto differs = [&](const char* what, const u64 dev, const u64 host) {
        cc->set_error("internal: merge block " + std::to_string(mi) + ", call " + std::to_string(used) + " of its chain: the device's " + what + " is " +
                      std::to_string(dev) + ", the host's " + std::to_string(host));
        return GAMDP_EHIP;
    };
    while (m.phase == Machine::MAIN) {
        if (used >= n_dp) { cc->set_error("internal: the device's chain of merge block " + std::to_string(mi) + " is shorter than the host's"); return GAMDP_EHIP; }
        ITask t;
        m.pending(t, no_cache, no_mu);
        u64 X = 0, cells = 0;
        const int st = preflight(m.mlen, m.slen, run.band, t.begin_a, t.end_a, t.begin_b, t.end_b, false, false, &X, &cells);
        const ChainWin& w = run.hwin[x.audit_first + used];
        const DevResult& rec = run.haud[x.audit_first + used];
        if (w.begin_a != t.begin_a) return differs("begin_a", w.begin_a, t.begin_a);
        if (w.end_a != t.end_a) return differs("end_a", w.end_a, t.end_a);
        if (w.begin_b != t.begin_b) return differs("begin_b", w.begin_b, t.begin_b);
        if (w.end_b != t.end_b) return differs("end_b", w.end_b, t.end_b);
        if ((w.info & 1u) != (t.b_rc ? 1u : 0u)) return differs("orientation", w.info & 1u, t.b_rc ? 1u : 0u);
        if ((w.info >> 8) != (u32)st) return differs("pre-check status", w.info >> 8, (u64)st);
        if (w.X != (u32)X) return differs("row count", w.X, X);
        {   // the cell the device picked for this call (N by window): the host's own answer for the same window, with the full margin
            const bool want_n = st == GAMDP_ST_OK && x.has_n != 0 &&
                                (run.n_by_contig || run.ms->window_has_n(t.a_id, false, 0, (int64_t)t.begin_a - (int64_t)run.band - 64, (int64_t)t.begin_a + (int64_t)X - 1 + (int64_t)run.band + 64) ||
                                 run.ss->window_has_n(t.b_id, t.b_rc, 0, (int64_t)t.begin_b - 64, (int64_t)t.begin_b + (int64_t)X - 1 + 64));
            if (((w.info >> 1) & 1u) != (want_n ? 1u : 0u)) return differs("choice of the N-aware cell", (w.info >> 1) & 1u, want_n ? 1u : 0u);
        }
        if (st != GAMDP_ST_OK && (rec.flags >> 8) != (u32)st) return differs("record status", rec.flags >> 8, (u64)st);
        gamdp_result r;
        fill_result(rec, cells, r);
        m.feed(r);
        used++;
    }
    if (used != n_dp) { cc->set_error("internal: the device's chain of merge block " + std::to_string(mi) + " is longer than the host's"); return GAMDP_EHIP; }
    return 0;
}

struct CohortStats { int rounds = 0; double pending_ms = 0, align_ms = 0, feed_ms = 0; int rc = 0; };

// The round loop over one cohort of merge blocks on one context (= one host thread + one stream): every round collects the
// pending find_alignment calls of its machines into ONE L0 batch, feeds the results back and advances the machines.  A
// machine whose main chain is on the device (ChainRun) joins once its chain has ended and been replayed; until then the
// rounds go on without it -- so the tails of the short chains are aligned while the long chains run, and a round is whatever
// became ready while the last one was in flight.  Cohorts run concurrently: while one waits for its kernel, another builds
// pending calls (findHits over contig tails, descriptor preparation) or feeds results -- host work hides behind GPU work.
void run_cohort(Ctx* c, std::vector<Machine>& M, const std::vector<u32>& ids, std::unordered_map<u32, std::vector<uint8_t>>& rc_cache,
                std::mutex& rc_mu, CohortStats& st, const ChainRun* run)
{
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto msec = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::vector<ITask> tasks;
    std::vector<u32> owner, first, waiting;
    std::vector<gamdp_result> res;
    const bool chained = run && run->launched;
    if (chained)
        for (u32 i : ids)
            if (M[i].phase == Machine::MAIN && M[i].on_device) waiting.push_back(i);
    bool drained = false;   // the chain launch is known to be complete
    for (;;) {
        const auto t0 = now();
        if (!waiting.empty()) {
            size_t keep = 0;
            for (u32 i : waiting) {
                if (run->ended(run->q_of[i])) {
                    st.rc = replay_chain(c, *run, M[i], i);
                    if (st.rc) return;
                } else waiting[keep++] = i;
            }
            waiting.resize(keep);
        }
        owner.clear();
        for (u32 i : ids)
            if (M[i].phase != Machine::DONE && !(M[i].phase == Machine::MAIN && M[i].on_device)) owner.push_back(i);
        if (owner.empty()) {
            if (waiting.empty()) break;
            // nothing to do until a chain ends; a launch that is over without every flag up has failed
            if (drained) { c->set_error("internal: the chain launch ended without handing over merge block " + std::to_string(waiting[0])); st.rc = GAMDP_EHIP; return; }
            const hipError_t qs = hipStreamQuery(run->stream);
            if (qs == hipSuccess) { drained = true; continue; }
            if (qs != hipErrorNotReady) { c->set_error(std::string("chain kernel: ") + hipGetErrorString(qs)); st.rc = GAMDP_EHIP; return; }
            if (diag().timing && diag().build && std::chrono::duration<double>(now() - run->t_launch).count() > 5.0) {
                run->dump(stderr);   // diagnostics build: a watchdog instead of a blind wait
                std::_Exit(3);
            }
            for (int spin = 0; spin < 256; spin++) {
                bool any = false;
                for (u32 i : waiting) any = any || run->done[run->q_of[i]] == run->epoch;
                if (any) break;
                std::this_thread::yield();
            }
            continue;
        }
        // one call per machine -- two for a machine whose left AND right tails are due (independent of each other)
        tasks.clear();
        first.assign(owner.size(), 0);
        for (size_t q = 0; q < owner.size(); q++) {
            Machine& m = M[owner[q]];
            first[q] = (u32)tasks.size();
            tasks.push_back(ITask{});
            m.pending(tasks.back(), rc_cache, rc_mu);
            if (m.phase == Machine::LEFT && m.right_follows_left()) {
                m.phase = Machine::RIGHT;          // (pending() looks at the phase and the main chain's end points only)
                tasks.push_back(ITask{});
                m.pending(tasks.back(), rc_cache, rc_mu);
                m.phase = Machine::LEFT;
            }
        }
        const auto t1 = now();
        res.assign(tasks.size(), gamdp_result{});
        st.rc = c->align(tasks, res.data(), nullptr);
        if (st.rc) return;
        const auto t2 = now();
        for (size_t q = 0; q < owner.size(); q++) {
            Machine& m = M[owner[q]];
            const u32 cnt = (q + 1 < owner.size() ? first[q + 1] : (u32)tasks.size()) - first[q];
            m.feed(res[first[q]]);
            if (cnt == 2 && m.phase == Machine::RIGHT) m.feed(res[first[q] + 1]);   // (not RIGHT: the left call threw, the machine is done)
        }
        const auto t3 = now();
        st.pending_ms += msec(t0, t1); st.align_ms += msec(t1, t2); st.feed_ms += msec(t2, t3);
        st.rounds++;
        if (diag().timing && chained)
            std::fprintf(stderr, "gamdp cohort %p: round %d at %.2f ms after the chain launch: %zu calls of %zu machines, pending %.2f ms, align %.2f ms, %zu still on the device\n",
                         (void*)c, st.rounds, msec(run->t_launch, t0), tasks.size(), owner.size(), msec(t0, t1), msec(t1, t2), waiting.size());
    }
}

}  // namespace
}  // namespace gamdp

extern "C" int gamdp_align_merge_blocks(gamdp_ctx* ctx, const gamdp_seqset* master, const gamdp_seqset* slave,
                                        const gamdp_mb_in* in, size_t n, uint32_t band, gamdp_mb_out* out,
                                        gamdp_result* audit, uint32_t audit_stride)
{
    if (!ctx || !master || !slave || (n && (!in || !out))) return GAMDP_EINVAL;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    const SeqSet* ms = reinterpret_cast<const SeqSet*>(master);
    const SeqSet* ss = reinterpret_cast<const SeqSet*>(slave);
    if (band > GAMDP_MAX_BAND) { c->set_error("band exceeds GAMDP_MAX_BAND"); return GAMDP_ENOTSUP; }
    if (hipSetDevice(c->device) != hipSuccess) { c->set_error("hipSetDevice failed"); return GAMDP_EHIP; }
    return guarded(c, [&]() -> int {
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<Machine> M(n);
    std::vector<u64> weight(n, 0);
    for (size_t i = 0; i < n; i++) {
        Machine& m = M[i];
        m.in = &in[i]; m.out = &out[i]; m.ms = ms; m.ss = ss; m.band = band;
        if (!ms->has_codes() || !ss->has_codes()) { c->set_error("merge blocks need sequence sets with host codes"); return GAMDP_EINVAL; }
        if (in[i].m_id < 0 || in[i].s_id < 0 || (size_t)in[i].m_id >= ms->lens.size() || (size_t)in[i].s_id >= ss->lens.size()) {
            c->set_error("merge block " + std::to_string(i) + ": contig id out of range");
            return GAMDP_EINVAL;
        }
        m.mlen = ms->lens[in[i].m_id];
        m.slen = ss->lens[in[i].s_id];
        if (audit) { m.audit = audit + i * (size_t)audit_stride; m.audit_cap = audit_stride; }
        m.init();
        for (u32 k = 0; k < in[i].n_blocks && in[i].blocks; k++)
            weight[i] += (u64)frame_len(in[i].blocks[k].s_begin, in[i].blocks[k].s_end) * (2ull * band + 1);
    }
    const double k0_ms = c->kernel_ms; const u64 k0_n = c->kernel_launches;
    // Cohorts: host threads, each with its own context (stream, staging buffers, scratch arena) on this device; the merge
    // blocks are dealt by predicted cells (LPT), so the cohorts' chains have similar depth.  A round lasts as long as its
    // longest call, so smaller cohorts mean shorter rounds that overlap on the (nearly empty) GPU -- up to a point: 4 cohorts
    // of >= 48 merge blocks, up to 16 from ~1 500 merge blocks on (measured on the GAGE-shaped workloads when every call went
    // through this loop: 192 merge blocks 12.5 / 11.5 / 8.8 / 13.1 ms with 1 / 2 / 4 / 8 cohorts, 1 967 merge blocks 103 / 77 /
    // 60 / 54 / 63 ms with 1 / 2 / 4 / 8 / 12; with the main chains on the device only the tails are left, two rounds bound by
    // findHits on the host: 192 merge blocks 7.4 / 7.2 / 7.0 / 7.1 ms with 1 / 2 / 4 / 8, 1 967: 56 / 46 / 38 / 37 ms with
    // 2 / 4 / 8 / 16).
    // GAMDP_L1_COHORTS=k (<= 16) and GAMDP_L1_COHORT_MIN=m set the cap and the floor by hand.  Results do not depend on the
    // split: every machine only sees its own results.
    static const int forced_cohorts = [] { const char* e = std::getenv("GAMDP_L1_COHORTS"); return e ? std::min(16, std::max(1, std::atoi(e))) : 0; }();
    static const size_t cohort_min = [] { const char* e = std::getenv("GAMDP_L1_COHORT_MIN"); const long v = e ? std::atol(e) : 48; return (size_t)std::max(1L, v); }();
    const int K = forced_cohorts ? (int)std::max<size_t>(1, std::min<size_t>((size_t)forced_cohorts, n / cohort_min))
                                 : (int)std::max<size_t>(1, std::max(std::min<size_t>(4, n / cohort_min), std::min<size_t>(16, n / (4 * cohort_min))));
    while ((int)c->helpers.size() < K - 1) {
        Ctx* h = new (std::nothrow) Ctx();
        if (!h || h->init(c->device) != 0) { c->set_error("helper context: " + (h ? h->err : std::string("out of memory"))); delete h; return GAMDP_ENODEV; }
        c->helpers.push_back(h);
    }
    // the K cohort contexts share the device for this call: 1/K of the owner's budget each, whatever was set or
    // determined before (helpers that already exist included)
    if (c->arena_budget(true) == 0) { c->set_error("hipMemGetInfo failed"); return GAMDP_EHIP; }
    struct DivGuard {
        Ctx* c; int K;
        ~DivGuard() { c->arena_div = 1; for (int k = 1; k < K; k++) c->helpers[(size_t)k - 1]->arena_div = 1; }
    } div_guard{c, K};
    // (half of the budget for the chain launch's scratch slots, the other half for the K round loops beside it)
    c->arena_div = 2u * (u32)K;
    for (int k = 1; k < K; k++) {
        Ctx* h = c->helpers[(size_t)k - 1];
        h->arena_limit = c->arena_limit; h->arena_share = c->arena_share; h->arena_div = 2u * (u32)K;
        h->trim_scratch();
    }
    c->trim_scratch();
    const u64 chain_arena = c->arena_limit / (2ull * (u64)(c->arena_share ? c->arena_share : 1));
    if (c->d_chain_scratch && c->cap_chain_scratch * sizeof(u32) > chain_arena) { (void)hipFree(c->d_chain_scratch); c->d_chain_scratch = nullptr; c->cap_chain_scratch = 0; }
    std::vector<u32> part(n, 0);
    if (K > 1) partition_lpt(weight.data(), n, K, part.data());
    std::vector<std::vector<u32>> ids((size_t)K);
    for (size_t i = 0; i < n; i++) ids[part[i]].push_back((u32)i);

    if (!c->ref_event && hipEventCreate(&c->ref_event) != hipSuccess) { c->set_error("hipEventCreate failed"); return GAMDP_EHIP; }
    if (hipEventRecord(c->ref_event, c->stream) != hipSuccess || hipEventSynchronize(c->ref_event) != hipSuccess) { c->set_error("hipEventRecord failed"); return GAMDP_EHIP; }
    std::vector<std::vector<std::pair<float, float>>> intervals((size_t)K);
    std::vector<CohortStats> cst((size_t)K);
    std::unordered_map<u32, std::vector<uint8_t>> rc_cache;
    std::mutex rc_mu;
    // the main chains: one launch on the device, beside the round loops that take over what it hands back
    ChainRun run;
    {
        const int rc_chain = launch_main_chains(c, M, ms, ss, band, chain_arena, run);
        if (rc_chain) return rc_chain;
    }
    const bool chained = run.launched;
    if (!chained) {
        // no chain launch after all (another band, GAMDP_L1_ROUNDS=1, no main chain to run, a frame too long for the slots):
        // the round loops get the whole budget, not the half that was kept for the launch's scratch slots
        c->arena_div = (u32)K;
        for (int k = 1; k < K; k++) c->helpers[(size_t)k - 1]->arena_div = (u32)K;
        if (c->d_chain_scratch) { (void)hipFree(c->d_chain_scratch); c->d_chain_scratch = nullptr; c->cap_chain_scratch = 0; }
    }
    float chain_ms = 0, chain_at = 0;
    struct FreeGuard {   // (see Ctx::defer_frees)
        Ctx* c; int K;
        void set(bool on) { c->defer_frees = on; for (int k = 1; k < K; k++) c->helpers[(size_t)k - 1]->defer_frees = on; }
        ~FreeGuard() { set(false); c->flush_frees(); for (int k = 1; k < K; k++) c->helpers[(size_t)k - 1]->flush_frees(); }
    } free_guard{c, K};
    free_guard.set(chained);
    auto body = [&](int k) noexcept {
        Ctx* cc = k == 0 ? c : c->helpers[(size_t)k - 1];
        if (k > 0) { cc->kernel_ms = 0; cc->kernel_launches = 0; cc->ref_event = c->ref_event; }
        cc->interval_sink = &intervals[(size_t)k];
        const int rc_k = guarded(cc, [&]() -> int {
            if (hipSetDevice(cc->device) != hipSuccess) { cc->set_error("hipSetDevice failed"); return GAMDP_EHIP; }
            run_cohort(cc, M, ids[(size_t)k], rc_cache, rc_mu, cst[(size_t)k], &run);
            return 0;
        });
        if (rc_k && !cst[(size_t)k].rc) cst[(size_t)k].rc = rc_k;
        cc->interval_sink = nullptr;
        if (k > 0) cc->ref_event = nullptr;  // borrowed
    };
    {
        Threads pool;   // joined on every path out, also when starting a later thread fails
        for (int k = 1; k < K; k++) pool.start(body, k);
        body(0);
    }
    const int rc_fin = run.finish(c, &chain_at, &chain_ms);   // (also after an error in a cohort: the launch owns buffers of the context)
    for (int k = 0; k < K; k++)
        if (cst[(size_t)k].rc) {
            if (k > 0) c->set_error(c->helpers[(size_t)k - 1]->err);
            return cst[(size_t)k].rc;
        }
    if (rc_fin) return rc_fin;
    // statistics of this call; the helpers' kernel time is accounted to the caller's context
    gamdp_l1_stats& S = c->last_l1;
    S = gamdp_l1_stats{};
    for (int k = 1; k < K; k++) { c->kernel_ms += c->helpers[(size_t)k - 1]->kernel_ms; c->kernel_launches += c->helpers[(size_t)k - 1]->kernel_launches; }
    S.merge_blocks = n; S.cohorts = (uint32_t)K;
    for (size_t i = 0; i < n; i++) { S.dp_calls += out[i].n_dp; S.cells += out[i].cells; }
    std::vector<std::pair<float, float>> all;
    for (int k = 0; k < K; k++) {
        S.rounds = std::max<uint32_t>(S.rounds, (uint32_t)cst[(size_t)k].rounds + (chained ? 1u : 0u));
        S.host_pending_ms += cst[(size_t)k].pending_ms; S.host_feed_ms += cst[(size_t)k].feed_ms;
        all.insert(all.end(), intervals[(size_t)k].begin(), intervals[(size_t)k].end());
    }
    if (chained) all.emplace_back(chain_at, chain_at + chain_ms);
    S.launches = (uint32_t)(c->kernel_launches - k0_n);
    S.kernel_sum_ms = c->kernel_ms - k0_ms;
    std::sort(all.begin(), all.end());
    float cur_lo = 0, cur_hi = -1;
    for (auto& iv : all) {
        if (cur_hi < cur_lo) { cur_lo = iv.first; cur_hi = iv.second; }
        else if (iv.first <= cur_hi) cur_hi = std::max(cur_hi, iv.second);
        else { S.gpu_busy_ms += cur_hi - cur_lo; cur_lo = iv.first; cur_hi = iv.second; }
    }
    if (cur_hi >= cur_lo) S.gpu_busy_ms += cur_hi - cur_lo;
    S.wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    if (gamdp::diag().timing)
        std::fprintf(stderr, "gamdp_align_merge_blocks: %zu merge blocks, %d cohorts, %u rounds, %u launches: wall %.2f ms, GPU busy %.2f ms (kernels %.2f ms), pending %.2f ms, feed %.2f ms\n",
                     n, K, S.rounds, S.launches, S.wall_ms, S.gpu_busy_ms, S.kernel_sum_ms, S.host_pending_ms, S.host_feed_ms);
    return 0;
    });
}

extern "C" int gamdp_ctx_l1_stats(const gamdp_ctx* ctx, gamdp_l1_stats* out)
{
    if (!ctx || !out) return GAMDP_EINVAL;
    *out = reinterpret_cast<const Ctx*>(ctx)->last_l1;
    return 0;
}
