// Host-internal types of libgamdp (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <utility>
#include <vector>

#include "gamdp.h"
#include "gamdp_dev.h"

namespace gamdp {

constexpr int64_t FORCE_MAXGAP_ = 10;  // FORCE_MAXGAP_LEN, banded_smith_waterman.hpp:37

struct Ctx;

// diagnostics switches (gamdp_host.cpp): all false in the product build except `timing`
struct Diag {
    bool build = false;  // compiled with -DGAMDP_DIAG
    bool timing = false, skip_traceback = false, no_dirfree = false, count_mat = false, force_n = false;
};
const Diag& diag();

// banded_smith_waterman.cc:90-132 on plain numbers: GAMDP_ST_OK = has to run on the GPU, else the final status
int preflight(u64 alen, u64 blen, u64 band, u64 begin_a, u64 end_a, u64 begin_b, u64 end_b, bool fs, bool fe,
              u64* X_out, u64* cells_out);

struct DevSeq {
    u32* p2;  // word holding base 0 in the 2-bit plane
    u32* pn;  // word holding base 0 in the N plane
};

// RefSequence equivalent: host codes (1 B/base, for findHits and lazy reverse complements) plus the
// packed planes resident in HBM.
struct SeqSet {
    Ctx* ctx = nullptr;
    std::vector<std::vector<uint8_t>> codes;  // empty for packed-only sets (gamdp_seqset_create_synth)
    std::vector<u64> lens;
    std::vector<uint8_t> has_n;
    // N by window: npre[i][k] = number of N among bases [0, 256 k) of sequence i, forward orientation (empty for a sequence without N)
    std::vector<std::vector<u32>> npre;
    // may bases [lo, hi] of the view (reverse complement or not, chopped by `off` bases) hold an N?  Positions outside the sequence
    // do not count; the answer is by blocks of 256 bases, i.e. "yes" a little more often than the truth.
    bool window_has_n(u32 id, bool rc, u64 off, int64_t lo, int64_t hi) const
    {
        if (!has_n[id]) return false;
        if (id >= npre.size() || npre[id].empty()) return true;   // no counts kept: the contig's flag decides
        return npre_window_has_n(npre[id].data(), (int64_t)lens[id], rc, off, lo, hi);   // (gamdp_dev.h: shared with the chain kernels)
    }
    // the same counts on the device, for the chain kernels (all sequences with N in one buffer; nullptr for a sequence without)
    u32* d_npre = nullptr;
    std::vector<const u32*> dev_npre;
    std::vector<DevSeq> fwd;
    mutable std::vector<DevSeq> rc;  // reverse complements, uploaded on first use
    mutable std::vector<u32*> rc_allocs;
    mutable std::mutex rc_mu;        // the cohort threads of one merge-block call share a set (ensure_rc)
    u32 *d2 = nullptr, *dn = nullptr;

    int upload(Ctx* ctx, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n, bool ascii);
    int ensure_rc(const std::vector<u32>& ids, Ctx* use = nullptr) const;
    int upload_synth(Ctx* ctx, uint64_t first_pair, uint64_t stride_pairs, uint32_t n_pairs, uint64_t len);
    bool has_codes() const { return !codes.empty() || lens.empty(); }
    ~SeqSet();
};

// one find_alignment call with explicit sequence sets per operand (the tails of findBestAlignment
// put the slave contig in the `a` role, PctgBuilder.cc:1544-1551)
struct ITask {
    const SeqSet* sa;
    const SeqSet* sb;
    u32 a_id, b_id;
    u64 a_off, b_off;
    bool a_rc, b_rc, force_start, force_end;
    u32 band;
    u64 begin_a, end_a, begin_b, end_b;
};

// Where the tasks of a batch call come from: an array of ITask (the merge-block driver builds its own), or the caller's gamdp_task
// array read in place (round 6: taking 100 000 calls over into an ITask array first was 0.2 - 0.5 ms of the call and 13 MB of traffic).
struct TaskSrc {
    const ITask* it = nullptr;
    const gamdp_task* gt = nullptr;
    const SeqSet* sa = nullptr;
    const SeqSet* sb = nullptr;
    ITask operator[](size_t i) const
    {
        if (it) return it[i];
        const gamdp_task& t = gt[i];
        return ITask{sa, sb, t.a_id, t.b_id, t.a_off, t.b_off, t.a_rc != 0, t.b_rc != 0, t.force_start != 0, t.force_end != 0, t.band, t.begin_a, t.end_a, t.begin_b, t.end_b};
    }
};

// a validated task: device descriptor + what the launch planner needs
struct Prepared {
    DevTask dt;
    int kid;
    u64 cells;
    u64 dir_words;
};

struct Ctx {
    int device = -1;
    int n_cu = 0;
    u32 max_lds_per_wg = 65536;
    hipStream_t stream = nullptr;
    // A batch call's small N-aware launch runs BESIDE its big throughput launch (Ctx::align): a second stream, created on first use,
    // and the event that tells the first one when it is done
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_done = nullptr, aux_go = nullptr;
    std::string err;
    // Scratch-arena budget.  arena_limit is THE budget of this context's device as its owner sees it: what
    // gamdp_ctx_set_arena_bytes set, or -- while 0 -- 75 % of the free HBM at the first call.  It is never overwritten by
    // a call.  What one align() call may claim is arena_limit / (arena_share * arena_div): arena_share = contexts a
    // gamdp_multi handle created on this same device (set once by gamdp_multi_create), arena_div = contexts of THIS call
    // that run at the same time on the device (the two of a chunked batch, the K cohorts of a merge-block call; set by
    // the entry point for the duration of the call, which also hands the owner's budget to its helpers).
    u64 arena_limit = 0;
    bool arena_auto = true;   // arena_limit was determined from the free memory, not set by the caller
    u32 arena_share = 1, arena_div = 1;
    // The automatic budget: 75 % of (free HBM + what this context and its helper contexts hold as scratch).  Determined on first
    // use and again at every entry point of the C ABI (`refresh`): sequence sets created or destroyed between calls, reverse
    // complements uploaded on demand and other users of the device move what is free.
    u64 arena_budget(bool refresh = false);
    u64 arena_call() const { return arena_limit / ((u64)(arena_share ? arena_share : 1) * (u64)(arena_div ? arena_div : 1)); }
    void trim_scratch();  // gives back a scratch allocation larger than this call's share (before helper contexts allocate theirs)

    u32* d_scratch = nullptr; u64 cap_scratch = 0;  // in u32 words
    DevTask* d_tasks = nullptr; u64 cap_tasks = 0;
    DevResult* d_results = nullptr; u64 cap_results = 0;
    uint8_t* d_ops = nullptr; u64 cap_ops = 0;
    u32* d_cursor = nullptr;         // [cap_cursor] work-queue heads, then [cap_cursor][LS_COUNT] device-side launch statistics
    u32 cap_cursor = 0;
    // gamdp_ctx_launch_info: what the last gamdp_align_batch call launched (Ctx::align appends while log_launches is set)
    std::vector<gamdp_launch_info> launch_log;
    bool log_launches = false;
    u32 log_piece = 0;
    DevTask* h_tasks = nullptr;      // pinned staging
    DevResult* h_results = nullptr;  // pinned staging
    u64 cap_pinned = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    double kernel_ms = 0;
    u64 kernel_launches = 0;
    // merge-block calls: helper contexts on the same device (one per extra cohort thread, created on first use), the
    // reference event kernel intervals are measured against, the intervals of the current call, its statistics
    std::vector<Ctx*> helpers;
    hipEvent_t ref_event = nullptr;
    std::vector<std::pair<float, float>>* interval_sink = nullptr;
    gamdp_l1_stats last_l1{};

    void set_error(const std::string& s) { err = s; }
    int init(int dev);
    // work arrays of align(), kept between calls
    std::vector<Prepared> w_prep;
    std::vector<int> w_status;
    std::vector<u64> w_key;
    std::vector<int8_t> w_kid;   // per task of the batch under way: its kernel (-1: settled by the pre-checks) ...
    std::vector<u32> w_rows;     // ... and its rows
    std::vector<std::vector<u32>> w_groups;            // the tasks of the batch under way by kernel
    std::vector<u32> w_sort_tmp; std::vector<size_t> w_sort_count;   // scratch of the planner's sort

    // buffers of the merge-block chain kernel (gamdp_l1.cpp), kept between calls
    void* d_chain = nullptr; u64 cap_chain = 0;    // device: DevMB[] | DevBlk[] | DevResult audit[] | ChainOut[] | cursor
    void* h_chain = nullptr; u64 cap_hchain = 0;   // pinned: what is uploaded (DevMB[] | DevBlk[])
    // the chain launch runs beside the round loop's launches: own stream, own scratch slots, and a pinned coherent mirror
    // (ChainOut[] | done flags | DevResult audit[]) the chains write when they end
    hipStream_t chain_stream = nullptr;
    // hipFree / hipHostFree wait for the whole device; while the chain launch runs, a round loop that regrows a buffer keeps
    // the old one until the call is over (flush_frees) instead of waiting for the launch to end
    bool defer_frees = false;
    std::vector<void*> deferred_dev, deferred_host;
    void free_dev(void* p) { if (!p) return; if (defer_frees) deferred_dev.push_back(p); else (void)hipFree(p); }
    void free_host(void* p) { if (!p) return; if (defer_frees) deferred_host.push_back(p); else (void)hipHostFree(p); }
    void flush_frees()
    {
        for (void* p : deferred_dev) (void)hipFree(p);
        for (void* p : deferred_host) (void)hipHostFree(p);
        deferred_dev.clear(); deferred_host.clear();
    }
    u32* d_chain_scratch = nullptr; u64 cap_chain_scratch = 0;   // u32 words
    void* h_mirror = nullptr; u64 cap_mirror = 0;
    u32 chain_epoch = 0;

    int align(const TaskSrc& tasks, size_t n, gamdp_result* out, const gamdp_ops* ops);
    int align(const ITask* tasks, size_t n, gamdp_result* out, const gamdp_ops* ops) { TaskSrc s; s.it = tasks; return align(s, n, out, ops); }
    int align(const std::vector<ITask>& tasks, gamdp_result* out, const gamdp_ops* ops) { return align(tasks.data(), tasks.size(), out, ops); }
    ~Ctx();
};

// what gamdp_fasta points to: the host-side RefSequence (names + 1 B/base codes), no GPU involved
struct Fasta {
    std::vector<std::string> names;
    std::vector<std::vector<uint8_t>> codes;
    std::string err;
};

// Nothing may leave the C ABI as a C++ exception (the host program would std::terminate inside gam-merge instead of
// falling back to its CPU path): the bodies of the entry points that allocate or start threads run under guarded(),
// host threads are kept in a Threads object (joined on every path out), and thread bodies catch for themselves.
template <class F>
int guarded(Ctx* c, F&& f) noexcept
{
    auto note = [&](const char* what) noexcept { try { if (c) c->set_error(what); } catch (...) {} };
    try { return f(); }
    catch (const std::bad_alloc&) { note("out of host memory"); return GAMDP_ENOMEM; }
    catch (const std::exception& e) { try { if (c) c->set_error(std::string("host exception: ") + e.what()); } catch (...) {} return GAMDP_EHIP; }
    catch (...) { note("unknown host exception"); return GAMDP_EHIP; }
}
template <class F>
int guarded_thread_body(F&& f) noexcept   // the return code of a thread body that must not throw
{
    try { return f(); }
    catch (const std::bad_alloc&) { return GAMDP_ENOMEM; }
    catch (...) { return GAMDP_EHIP; }
}
struct Threads {
    std::vector<std::thread> th;
    template <class... A> void start(A&&... a) { th.emplace_back(std::forward<A>(a)...); }
    void join() { for (auto& t : th) if (t.joinable()) t.join(); }
    ~Threads() { join(); }
};

// DevResult record + the cell count of its call -> the C ABI's result (gamdp_host.cpp)
void fill_result(const DevResult& r, u64 cells, gamdp_result& o);

// deterministic longest-processing-time-first partition (gamdp_multi.cpp)
void partition_lpt(const u64* weights, size_t n, int parts, u32* part_of);

// ABlast::findHits (ablast.cc:41-76) on code arrays
void find_hits(const uint8_t* a, u64 alen, u64 a_start, u64 a_end, const uint8_t* b, u64 blen, u64 b_start, u64 b_end,
               u64 word, std::vector<uint32_t>& hits);

}  // namespace gamdp
