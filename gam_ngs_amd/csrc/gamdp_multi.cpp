// Several GPUs of one node behind one handle: the MI355X equivalent of gam-merge's worker pool
// (lib/src/pctg/ThreadedBuildPctg.cc:143-197: N pthreads pull assembly graphs from a mutex-guarded cursor,
// :50-74, and their output lists are spliced in thread order, :180-181).
//
// Alignment tasks / merge blocks are independent (lib/src/pctg/BuildPctgFunctions.cc:82-84 mutates only its own
// MergeBlock), so N devices = a static partition of the list: longest-processing-time first by predicted cell
// updates, one host thread + context + resident copy of the sequences per device, every device writing the result
// slots of its own tasks.  There is no exchange step and therefore no collective: results are gathered by index on
// the host, and the output does not depend on how many devices took part (unlike the reference's `--threads > 1`,
// whose paired-contig order is nondeterministic).
#include <algorithm>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "gamdp.h"
#include "gamdp_internal.h"

namespace gamdp {

struct Multi {
    std::vector<gamdp_ctx*> ctxs;
    std::string err;
};

struct MultiSeqSet {
    Multi* m = nullptr;
    std::vector<gamdp_seqset*> sets;  // sets[d] lives on ctxs[d]'s device
};

// Greedy LPT: items in order of decreasing weight (ties: lower index first) go to the least-loaded part (ties:
// lower part).  Deterministic, so every process of a multi-process run derives the same assignment on its own.
void partition_lpt(const u64* weights, size_t n, int parts, u32* part_of)
{
    std::vector<u32> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return weights[x] > weights[y]; });
    std::vector<u64> load((size_t)parts, 0);
    for (u32 i : order) {
        int best = 0;
        for (int p = 1; p < parts; p++)
            if (load[(size_t)p] < load[(size_t)best]) best = p;
        part_of[i] = (u32)best;
        load[(size_t)best] += weights[i];
    }
}

namespace {

inline int32_t frame_len(int32_t b, int32_t e) { return e < b ? 0 : e - b + 1; }  // Frame.cc:124-127

// run fn(d) for every device on its own host thread; returns the first non-zero code in device order
template <class F>
int on_all_devices(Multi* m, F fn)
{
    const size_t nd = m->ctxs.size();
    std::vector<int> rc(nd, 0);
    auto body = [&](size_t d) noexcept { rc[d] = guarded(reinterpret_cast<Ctx*>(m->ctxs[d]), [&] { return fn(d); }); };
    if (nd == 1) body(0);
    else {
        Threads pool;   // joined on every path out
        for (size_t d = 0; d < nd; d++) pool.start(body, d);
    }
    for (size_t d = 0; d < nd; d++)
        if (rc[d]) {
            m->err = "device slot " + std::to_string(d) + ": " + gamdp_last_error(m->ctxs[d]);
            return rc[d];
        }
    return 0;
}

}  // namespace
}  // namespace gamdp

using namespace gamdp;

extern "C" {

int gamdp_partition_lpt(const uint64_t* weights, size_t n, int parts, uint32_t* part_of)
{
    if (parts < 1 || (n && (!weights || !part_of))) return GAMDP_EINVAL;
    partition_lpt(weights, n, parts, part_of);
    return 0;
}

int gamdp_multi_create(const int* devices, int n, gamdp_multi** out)
{
    if (!out || !devices || n < 1) return GAMDP_EINVAL;
    *out = nullptr;
    Multi* m = new (std::nothrow) Multi();
    if (!m) return GAMDP_ENOMEM;
    for (int d = 0; d < n; d++) {
        gamdp_ctx* c = nullptr;
        const int rc = gamdp_ctx_create(devices[d], &c);
        if (rc) {
            for (gamdp_ctx* x : m->ctxs) gamdp_ctx_destroy(x);
            delete m;
            return rc;
        }
        m->ctxs.push_back(c);
    }
    // a device listed more than once: its contexts split that device's scratch budget
    for (int d = 0; d < n; d++) {
        u32 same = 0;
        for (int e = 0; e < n; e++) same += devices[e] == devices[d];
        reinterpret_cast<Ctx*>(m->ctxs[(size_t)d])->arena_share = same;
    }
    *out = reinterpret_cast<gamdp_multi*>(m);
    return 0;
}

void gamdp_multi_destroy(gamdp_multi* mm)
{
    Multi* m = reinterpret_cast<Multi*>(mm);
    if (!m) return;
    for (gamdp_ctx* c : m->ctxs) gamdp_ctx_destroy(c);
    delete m;
}

int gamdp_multi_size(const gamdp_multi* mm) { return mm ? (int)reinterpret_cast<const Multi*>(mm)->ctxs.size() : 0; }

gamdp_ctx* gamdp_multi_ctx(gamdp_multi* mm, int i)
{
    Multi* m = reinterpret_cast<Multi*>(mm);
    return (m && i >= 0 && (size_t)i < m->ctxs.size()) ? m->ctxs[(size_t)i] : nullptr;
}

const char* gamdp_multi_last_error(const gamdp_multi* mm) { return mm ? reinterpret_cast<const Multi*>(mm)->err.c_str() : "null handle"; }

int gamdp_multi_seqset_create(gamdp_multi* mm, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n, int is_ascii,
                              gamdp_multi_seqset** out)
{
    Multi* m = reinterpret_cast<Multi*>(mm);
    if (!m || !out) return GAMDP_EINVAL;
    *out = nullptr;
    MultiSeqSet* s = new (std::nothrow) MultiSeqSet();
    if (!s) return GAMDP_ENOMEM;
    s->m = m;
    s->sets.assign(m->ctxs.size(), nullptr);
    // both assemblies are a few hundred MB packed at most: every device keeps its own resident copy
    const int rc = on_all_devices(m, [&](size_t d) { return gamdp_seqset_create(m->ctxs[d], seqs, lens, n, is_ascii, &s->sets[d]); });
    if (rc) {
        for (gamdp_seqset* x : s->sets) gamdp_seqset_destroy(x);
        delete s;
        return rc;
    }
    *out = reinterpret_cast<gamdp_multi_seqset*>(s);
    return 0;
}

int gamdp_multi_seqset_create_from_fasta(gamdp_multi* mm, const gamdp_fasta* f, gamdp_multi_seqset** out)
{
    if (!f) return GAMDP_EINVAL;
    const Fasta* fa = reinterpret_cast<const Fasta*>(f);
    std::vector<const uint8_t*> ptrs(fa->codes.size());
    std::vector<uint64_t> lens(fa->codes.size());
    for (size_t i = 0; i < fa->codes.size(); i++) { ptrs[i] = fa->codes[i].data(); lens[i] = fa->codes[i].size(); }
    return gamdp_multi_seqset_create(mm, ptrs.data(), lens.data(), (uint32_t)ptrs.size(), 0, out);
}

void gamdp_multi_seqset_destroy(gamdp_multi_seqset* ss)
{
    MultiSeqSet* s = reinterpret_cast<MultiSeqSet*>(ss);
    if (!s) return;
    for (gamdp_seqset* x : s->sets) gamdp_seqset_destroy(x);
    delete s;
}

gamdp_seqset* gamdp_multi_seqset_on(gamdp_multi_seqset* ss, int i)
{
    MultiSeqSet* s = reinterpret_cast<MultiSeqSet*>(ss);
    return (s && i >= 0 && (size_t)i < s->sets.size()) ? s->sets[(size_t)i] : nullptr;
}

int gamdp_multi_align_batch(gamdp_multi* mm, const gamdp_multi_seqset* set_a, const gamdp_multi_seqset* set_b,
                            const gamdp_task* tasks, size_t n, gamdp_result* out)
{
    Multi* m = reinterpret_cast<Multi*>(mm);
    const MultiSeqSet* sa = reinterpret_cast<const MultiSeqSet*>(set_a);
    const MultiSeqSet* sb = reinterpret_cast<const MultiSeqSet*>(set_b);
    if (!m || !sa || !sb || sa->m != m || sb->m != m || (n && (!tasks || !out))) return GAMDP_EINVAL;
    const size_t nd = m->ctxs.size();
    // weight = the cell updates the reference's fill loops would make (what GCUPS counts)
    std::vector<u64> w(n, 0);
    const SeqSet* la = reinterpret_cast<const SeqSet*>(sa->sets[0]);
    const SeqSet* lb = reinterpret_cast<const SeqSet*>(sb->sets[0]);
    for (size_t i = 0; i < n; i++) {
        const gamdp_task& t = tasks[i];
        if (t.a_id >= la->lens.size() || t.b_id >= lb->lens.size()) { m->err = "sequence id out of range"; return GAMDP_EINVAL; }
        const u64 al = la->lens[t.a_id], bl = lb->lens[t.b_id];
        if (t.a_off > al || t.b_off > bl) continue;  // INVALID, costs nothing
        u64 X = 0;
        (void)preflight(al - t.a_off, bl - t.b_off, t.band, t.begin_a, t.end_a, t.begin_b, t.end_b, t.force_start != 0,
                        t.force_end != 0, &X, &w[i]);
    }
    std::vector<u32> part(n);
    partition_lpt(w.data(), n, (int)nd, part.data());
    std::vector<std::vector<u32>> idx(nd);
    for (size_t i = 0; i < n; i++) idx[part[i]].push_back((u32)i);
    return on_all_devices(m, [&](size_t d) {
        const std::vector<u32>& mine = idx[d];
        if (mine.empty()) return 0;
        std::vector<gamdp_task> tk(mine.size());
        std::vector<gamdp_result> rs(mine.size());
        for (size_t k = 0; k < mine.size(); k++) tk[k] = tasks[mine[k]];
        const int rc = gamdp_align_batch(m->ctxs[d], sa->sets[d], sb->sets[d], tk.data(), tk.size(), rs.data(), nullptr);
        if (rc) return rc;
        for (size_t k = 0; k < mine.size(); k++) out[mine[k]] = rs[k];  // disjoint slots: no lock
        return 0;
    });
}

int gamdp_multi_align_merge_blocks(gamdp_multi* mm, const gamdp_multi_seqset* master, const gamdp_multi_seqset* slave,
                                   const gamdp_mb_in* in, size_t n, uint32_t band, gamdp_mb_out* out, gamdp_result* audit,
                                   uint32_t audit_stride)
{
    Multi* m = reinterpret_cast<Multi*>(mm);
    const MultiSeqSet* ms = reinterpret_cast<const MultiSeqSet*>(master);
    const MultiSeqSet* ss = reinterpret_cast<const MultiSeqSet*>(slave);
    if (!m || !ms || !ss || ms->m != m || ss->m != m || (n && (!in || !out))) return GAMDP_EINVAL;
    const size_t nd = m->ctxs.size();
    // predicted cells of a merge block: one pass of its block chain, rows = slave frame length (x_size follows the b
    // window, banded_smith_waterman.cc:93) times the band width; retries and tails are not predictable beforehand
    std::vector<u64> w(n, 0);
    for (size_t i = 0; i < n; i++)
        for (u32 k = 0; k < in[i].n_blocks && in[i].blocks; k++)
            w[i] += (u64)frame_len(in[i].blocks[k].s_begin, in[i].blocks[k].s_end) * (2ull * band + 1);
    std::vector<u32> part(n);
    partition_lpt(w.data(), n, (int)nd, part.data());
    std::vector<std::vector<u32>> idx(nd);
    for (size_t i = 0; i < n; i++) idx[part[i]].push_back((u32)i);
    return on_all_devices(m, [&](size_t d) {
        const std::vector<u32>& mine = idx[d];
        if (mine.empty()) return 0;
        std::vector<gamdp_mb_in> li(mine.size());
        std::vector<gamdp_mb_out> lo(mine.size());
        std::vector<gamdp_result> la(audit ? mine.size() * (size_t)audit_stride : 0);
        for (size_t k = 0; k < mine.size(); k++) li[k] = in[mine[k]];
        const int rc = gamdp_align_merge_blocks(m->ctxs[d], ms->sets[d], ss->sets[d], li.data(), li.size(), band, lo.data(),
                                                audit ? la.data() : nullptr, audit_stride);
        if (rc) return rc;
        for (size_t k = 0; k < mine.size(); k++) {
            out[mine[k]] = lo[k];
            if (audit)
                std::memcpy(audit + (size_t)mine[k] * audit_stride, la.data() + k * (size_t)audit_stride,
                            sizeof(gamdp_result) * audit_stride);
        }
        return 0;
    });
}

}  // extern "C"
