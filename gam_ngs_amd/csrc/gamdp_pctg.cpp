// Post-alignment stage of gam-merge's buildPctg (SURVEY.md 8f rows f1 + f2), host side, no GPU, no Boost:
// the merge lists whose MergeBlocks were just aligned (gamdp_align_merge_blocks) are cut, oriented and cleaned,
// every surviving list is woven into one paired contig, and the paired contigs are written as .gam.fasta / .pctgs.
//
// Reference (behaviour restated from reading; nothing here can be compiled from the reference in this image because
// PctgBuilder.hpp pulls in Boost.Graph):
//   BuildPctgFunctions.cc:86-92      order of the stages
//   PctgBuilder.cc:667-723           splitMergeBlocksByAlign      -> cut_at_failed_alignments
//   PctgBuilder.cc:543-665           splitMergeBlocksByDirection  -> orient_and_cut_at_turns
//   PctgBuilder.cc:507-541           sortMergeBlocksByDirection   -> put_lists_forward
//   PctgBuilder.cc:291-505           splitMergeBlocksByInclusions -> to_strand_coordinates_and_drop_inclusions
//   PctgBuilder.cc:102-168,172-288   append*ToPctg, buildPctgs    -> weave
//   PctgBuilder.cc:71-99, BuildPctgFunctions.cc:111-129, src/Merge.cc:380-385,437-452   ids + single-contig pctgs
//   io_contig.code.hpp:246-262, PairedContig.cc:305-349, src/Merge.cc:457-465           the two writers
// Known oddities of the reference are kept because they decide the output: the block at which a merge list turns
// around is dropped (:643-651), a list is abandoned at the first block that starts before its predecessor (:408,:469),
// a master contig that is revisited is NOT re-oriented (:228-242), and the slave never contributes a tail (:216,:271).
// The one step that needs BAM evidence (computeZScore, :147-168) is a callback supplied by the host.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <set>
#include <string>
#include <vector>

#include "gamdp.h"
#include "gamdp_internal.h"

namespace gamdp {
namespace {

typedef gamdp_mblock MB;
typedef std::vector<MB> MList;
typedef std::vector<MList> MLists;

// ---- stage 1 ---------------------------------------------------------------------------------------------------
// Blocks whose alignment failed disappear; the list is cut there unless the blocks on both sides of the hole lie on
// the same master contig, and at every place where consecutive survivors share neither contig.
void cut_at_failed_alignments(MLists& lists)
{
    MLists out;
    for (MList& in : lists) {
        MList run;
        bool hole = false;  // at least one failed block since the last survivor
        for (size_t i = 0; i < in.size(); i++) {
            MB& b = in[i];
            if (!b.align_ok) { hole = true; continue; }
            if (hole) b.ext_slave_prev = 0;
            if (i + 1 < in.size() && !in[i + 1].align_ok) b.ext_slave_next = 0;
            if (!run.empty()) {
                const MB& last = run.back();
                const bool joins = hole ? (last.m_id == b.m_id) : (last.m_id == b.m_id || last.s_id == b.s_id);
                if (!joins) { out.push_back(run); run.clear(); }
            }
            run.push_back(b);
            hole = false;
        }
        if (!run.empty()) out.push_back(run);
    }
    lists.swap(out);
}

// ---- stage 2 ---------------------------------------------------------------------------------------------------
// Walk each list propagating the strand of both contigs (the first master is forward, every alignment says whether
// the two contigs of its block are opposite) and the direction in which the list advances along the shared contig.
// Where the direction flips the list is cut and the block sitting at the turn is discarded, except when three
// consecutive blocks share a contig.
void orient_and_cut_at_turns(MLists& lists)
{
    MLists out;
    for (MList& in : lists) {
        MList run;
        bool start = true, cut_before = false, fwd = true, fwd_ref = true, m_rev = false, s_rev = false;
        int32_t m_cur = 0, s_cur = 0;
        for (size_t i = 0; i < in.size(); i++) {
            MB& b = in[i];
            const MB* nx = (i + 1 < in.size()) ? &in[i + 1] : nullptr;
            if (start) {
                m_cur = b.m_id; s_cur = b.s_id;
                m_rev = false; s_rev = b.align_rev != 0;
                b.m_rev = m_rev; b.s_rev = s_rev;
                if (cut_before) { b.ext_slave_prev = 0; cut_before = false; }
                if (nx) {
                    if (b.m_id == nx->m_id) fwd = b.m_start <= nx->m_start;
                    else fwd = s_rev ? (b.s_start >= nx->s_start) : (b.s_start <= nx->s_start);
                }
                start = false;
                fwd_ref = fwd;
                run.push_back(b);
                continue;
            }
            if (m_cur == b.m_id) s_rev = m_rev != (b.align_rev != 0);
            if (s_cur == b.s_id) m_rev = s_rev != (b.align_rev != 0);
            b.m_rev = m_rev; b.s_rev = s_rev;
            if (nx) {
                if (b.m_id == nx->m_id) fwd = m_rev ? (b.m_start >= nx->m_start) : (b.m_start <= nx->m_start);
                else fwd = s_rev ? (b.s_start >= nx->s_start) : (b.s_start <= nx->s_start);
                if (fwd != fwd_ref) {
                    const MB& last = run.back();
                    const bool same_master3 = last.m_id == b.m_id && b.m_id == nx->m_id;
                    const bool same_slave3 = last.s_id == b.s_id && b.s_id == nx->s_id;
                    if (!same_master3 && !same_slave3) {
                        run.back().ext_slave_next = 0;
                        cut_before = true;
                        start = true;
                        out.push_back(run);
                        run.clear();
                        continue;  // b itself goes nowhere
                    }
                }
            }
            run.push_back(b);
            m_cur = b.m_id; s_cur = b.s_id;
        }
        if (!run.empty()) out.push_back(run);
    }
    lists.swap(out);
}

// ---- stage 3 ---------------------------------------------------------------------------------------------------
// A list that runs backwards along the contig shared by its first two blocks is reversed.
void put_lists_forward(MLists& lists)
{
    for (MList& l : lists) {
        if (l.size() < 2) continue;
        const MB &p = l[0], &q = l[1];
        bool fwd;
        if (p.m_id == q.m_id) fwd = p.m_start <= q.m_start;
        else fwd = p.align_rev ? (p.s_start >= q.s_start) : (p.s_start <= q.s_start);
        if (fwd) continue;
        for (MB& b : l) std::swap(b.ext_slave_next, b.ext_slave_prev);
        std::reverse(l.begin(), l.end());
    }
}

// ---- stage 4 ---------------------------------------------------------------------------------------------------
void to_strand(MB& b, const Fasta& master, const Fasta& slave)
{
    if (b.m_rev) {
        const int32_t n = (int32_t)master.codes[b.m_id].size(), s = b.m_start;
        b.m_start = n - b.m_end - 1;
        b.m_end = n - s - 1;
        std::swap(b.m_ltail, b.m_rtail);
    }
    if (b.s_rev) {
        const int32_t n = (int32_t)slave.codes[b.s_id].size(), s = b.s_start;
        b.s_start = n - b.s_end - 1;
        b.s_end = n - s - 1;
        std::swap(b.s_ltail, b.s_rtail);
    }
}

// Coordinates move to the strand each contig is used on, then blocks nested inside a neighbour on the shared contig
// are removed: a block that swallows its predecessors replaces them, a block inside its predecessor is skipped (and
// the list is cut if the walk would continue on the other contig), and a block that starts before its predecessor
// without containing it ends the list.
void to_strand_coordinates_and_drop_inclusions(MLists& lists, const Fasta& master, const Fasta& slave)
{
    MLists out;
    for (MList& in : lists) {
        MList run;
        MB prev{};
        bool start = true;
        for (size_t i = 0; i < in.size(); i++) {
            MB b = in[i];
            const bool has_next = i + 1 < in.size();
            to_strand(b, master, slave);
            if (start) {
                start = false;
                run.push_back(b);
                prev = b;
                continue;
            }
            const bool on_master = prev.m_id == b.m_id;  // otherwise the slave contig is the shared one
            const int32_t p0 = on_master ? prev.m_start : prev.s_start, p1 = on_master ? prev.m_end : prev.s_end;
            const int32_t c0 = on_master ? b.m_start : b.s_start, c1 = on_master ? b.m_end : b.s_end;
            if (p0 > c0 && p1 <= c1) {  // b swallows its predecessor(s)
                while (!run.empty()) {
                    const MB& t = run.back();
                    const int32_t t0 = on_master ? t.m_start : t.s_start, t1 = on_master ? t.m_end : t.s_end;
                    const bool same = on_master ? (t.m_id == b.m_id) : (t.s_id == b.s_id);
                    if (!(t0 > c0 && t1 <= c1 && same)) break;
                    run.pop_back();
                }
                if (!run.empty() && run.back().m_id != b.m_id && run.back().s_id != b.s_id) {
                    run.back().ext_slave_next = 0;
                    out.push_back(run);
                    run.clear();
                }
                run.push_back(b);
                prev = b;
            } else if (p0 > c0) {  // b starts before its predecessor: the rest of the list is given up
                if (!run.empty()) run.back().ext_slave_next = 0;
                break;
            } else if (p1 >= c1) {  // b lies inside its predecessor
                if (has_next) {
                    const MB& nx = in[i + 1];
                    const bool stays = on_master ? (b.m_id == nx.m_id) : (b.s_id == nx.s_id);
                    if (stays) continue;
                    if (!run.empty()) run.back().ext_slave_next = 0;
                    out.push_back(run);
                    run.clear();
                    if (on_master) in[i + 1].ext_slave_prev = 0;  // (the slave-side twin of this line writes to a copy, :484)
                    start = true;
                }
            } else {
                run.push_back(b);
                prev = b;
            }
        }
        if (!run.empty()) out.push_back(run);
    }
    lists.swap(out);
}

// ---- paired contigs --------------------------------------------------------------------------------------------
struct Pctg {
    std::vector<uint8_t> codes;
    std::vector<gamdp_pctg_row> rows;  // CtgInPctgInfo list
    std::set<int32_t> master_ids, slave_ids;
};

std::vector<uint8_t> oriented(const std::vector<uint8_t>& c, bool rev)
{
    if (!rev) return c;
    std::vector<uint8_t> r(c.size());
    for (size_t i = 0; i < c.size(); i++) {
        const uint8_t x = c[c.size() - 1 - i];
        r[i] = x < 4 ? (uint8_t)(x ^ 1) : x;  // A<->T, C<->G, N stays (nucleotide.code.hpp:128-144)
    }
    return r;
}

void put(Pctg& p, bool is_master, int32_t id, const std::vector<uint8_t>& ctg, int32_t from, int32_t to, bool rev)
{
    if (to < from || from < 0 || (size_t)to >= ctg.size()) return;
    (is_master ? p.master_ids : p.slave_ids).insert(id);
    p.codes.insert(p.codes.end(), ctg.begin() + from, ctg.begin() + to + 1);
    p.rows.push_back(gamdp_pctg_row{from, to, id, (uint8_t)rev, (uint8_t)is_master, {0, 0}});
}

struct Weaver {
    const Fasta& master;
    const Fasta& slave;
    gamdp_region_vote_fn vote;
    void* user;
    int err = 0;

    // the aligned region itself: the master's copy, unless the two copies differ in length by more than 3 % and the
    // host's read-pair evidence prefers the slave's
    void put_region(Pctg& p, const MB& b, const std::vector<uint8_t>& m, const std::vector<uint8_t>& s)
    {
        p.master_ids.insert(b.m_id);
        p.slave_ids.insert(b.s_id);
        const int64_t ml = b.m_end >= b.m_start ? (int64_t)b.m_end - b.m_start + 1 : 0;
        const int64_t sl = b.s_end >= b.s_start ? (int64_t)b.s_end - b.s_start + 1 : 0;
        const int64_t big = std::max(ml, sl), small = std::min(ml, sl);
        bool use_master = (double)small >= 0.97 * (double)big;
        if (!use_master) {
            if (!vote) { err = GAMDP_EINVAL; return; }
            const int v = vote(user, b.m_id, b.m_start, b.m_end, b.s_id, b.s_start, b.s_end);
            if (v < 0) { err = v; return; }
            use_master = v == 0;
        }
        if (use_master) put(p, true, b.m_id, m, b.m_start, b.m_end, b.m_rev);
        else put(p, false, b.s_id, s, b.s_start, b.s_end, b.s_rev);
    }

    bool weave(const MList& l, Pctg& p)
    {
        std::vector<uint8_t> m, s;
        int32_t m_next = 0, s_next = 0, m_before = 0;
        for (size_t i = 0; i < l.size() && !err; i++) {
            const MB& b = l[i];
            if (i == 0) {
                m = oriented(master.codes[b.m_id], b.m_rev);
                s = oriented(slave.codes[b.s_id], b.s_rev);
                if (b.m_ltail && b.m_start > 0) put(p, true, b.m_id, m, 0, b.m_start - 1, b.m_rev);
                put_region(p, b, m, s);
            } else if (b.m_id == m_before) {  // the walk stays on the master contig (kept in its first orientation)
                s = oriented(slave.codes[b.s_id], b.s_rev);
                if (m_next <= b.m_start) {
                    put(p, true, b.m_id, m, m_next, b.m_start - 1, b.m_rev);
                    put_region(p, b, m, s);
                } else {
                    put(p, true, b.m_id, m, m_next, b.m_end, b.m_rev);
                }
            } else {  // ... or on the slave contig
                m = oriented(master.codes[b.m_id], b.m_rev);
                if (s_next <= b.s_start) {
                    put(p, false, b.s_id, s, s_next, b.s_start - 1, b.s_rev);
                    put_region(p, b, m, s);
                } else {
                    put(p, false, b.s_id, s, s_next, b.s_end, b.s_rev);
                    p.master_ids.insert(b.m_id);
                }
            }
            if (i + 1 == l.size() && b.m_rtail) {
                const int32_t n = (int32_t)m.size();
                if (n - b.m_end - 1 > 0) put(p, true, b.m_id, m, b.m_end + 1, n - 1, b.m_rev);
            }
            m_before = b.m_id;
            m_next = b.m_end + 1;
            s_next = b.s_end + 1;
        }
        return !err;
    }
};

bool ids_in_range(const MB& b, const Fasta& master, const Fasta& slave)
{
    return b.m_id >= 0 && (size_t)b.m_id < master.codes.size() && b.s_id >= 0 && (size_t)b.s_id < slave.codes.size();
}

int unflatten(const gamdp_mblock* blocks, const uint32_t* sizes, uint32_t n_lists, const Fasta& master, const Fasta& slave,
              MLists& lists)
{
    if (n_lists && (!sizes || !blocks)) return GAMDP_EINVAL;
    size_t at = 0;
    for (uint32_t i = 0; i < n_lists; i++) {
        lists.emplace_back(blocks + at, blocks + at + sizes[i]);
        for (const MB& b : lists.back())
            if (!ids_in_range(b, master, slave)) return GAMDP_EINVAL;
        at += sizes[i];
    }
    return 0;
}

void run_stages(MLists& lists, const Fasta& master, const Fasta& slave, unsigned stages)
{
    if (stages & GAMDP_STAGE_ALIGN) cut_at_failed_alignments(lists);
    if (stages & GAMDP_STAGE_DIRECTION) orient_and_cut_at_turns(lists);
    if (stages & GAMDP_STAGE_SORT) put_lists_forward(lists);
    if (stages & GAMDP_STAGE_INCLUSIONS) to_strand_coordinates_and_drop_inclusions(lists, master, slave);
}

const char LETTER[5] = {'A', 'T', 'C', 'G', 'N'};

}  // namespace

struct PctgSet {
    const Fasta* master;
    const Fasta* slave;
    std::vector<Pctg> pctgs;
    size_t merged = 0;  // pctgs[0..merged) come from merge lists, the rest are single master contigs
    bool finished = false;
    std::string err;
};

}  // namespace gamdp

using namespace gamdp;

extern "C" {

int gamdp_zscore_vote(const double* master_z, const double* slave_z, size_t n)
{
    // PctgBuilder.cc:155-166: per window the assembly whose |z| is smaller AND non-zero collects the evidence, a zero
    // hands it to the other one; ties count for nobody; the master wins a draw
    size_t m_ev = 0, s_ev = 0;
    for (size_t i = 0; i < n; i++) {
        const double m = master_z[i] < 0 ? -master_z[i] : master_z[i], s = slave_z[i] < 0 ? -slave_z[i] : slave_z[i];
        if (s < m) { if (s != 0) s_ev++; else m_ev++; }
        if (m < s) { if (m != 0) m_ev++; else s_ev++; }
    }
    return m_ev >= s_ev ? 0 : 1;
}

int gamdp_merge_lists_prepare(const gamdp_fasta* master, const gamdp_fasta* slave, const gamdp_mblock* blocks,
                              const uint32_t* list_sizes, uint32_t n_lists, unsigned stages, gamdp_mblock* out_blocks,
                              uint64_t cap_blocks, uint32_t* out_sizes, uint32_t cap_lists, uint32_t* n_out_lists)
{
    const Fasta* m = reinterpret_cast<const Fasta*>(master);
    const Fasta* s = reinterpret_cast<const Fasta*>(slave);
    if (!m || !s || !n_out_lists) return GAMDP_EINVAL;
    MLists lists;
    if (const int rc = unflatten(blocks, list_sizes, n_lists, *m, *s, lists)) return rc;
    run_stages(lists, *m, *s, stages);
    uint64_t total = 0;
    for (const MList& l : lists) total += l.size();
    *n_out_lists = (uint32_t)lists.size();
    if (lists.size() > cap_lists || total > cap_blocks || (lists.size() && (!out_sizes || !out_blocks))) return GAMDP_ENOMEM;
    uint64_t at = 0;
    for (size_t i = 0; i < lists.size(); i++) {
        out_sizes[i] = (uint32_t)lists[i].size();
        for (const MB& b : lists[i]) out_blocks[at++] = b;
    }
    return 0;
}

int gamdp_pctgs_create(const gamdp_fasta* master, const gamdp_fasta* slave, gamdp_pctgs** out)
{
    if (!master || !slave || !out) return GAMDP_EINVAL;
    PctgSet* p = new (std::nothrow) PctgSet();
    if (!p) return GAMDP_ENOMEM;
    p->master = reinterpret_cast<const Fasta*>(master);
    p->slave = reinterpret_cast<const Fasta*>(slave);
    *out = reinterpret_cast<gamdp_pctgs*>(p);
    return 0;
}

void gamdp_pctgs_destroy(gamdp_pctgs* p) { delete reinterpret_cast<PctgSet*>(p); }

const char* gamdp_pctgs_last_error(const gamdp_pctgs* p) { return p ? reinterpret_cast<const PctgSet*>(p)->err.c_str() : ""; }

int gamdp_pctgs_add_graph(gamdp_pctgs* set, const gamdp_mblock* blocks, const uint32_t* list_sizes, uint32_t n_lists,
                          gamdp_region_vote_fn vote, void* user)
{
    PctgSet* p = reinterpret_cast<PctgSet*>(set);
    if (!p || p->finished) return GAMDP_EINVAL;
    MLists lists;
    if (const int rc = unflatten(blocks, list_sizes, n_lists, *p->master, *p->slave, lists)) {
        p->err = "merge block refers to a contig id outside the assemblies";
        return rc;
    }
    run_stages(lists, *p->master, *p->slave, GAMDP_STAGE_ALL);
    std::vector<Pctg> made;
    Weaver w{*p->master, *p->slave, vote, user};
    for (const MList& l : lists) {
        if (l.empty()) continue;
        Pctg pc;
        if (!w.weave(l, pc)) {
            p->err = vote ? "the region vote callback failed" : "a block region needs read-pair evidence but no vote callback was given";
            return w.err;
        }
        if (!pc.codes.empty()) made.push_back(std::move(pc));
    }
    for (Pctg& pc : made) p->pctgs.push_back(std::move(pc));  // nothing of a failed graph is kept
    return 0;
}

int gamdp_pctgs_finish(gamdp_pctgs* set)
{
    PctgSet* p = reinterpret_cast<PctgSet*>(set);
    if (!p || p->finished) return GAMDP_EINVAL;
    p->merged = p->pctgs.size();
    std::vector<char> used(p->master->codes.size(), 0);
    for (const Pctg& pc : p->pctgs)
        for (int32_t id : pc.master_ids) used[id] = 1;
    for (size_t i = 0; i < used.size(); i++) {
        if (used[i] || p->master->codes[i].empty()) continue;
        Pctg pc;
        put(pc, true, (int32_t)i, p->master->codes[i], 0, (int32_t)p->master->codes[i].size() - 1, false);
        p->pctgs.push_back(std::move(pc));
    }
    p->finished = true;
    return 0;
}

uint32_t gamdp_pctgs_count(const gamdp_pctgs* p) { return p ? (uint32_t)reinterpret_cast<const PctgSet*>(p)->pctgs.size() : 0; }
uint32_t gamdp_pctgs_merged_count(const gamdp_pctgs* p)
{
    const PctgSet* s = reinterpret_cast<const PctgSet*>(p);
    return !s ? 0 : (uint32_t)(s->finished ? s->merged : s->pctgs.size());
}

const uint8_t* gamdp_pctgs_codes(const gamdp_pctgs* set, uint32_t i, uint64_t* len)
{
    const PctgSet* p = reinterpret_cast<const PctgSet*>(set);
    if (!p || i >= p->pctgs.size()) { if (len) *len = 0; return nullptr; }
    if (len) *len = p->pctgs[i].codes.size();
    return p->pctgs[i].codes.data();
}

uint32_t gamdp_pctgs_rows(const gamdp_pctgs* set, uint32_t i, gamdp_pctg_row* out, uint32_t cap)
{
    const PctgSet* p = reinterpret_cast<const PctgSet*>(set);
    if (!p || i >= p->pctgs.size()) return 0;
    const std::vector<gamdp_pctg_row>& r = p->pctgs[i].rows;
    for (uint32_t k = 0; k < cap && k < r.size() && out; k++) out[k] = r[k];
    return (uint32_t)r.size();
}

int gamdp_pctgs_contig_use(const gamdp_pctgs* set, uint8_t* master_used, uint8_t* slave_used)
{
    const PctgSet* p = reinterpret_cast<const PctgSet*>(set);
    if (!p) return GAMDP_EINVAL;
    if (master_used) std::memset(master_used, 0, p->master->codes.size());
    if (slave_used) std::memset(slave_used, 0, p->slave->codes.size());
    for (const Pctg& pc : p->pctgs) {
        if (master_used) for (int32_t id : pc.master_ids) master_used[id] = 1;
        if (slave_used) for (int32_t id : pc.slave_ids) slave_used[id] = 1;
    }
    return 0;
}

int gamdp_pctgs_write_fasta(const gamdp_pctgs* set, const char* path)
{
    const PctgSet* p = reinterpret_cast<const PctgSet*>(set);
    if (!p || !path) return GAMDP_EINVAL;
    FILE* f = std::fopen(path, "w");
    if (!f) return GAMDP_EINVAL;
    std::string rec;
    for (size_t i = 0; i < p->pctgs.size(); i++) {
        const std::vector<uint8_t>& c = p->pctgs[i].codes;
        rec = ">PairedContig_" + std::to_string(i);
        for (size_t k = 0; k < c.size(); k++) {
            if (k % 60 == 0) rec.push_back('\n');
            rec.push_back(LETTER[c[k] > 4 ? 4 : c[k]]);
        }
        rec.push_back('\n');
        if (std::fwrite(rec.data(), 1, rec.size(), f) != rec.size()) { std::fclose(f); return GAMDP_EINVAL; }
    }
    return std::fclose(f) ? GAMDP_EINVAL : 0;
}

int gamdp_pctgs_write_descriptors(const gamdp_pctgs* set, const char* path)
{
    const PctgSet* p = reinterpret_cast<const PctgSet*>(set);
    if (!p || !path) return GAMDP_EINVAL;
    FILE* f = std::fopen(path, "w");
    if (!f) return GAMDP_EINVAL;
    std::fputs("#Name\tSize\tAssembly\tContigID\tBegin\tEnd\tReversed\n", f);
    const size_t merged = p->finished ? p->merged : p->pctgs.size();
    for (size_t i = 0; i < p->pctgs.size(); i++) {
        if (i == merged) std::fputs("# ----------------------------------------------------\n", f);
        const Pctg& pc = p->pctgs[i];
        for (const gamdp_pctg_row& r : pc.rows) {
            const std::string& name = r.is_master ? p->master->names[r.ctg_id] : p->slave->names[r.ctg_id];
            std::fprintf(f, "PairedContig_%zu\t%zu\t%s\t%s\t%lld\t%lld\t%s\n", i, pc.codes.size(), r.is_master ? "Master" : "Slave",
                         name.c_str(), (long long)r.start, (long long)r.end, r.reversed ? "R" : "F");
        }
    }
    return std::fclose(f) ? GAMDP_EINVAL : 0;
}

}  // extern "C"

// ---- gam-merge's side outputs (src/Merge.cc:273-297, 335-373, 412-431) ------------------------------------------
namespace {
// Block.cc:810-862 / 865-925 up to the final flip: which contigs carry a block
int mark_block_contigs(const gamdp_block_rec* blocks, uint64_t n, uint32_t n_master, uint32_t n_slave, uint8_t* m, uint8_t* s)
{
    if ((n && !blocks) || !m || !s) return GAMDP_EINVAL;
    std::memset(m, 0, n_master);
    std::memset(s, 0, n_slave);
    for (uint64_t k = 0; k < n; k++) {
        const int32_t mi = blocks[k].m_ctg, si = blocks[k].s_ctg;
        if (mi < 0 || (uint32_t)mi >= n_master || si < 0 || (uint32_t)si >= n_slave) return GAMDP_EINVAL;  // the reference exits here
        m[mi] = 1;
        s[si] = 1;
    }
    return 0;
}
}  // namespace

int gamdp_no_blocks_contigs(const gamdp_block_rec* blocks, uint64_t n_blocks, uint32_t n_master, uint32_t n_slave,
                            uint8_t* master_nbc, uint8_t* slave_nbc)
{
    const int rc = mark_block_contigs(blocks, n_blocks, n_master, n_slave, master_nbc, slave_nbc);
    if (rc) return rc;
    for (uint32_t i = 0; i < n_master; i++) master_nbc[i] ^= 1;   // flip: contigs WITHOUT blocks
    for (uint32_t i = 0; i < n_slave; i++) slave_nbc[i] ^= 1;
    return 0;
}

int gamdp_no_blocks_after_filter(const gamdp_block_rec* filtered, uint64_t n_blocks, uint32_t n_master, uint32_t n_slave,
                                 const uint8_t* master_nbc, const uint8_t* slave_nbc, uint8_t* master_af, uint8_t* slave_af)
{
    if (!master_nbc || !slave_nbc) return GAMDP_EINVAL;
    const int rc = mark_block_contigs(filtered, n_blocks, n_master, n_slave, master_af, slave_af);
    if (rc) return rc;
    // |= the contigs that had no block before the filter, then flip (Block.cc:915-923)
    for (uint32_t i = 0; i < n_master; i++) master_af[i] = !(master_af[i] || master_nbc[i]);
    for (uint32_t i = 0; i < n_slave; i++) slave_af[i] = !(slave_af[i] || slave_nbc[i]);
    return 0;
}

int gamdp_pctgs_not_merged(const gamdp_pctgs* set, const uint8_t* slave_nbc_bf, const uint8_t* slave_nbc_af, uint8_t* not_merged)
{
    const PctgSet* p = reinterpret_cast<const PctgSet*>(set);
    if (!p || !slave_nbc_bf || !slave_nbc_af || !not_merged) return GAMDP_EINVAL;
    const size_t n = p->slave->codes.size();
    std::memset(not_merged, 0, n);   // first: used = in a paired contig | no blocks before | no blocks after (Merge.cc:416-426)
    for (const Pctg& pc : p->pctgs) for (int32_t id : pc.slave_ids) not_merged[id] = 1;
    for (size_t i = 0; i < n; i++) not_merged[i] = !(not_merged[i] || slave_nbc_bf[i] || slave_nbc_af[i]);
    return 0;
}

int gamdp_fasta_write_selected(const gamdp_fasta* fa, const uint8_t* select, const char* path)
{
    const Fasta* f = reinterpret_cast<const Fasta*>(fa);
    if (!f || !select || !path) return GAMDP_EINVAL;
    FILE* out = std::fopen(path, "w");
    if (!out) return GAMDP_EINVAL;
    std::string rec;
    for (size_t i = 0; i < f->codes.size(); i++) {
        if (!select[i]) continue;
        const std::vector<uint8_t>& c = f->codes[i];
        rec = ">" + f->names[i];
        for (size_t k = 0; k < c.size(); k++) {
            if (k % 60 == 0) rec.push_back('\n');
            rec.push_back(LETTER[c[k] > 4 ? 4 : c[k]]);
        }
        rec.push_back('\n');
        if (std::fwrite(rec.data(), 1, rec.size(), out) != rec.size()) { std::fclose(out); return GAMDP_EINVAL; }
    }
    return std::fclose(out) ? GAMDP_EINVAL : 0;
}

