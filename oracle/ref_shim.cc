// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product path.
//
// C-ABI shim around the *unmodified* reference sources, compiled where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/libgamref.so (git-ignored).
// It exists so that (1) oracle/gamdp_oracle.c can be pinned against the real reference and
// (2) tests/golden/make_golden.py can generate the committed golden vectors.
//
// Reference entry points wrapped here:
//   BandedSmithWaterman::find_alignment   lib/src/alignment/banded_smith_waterman.cc:69-322
//   first_match_pos / last_match_pos      lib/src/alignment/my_alignment.cc:167-193, 228-262
//   ABlast::findHits                      lib/src/alignment/ablast.cc:41-76
//   reverse_complement / chop_begin       lib/include/assembly/contig.code.hpp:187-229, 257-261
//   Nucleotide(char)                      lib/include/assembly/nucleotide.code.hpp:47-75
//   readNextContigID / readNextSequence   lib/include/assembly/io_contig.code.hpp:511-563 (the loop of loadSequences, :578-590)
//   PairedContig, CtgInPctgInfo, operator<<(ostream&, const Contig&), writePctgDescriptors
//                                         lib/src/pctg/PairedContig.cc:33-189, 305-349; io_contig.code.hpp:246-262
#include <cstdint>
#include <cstring>
#include <list>
#include <stdexcept>
#include <string>
#include <thread>
#include <atomic>
#include <utility>
#include <vector>

#include "alignment/ablast.hpp"
#include "alignment/banded_smith_waterman.hpp"
#include "alignment/my_alignment.hpp"
#include "assembly/contig.hpp"
#include "assembly/io_contig.hpp"
#include "assembly/RefSequence.hpp"
#include "pctg/PairedContig.hpp"
#include <fstream>
#include <sstream>

extern "C" {

struct gamref_result {
    uint64_t begin_a, begin_b;
    uint64_t a_size, b_size;
    int64_t score;
    double homology;
    uint64_t length;
    uint64_t n_match;
    uint64_t first_a, first_b;
    uint64_t last_a, last_b;
    int32_t first_found, last_found;
    int32_t status;  // 0 ok, 2 = reference threw std::out_of_range, 3 = other exception
};

static Contig make_contig(const char* s, uint64_t n)
{
    Contig c(std::string("c"), size_t(n));
    for (uint64_t i = 0; i < n; i++) c.at(i) = s[i];  // Nucleotide::operator=(char)
    return c;
}

static void summarise(const MyAlignment& r, gamref_result* out, uint8_t* ops, uint64_t ops_cap)
{
    out->begin_a = r.begin_a();
    out->begin_b = r.begin_b();
    out->a_size = r.a_size();
    out->b_size = r.b_size();
    out->score = r.score();
    out->homology = r.homology();
    out->length = r.length();
    uint64_t nm = 0;
    for (uint64_t i = 0; i < r.sequence().size(); i++) {
        if (r.sequence()[i] == MATCH) nm++;
        if (ops && i < ops_cap) ops[i] = uint8_t(r.sequence()[i]);
    }
    out->n_match = nm;
    std::pair<uint64_t, uint64_t> p;
    out->first_found = first_match_pos(r, p) ? 1 : 0;
    out->first_a = p.first;
    out->first_b = p.second;
    out->last_found = last_match_pos(r, p) ? 1 : 0;
    out->last_a = p.first;
    out->last_b = p.second;
}

int gamref_find_alignment(const char* a, uint64_t alen, const char* b, uint64_t blen, uint64_t band,
                          uint64_t begin_a, uint64_t end_a, uint64_t begin_b, uint64_t end_b,
                          int force_start, int force_end, gamref_result* out, uint8_t* ops,
                          uint64_t ops_cap)
{
    std::memset(out, 0, sizeof(*out));
    try {
        Contig ca = make_contig(a, alen), cb = make_contig(b, blen);
        BandedSmithWaterman bsw{BandedSmithWaterman::size_type(band)};
        MyAlignment r = bsw.find_alignment(ca, begin_a, end_a, cb, begin_b, end_b, force_start != 0,
                                           force_end != 0);
        summarise(r, out, ops, ops_cap);
        out->status = 0;
    } catch (const std::out_of_range&) {
        out->status = 2;
    } catch (...) {
        out->status = 3;
    }
    return out->status;
}

// returns number of hits (may exceed cap; only the first cap are written)
int64_t gamref_find_hits(const char* a, uint64_t alen, uint64_t a_start, uint64_t a_end,
                         const char* b, uint64_t blen, uint64_t b_start, uint64_t b_end,
                         uint64_t word, uint32_t* hits, uint64_t cap)
{
    try {
        Contig ca = make_contig(a, alen), cb = make_contig(b, blen);
        ABlast ab{size_t(word)};
        std::list<uint32_t> h = ab.findHits(ca, a_start, a_end, cb, b_start, b_end);
        uint64_t k = 0;
        for (std::list<uint32_t>::const_iterator it = h.begin(); it != h.end(); ++it, ++k)
            if (k < cap) hits[k] = *it;
        return int64_t(h.size());
    } catch (...) {
        return -1;
    }
}

// in-place reverse complement through the reference's Contig functions; writes ACGTN chars
void gamref_reverse_complement(char* s, uint64_t n)
{
    Contig c = make_contig(s, n);
    reverse_complement(c);
    for (uint64_t i = 0; i < n; i++) s[i] = char(c.at(i));
}

// normalise chars exactly as the reference's loader does (acgtn any case, everything else -> N)
void gamref_normalise(char* s, uint64_t n)
{
    Contig c = make_contig(s, n);
    for (uint64_t i = 0; i < n; i++) s[i] = char(c.at(i));
}


// CPU-baseline helper for bench.py: n independent find_alignment calls (full windows, no force flags)
// of the REFERENCE code on `threads` workers pulling from a shared cursor, like the reference's
// pthread pool (lib/src/pctg/ThreadedBuildPctg.cc:50-74, 143-197).  Returns the total x_size*y_size.
uint64_t gamref_bench_pairs(const char* const* a, const uint64_t* alen, const char* const* b, const uint64_t* blen,
                            uint64_t n, uint64_t band, int threads, gamref_result* results)
{
    std::vector<Contig> ca, cb;
    for (uint64_t i = 0; i < n; i++) { ca.push_back(make_contig(a[i], alen[i])); cb.push_back(make_contig(b[i], blen[i])); }
    std::atomic<uint64_t> cursor(0), cells(0);
    auto worker = [&]() {
        BandedSmithWaterman bsw{BandedSmithWaterman::size_type(band)};
        for (;;) {
            const uint64_t k = cursor.fetch_add(1);
            if (k >= n) break;
            gamref_result r;
            std::memset(&r, 0, sizeof(r));
            try {
                MyAlignment al = bsw.find_alignment(ca[k], 0, alen[k] - 1, cb[k], 0, blen[k] - 1);
                summarise(al, &r, nullptr, 0);
            } catch (...) { r.status = 2; }
            if (results) results[k] = r;
            uint64_t x = blen[k] < alen[k] + band ? blen[k] : alen[k] + band;
            if (x > 500000) x = 500000;
            cells += x * (2 * band + 1);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < (threads < 1 ? 1 : threads); t++) th.emplace_back(worker);
    for (auto& t : th) t.join();
    return cells.load();
}


// FASTA loading with the reference's own readers, in the loop of loadSequences (io_contig.code.hpp:578-590) but
// with contigs sized by what is read (RefLength 0).  names: '\n'-separated; seqs: concatenated ACGTN chars.
// Returns the number of records, -1 on a reference exception, -2 if a buffer is too small.
int64_t gamref_load_fasta(const char* path, char* names, uint64_t names_cap, uint64_t* lens, uint64_t max_seqs,
                          char* seqs, uint64_t seqs_cap)
{
    try {
        std::ifstream ifs(path, std::ifstream::in);
        uint64_t n = 0, np = 0, sp = 0;
        while (!ifs.eof()) {
            std::string name;
            readNextContigID(ifs, name);
            Contig ctg(name, size_t(0));
            readNextSequence(ifs, ctg);
            if (n >= max_seqs || np + name.size() + 1 > names_cap || sp + ctg.size() > seqs_cap) return -2;
            std::memcpy(names + np, name.data(), name.size());
            np += name.size();
            names[np++] = '\n';
            for (size_t i = 0; i < ctg.size(); i++) seqs[sp++] = char(ctg.at(i));
            lens[n++] = ctg.size();
        }
        return int64_t(n);
    } catch (...) {
        return -1;
    }
}


// Renders paired contigs with the reference's own classes and writers.  The caller describes each paired contig as
// a list of pieces (assembly, contig id, first, last, reversed flag); a piece is appended the way the reference's
// appendMasterToPctg / appendSlaveToPctg do it through PairedContig's public interface (resize + operator[] +
// getMergeList().push_back(CtgInPctgInfo(...)), PctgBuilder.cc:102-132 -- those members themselves cannot be compiled
// here).  A piece whose source flag is set is cut from the reference's reverse_complement() of the contig.
// pieces: 7 int64 per piece = {pctg index, is_master, ctg id, start, end, reversed flag of the row, source is
// reverse-complemented}, sorted by pctg index.
// Output: fasta text (".gam.fasta": `os << pctg << std::endl` per paired contig, src/Merge.cc:458) and the .pctgs text.
int64_t gamref_render_pctgs(const char* const* m_names, const char* const* m_seqs, uint32_t n_master,
                            const char* const* s_names, const char* const* s_seqs, uint32_t n_slave,
                            const int64_t* pieces, uint64_t n_pieces, uint32_t n_pctgs, uint64_t first_single,
                            char* fasta_out, uint64_t fasta_cap, char* desc_out, uint64_t desc_cap)
{
    RefSequence masterRef(n_master), slaveRef(n_slave);
    for (uint32_t i = 0; i < n_master; i++) {
        masterRef[i].RefName = m_names[i];
        masterRef[i].Sequence = new Contig(make_contig(m_seqs[i], std::strlen(m_seqs[i])));
        masterRef[i].Sequence->set_name(m_names[i]);
        masterRef[i].RefLength = (int32_t)masterRef[i].Sequence->size();
    }
    for (uint32_t i = 0; i < n_slave; i++) {
        slaveRef[i].RefName = s_names[i];
        slaveRef[i].Sequence = new Contig(make_contig(s_seqs[i], std::strlen(s_seqs[i])));
        slaveRef[i].Sequence->set_name(s_names[i]);
        slaveRef[i].RefLength = (int32_t)slaveRef[i].Sequence->size();
    }
    std::list<PairedContig> result;
    uint64_t at = 0;
    for (uint32_t p = 0; p < n_pctgs; p++) {
        PairedContig pctg;
        while (at < n_pieces && pieces[7 * at] == (int64_t)p) {
            const int64_t* q = pieces + 7 * at++;
            const bool is_master = q[1] != 0;
            const int32_t id = (int32_t)q[2];
            Contig ctg(*(is_master ? masterRef[id].Sequence : slaveRef[id].Sequence));
            if (q[6]) reverse_complement(ctg);
            if (is_master) pctg.addMasterCtgId(id); else pctg.addSlaveCtgId(id);
            int64_t idx = pctg.size();
            pctg.resize(pctg.size() + (q[4] - q[3] + 1));
            for (int64_t i = q[3]; i <= q[4]; i++) pctg[idx++] = ctg.at(i);
            pctg.getMergeList().push_back(CtgInPctgInfo(id, q[3], q[4], q[5] != 0, is_master));
        }
        result.push_back(pctg);
    }
    uint64_t pctg_id = 0;  // src/Merge.cc:380-385
    for (std::list<PairedContig>::iterator it = result.begin(); it != result.end(); ++it) it->setId(pctg_id++);
    std::ostringstream fa, de;
    for (std::list<PairedContig>::const_iterator it = result.begin(); it != result.end(); ++it) fa << *it << std::endl;
    writePctgDescriptors(de, result, masterRef, slaveRef, first_single);
    for (uint32_t i = 0; i < n_master; i++) delete masterRef[i].Sequence;
    for (uint32_t i = 0; i < n_slave; i++) delete slaveRef[i].Sequence;
    const std::string f = fa.str(), d = de.str();
    if (f.size() + 1 > fasta_cap || d.size() + 1 > desc_cap) return -1;
    std::memcpy(fasta_out, f.c_str(), f.size() + 1);
    std::memcpy(desc_out, d.c_str(), d.size() + 1);
    return (int64_t)f.size();
}

// The contigs with select[i] != 0 as gam-merge writes its ".noblocks.*.fasta" / ".notmerged.fasta" side outputs:
// `stream << *contig << std::endl` with the reference's own operator<<(ostream&, const Contig&) (src/Merge.cc:350, 370, 429).
int64_t gamref_render_contigs(const char* const* names, const char* const* seqs, uint32_t n, const uint8_t* select,
                              char* out, uint64_t cap)
{
    std::ostringstream os;
    for (uint32_t i = 0; i < n; i++) {
        if (!select[i]) continue;
        Contig ctg(make_contig(seqs[i], std::strlen(seqs[i])));
        ctg.set_name(names[i]);
        os << ctg << std::endl;
    }
    const std::string t = os.str();
    if (t.size() + 1 > cap) return -1;
    std::memcpy(out, t.c_str(), t.size() + 1);
    return (int64_t)t.size();
}

}  // extern "C"
