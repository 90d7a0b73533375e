// File formats on the input side of the path: the FASTA loader, so that the assemblies
// can go from disk into a packed sequence set without the rest of gam-merge.
//
// Reference: readNextContigID / readNextSequence / loadSequences,
//            lib/include/assembly/io_contig.code.hpp:511-538, 540-563, 568-596.
// Semantics kept: blanks and newlines before a header are skipped; anything but '>' there is an error; the contig
// name is the header up to the first blank; every character of the record other than '\n', ' ' and '>' is a base
// and goes through Nucleotide(char) (so "acgtn" in any case are themselves and everything else -- IUPAC codes,
// '\r', digits -- becomes N).  Difference by design: the reference sizes each contig from the BAM header
// (RefLength) and pads with N when the FASTA record is shorter; here a contig is exactly the bases read.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "gamdp.h"
#include "gamdp_internal.h"

namespace gamdp {

static inline uint8_t enc(char ch)
{
    switch (ch) {
    case 'A': case 'a': return 0;
    case 'T': case 't': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 3;
    default: return 4;
    }
}

// returns 0, or GAMDP_EINVAL with f.err set
static int parse_fasta(std::istream& is, Fasta& f)
{
    std::string data((std::istreambuf_iterator<char>(is)), std::istreambuf_iterator<char>());
    size_t p = 0;
    const size_t n = data.size();
    while (p < n) {
        while (p < n && (data[p] == ' ' || data[p] == '\n')) p++;  // readNextContigID: skip blanks before '>'
        if (p >= n) {
            // the reference would run readNextContigID on EOF here and fail in substr(); a file that ends in blank
            // lines is never produced by the tools upstream -- accept it
            break;
        }
        if (data[p] != '>') {
            f.err = std::string("Found invalid character: ") + data[p];
            return GAMDP_EINVAL;
        }
        size_t e = data.find('\n', p);
        if (e == std::string::npos) e = n;
        std::string id = data.substr(p + 1, e - p - 1);
        const size_t sp = id.find(' ');
        if (sp != std::string::npos) id = id.substr(0, sp);
        p = (e < n) ? e + 1 : n;
        std::vector<uint8_t> seq;
        while (p < n && data[p] != '>') {  // readNextSequence
            const char c = data[p++];
            if (c != '\n' && c != ' ') seq.push_back(enc(c));
        }
        f.names.push_back(id);
        f.codes.push_back(std::move(seq));
    }
    return 0;
}

struct Blocks {
    std::vector<gamdp_block_rec> recs;
};

// Frame's operator>> (Frame.cc:207-222): assembly id (ignored), contig id, strand, begin, end, blockReadsLen, readsLen
// through the stream's own formatted extraction -- what the reference uses, so odd tokens behave alike
static bool read_frame(std::istream& in, int32_t& ctg, char& strand, int32_t& begin, int32_t& end, uint64_t& block_len, uint64_t& reads_len)
{
    int32_t assembly;
    in >> assembly >> ctg >> strand >> begin >> end >> block_len >> reads_len;
    return (bool)in;
}

}  // namespace gamdp

using namespace gamdp;

extern "C" {

int gamdp_fasta_open(const char* path, gamdp_fasta** out)
{
    if (!path || !out) return GAMDP_EINVAL;
    *out = nullptr;
    std::ifstream ifs(path, std::ifstream::in | std::ifstream::binary);
    if (!ifs) return GAMDP_EINVAL;
    Fasta* f = new (std::nothrow) Fasta();
    if (!f) return GAMDP_ENOMEM;
    const int rc = parse_fasta(ifs, *f);
    if (rc) {
        std::fprintf(stderr, "libgamdp: %s: %s\n", path, f->err.c_str());
        delete f;
        return rc;
    }
    *out = reinterpret_cast<gamdp_fasta*>(f);
    return 0;
}

int gamdp_fasta_create(const char* const* names, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n, int is_ascii,
                       gamdp_fasta** out)
{
    if (!out || (n && (!names || !seqs || !lens))) return GAMDP_EINVAL;
    *out = nullptr;
    Fasta* f = new (std::nothrow) Fasta();
    if (!f) return GAMDP_ENOMEM;
    for (uint32_t i = 0; i < n; i++) {
        f->names.emplace_back(names[i] ? names[i] : "");
        std::vector<uint8_t> v(lens[i]);
        for (uint64_t k = 0; k < lens[i]; k++) {
            if (is_ascii) v[k] = enc((char)seqs[i][k]);
            else if (seqs[i][k] > 4) { delete f; return GAMDP_EINVAL; }
            else v[k] = seqs[i][k];
        }
        f->codes.push_back(std::move(v));
    }
    *out = reinterpret_cast<gamdp_fasta*>(f);
    return 0;
}

void gamdp_fasta_close(gamdp_fasta* f) { delete reinterpret_cast<Fasta*>(f); }

uint32_t gamdp_fasta_count(const gamdp_fasta* f) { return f ? (uint32_t)reinterpret_cast<const Fasta*>(f)->names.size() : 0; }

const char* gamdp_fasta_name(const gamdp_fasta* f, uint32_t i)
{
    const Fasta* x = reinterpret_cast<const Fasta*>(f);
    return (x && i < x->names.size()) ? x->names[i].c_str() : nullptr;
}

const uint8_t* gamdp_fasta_codes(const gamdp_fasta* f, uint32_t i, uint64_t* len)
{
    const Fasta* x = reinterpret_cast<const Fasta*>(f);
    if (!x || i >= x->codes.size()) { if (len) *len = 0; return nullptr; }
    if (len) *len = x->codes[i].size();
    return x->codes[i].data();
}

int gamdp_seqset_create_from_fasta(gamdp_ctx* ctx, const gamdp_fasta* f, gamdp_seqset** out)
{
    const Fasta* x = reinterpret_cast<const Fasta*>(f);
    if (!ctx || !x || !out) return GAMDP_EINVAL;
    std::vector<const uint8_t*> ptr(x->codes.size());
    std::vector<uint64_t> len(x->codes.size());
    for (size_t i = 0; i < x->codes.size(); i++) { ptr[i] = x->codes[i].data(); len[i] = x->codes[i].size(); }
    return gamdp_seqset_create(ctx, ptr.data(), len.data(), (uint32_t)ptr.size(), /*is_ascii=*/0, out);
}

int gamdp_blocks_open(const char* path, int64_t min_block_size, gamdp_blocks** out)
{
    if (!path || !out) return GAMDP_EINVAL;
    *out = nullptr;
    std::ifstream ifs(path);
    if (!ifs) return GAMDP_EINVAL;
    Blocks* b = new (std::nothrow) Blocks();
    if (!b) return GAMDP_ENOMEM;
    std::string line;
    while (ifs.good()) {
        std::getline(ifs, line);
        if (line.empty() || line[0] == '#') continue;
        std::stringstream ss(line);
        gamdp_block_rec r;
        std::memset(&r, 0, sizeof r);
        long long n = 0;
        ss >> n;
        r.n_reads = n;
        const bool ok = (bool)ss && read_frame(ss, r.m_ctg, r.m_strand, r.m_begin, r.m_end, r.m_block_reads_len, r.m_reads_len) &&
                        read_frame(ss, r.s_ctg, r.s_strand, r.s_begin, r.s_end, r.s_block_reads_len, r.s_reads_len);
        if (ok && r.n_reads >= min_block_size) b->recs.push_back(r);
    }
    *out = reinterpret_cast<gamdp_blocks*>(b);
    return 0;
}

void gamdp_blocks_close(gamdp_blocks* b) { delete reinterpret_cast<Blocks*>(b); }
uint64_t gamdp_blocks_count(const gamdp_blocks* b) { return b ? reinterpret_cast<const Blocks*>(b)->recs.size() : 0; }
const gamdp_block_rec* gamdp_blocks_data(const gamdp_blocks* b)
{
    const Blocks* x = reinterpret_cast<const Blocks*>(b);
    return (x && !x->recs.empty()) ? x->recs.data() : nullptr;
}

int gamdp_blocks_write(const char* path, const gamdp_block_rec* recs, uint64_t n)
{
    if (!path || (n && !recs)) return GAMDP_EINVAL;
    std::ofstream o(path);
    if (!o) return GAMDP_EINVAL;
    o << "# MasterAssemblyID\tMasterContigID\tMasterStrand\tMasterBegin\tMasterEnd\tMasterBlockReadsLength\tMasterReadsLength\t"
      << "SlaveAssemblyID\tSlaveContigID\tSlaveStrand\tSlaveBegin\tSlaveEnd\tSlaveBlockReadsLength\tSlaveReadsLength\n";
    for (uint64_t i = 0; i < n; i++) {
        const gamdp_block_rec& r = recs[i];
        o << (long long)r.n_reads << "\t" << 0 << "\t" << r.m_ctg << "\t" << r.m_strand << "\t" << r.m_begin << "\t" << r.m_end << "\t"
          << (unsigned long long)r.m_block_reads_len << "\t" << (unsigned long long)r.m_reads_len << "\t" << 0 << "\t" << r.s_ctg << "\t"
          << r.s_strand << "\t" << r.s_begin << "\t" << r.s_end << "\t" << (unsigned long long)r.s_block_reads_len << "\t"
          << (unsigned long long)r.s_reads_len << std::endl;
    }
    return o.good() ? 0 : GAMDP_EINVAL;
}

}  // extern "C"
