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

}  // extern "C"
