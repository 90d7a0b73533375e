/* TEST INFRASTRUCTURE ONLY -- CPU restatement (plain C) of the reference's contig-pair alignment
 * hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the
 * product path (gam_ngs_amd/csrc) never links, loads or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function here is checked against the reference's own code
 * (oracle/_ref/libgamref.so, built from /root/reference by oracle/Makefile) on randomised and
 * hand-built cases by tests/test_oracle_vs_ref.py (runs only where /root/reference exists), and
 * against the committed golden vectors in tests/golden/ (generated from the reference by
 * tests/golden/make_golden.py) everywhere else.  The L1 driver (merge-block chain logic) has no
 * buildable reference (needs Boost.Graph) and is pinned only through the L0 calls it makes.
 *
 * Base codes follow lib/include/assembly/nucleotide.hpp:35-43: A=0 T=1 C=2 G=3 N=4.
 */
#ifndef GAMDP_ORACLE_H
#define GAMDP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    GAMDP_ORACLE_OK = 0,           /* a MyAlignment was produced                               */
    GAMDP_ORACLE_EMPTY = 1,        /* reference returns MyAlignment() (all zero)               */
    GAMDP_ORACLE_OUT_OF_RANGE = 2, /* reference throws std::out_of_range (Contig::at)          */
    GAMDP_ORACLE_INVALID = 3       /* arguments for which the reference has undefined behaviour */
};

/* edit-string alphabet, lib/include/alignment/my_alignment.hpp:57-62 */
enum { GAMDP_OP_GAP_A = 0, GAMDP_OP_GAP_B = 1, GAMDP_OP_MATCH = 2, GAMDP_OP_MISMATCH = 3 };

typedef struct gamdp_oracle_result {
    uint64_t begin_a, begin_b; /* MyAlignment::begin_a/begin_b                                 */
    int64_t score;             /* MyAlignment::score                                           */
    uint64_t n_match, length;  /* #MATCH ops, #ops                                             */
    uint64_t first_a, first_b; /* first_match_pos  (my_alignment.cc:167-193)                   */
    uint64_t last_a, last_b;   /* last_match_pos   (my_alignment.cc:228-262)                   */
    uint64_t cells;            /* x_size * y_size of the fill loops (0 on early return)        */
    double homology;           /* banded_smith_waterman.cc:319                                 */
    uint8_t first_found, last_found, status, pad_[5];
} gamdp_oracle_result;

/* Nucleotide(char), nucleotide.code.hpp:47-75 */
void gamdp_oracle_encode(const char* chars, uint64_t n, uint8_t* codes);
/* operator char(), nucleotide.code.hpp:111-126 */
void gamdp_oracle_decode(const uint8_t* codes, uint64_t n, char* chars);
/* reverse_complement, contig.code.hpp:187-229 + nucleotide.code.hpp:128-144 (in place) */
void gamdp_oracle_revcomp(uint8_t* codes, uint64_t n);

/* BandedSmithWaterman(band).find_alignment(...), banded_smith_waterman.cc:69-322.
 * ops (optional) receives min(length, ops_cap) edit ops in forward order. Returns status. */
int gamdp_oracle_align(const uint8_t* a, uint64_t alen, const uint8_t* b, uint64_t blen,
                       uint64_t band, uint64_t begin_a, uint64_t end_a, uint64_t begin_b,
                       uint64_t end_b, int force_start, int force_end, gamdp_oracle_result* out,
                       uint8_t* ops, uint64_t ops_cap);

/* ABlast(word).findHits(...), ablast.cc:41-76 + ablast.hpp:52-107.  Returns the number of hits
 * (only the first cap are written), ascending. */
int64_t gamdp_oracle_find_hits(const uint8_t* a, uint64_t alen, uint64_t a_start, uint64_t a_end,
                               const uint8_t* b, uint64_t blen, uint64_t b_start, uint64_t b_end,
                               uint64_t word, uint32_t* hits, uint64_t cap);

/* ---- L1: the merge-block chain driver (PctgBuilder.cc:726-844, 1361-1731) ---------------- */

typedef struct gamdp_oracle_block { /* the Block/Frame fields the driver reads                  */
    int32_t m_begin, m_end, s_begin, s_end; /* Frame::getBegin/getEnd (0-based inclusive)       */
    char m_strand, s_strand;                /* Frame::getStrand                                 */
    int64_t n_reads;                        /* Block::getReadsNumber                            */
} gamdp_oracle_block;

typedef struct gamdp_oracle_mb { /* MergeDescriptor.hpp:40-69, the fields alignMergeBlock uses  */
    uint8_t m_ltail, m_rtail, s_ltail, s_rtail;             /* in                               */
    uint8_t align_ok, align_rev, status, touched;           /* out; touched=coords were written */
    int32_t m_start, m_end, s_start, s_end;                 /* out (only if touched)            */
    uint32_t n_dp;                                          /* #find_alignment calls made       */
    uint64_t cells;                                         /* sum of x_size*y_size             */
} gamdp_oracle_mb;

/* PctgBuilder::alignMergeBlock on one merge block.  master/slave are code arrays (not modified).
 * audit (optional) receives the result of each DP call in call order (up to audit_cap).
 * status: OK, or OUT_OF_RANGE when any DP call would make the reference throw (the reference then
 * abandons the whole graph, ThreadedBuildPctg.cc:322-329). */
int gamdp_oracle_align_merge_block(const uint8_t* master, uint64_t mlen, const uint8_t* slave,
                                   uint64_t slen, const gamdp_oracle_block* blocks,
                                   uint32_t n_blocks, uint64_t band, gamdp_oracle_mb* mb,
                                   gamdp_oracle_result* audit, uint32_t audit_cap);

/* ---- helpers for bench.py's cpu_baseline leg and the synthetic workload ------------------- */

/* Synthetic pair k of SURVEY.md section 8(d): master = len uniform ACGT bases, slave = master with
 * 3% substitutions, 1% insertions, 1% deletions.  splitmix64 keyed by (0x47414D, k).
 * slave must hold at least len + len/8 + 64 codes; returns the slave length. */
uint64_t gamdp_oracle_synth_pair(uint64_t k, uint64_t len, uint8_t* master, uint8_t* slave);

/* Align n synthetic pairs (k = first .. first+n-1) of length len with the given band on `threads`
 * pthreads pulling from a shared cursor (mirrors ThreadedBuildPctg.cc:50-74).  Writes results[n]
 * if not NULL; returns total cells. */
uint64_t gamdp_oracle_bench_pairs(uint64_t first, uint64_t n, uint64_t len, uint64_t band,
                                  int threads, gamdp_oracle_result* results);

#ifdef __cplusplus
}
#endif
#endif
