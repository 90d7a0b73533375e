/* TEST INFRASTRUCTURE ONLY -- see gamdp_oracle.h.  Plain-C restatement, written from the reference's
 * behaviour (file:line cited per function), not copied: flat int32 band buffer instead of the
 * reference's row-pointer int64 matrix, ops collected into a reversed array instead of a std::list.
 * Integer conversions deliberately mimic the reference's `long` / `unsigned long` expressions,
 * because the end-cell search depends on them (see comments marked [types]).
 */
#define _POSIX_C_SOURCE 200809L
#include "gamdp_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define G_GAP (-8)          /* GAP_SCORE, my_alignment.hpp:46 */
#define FORCE_MAXGAP 10     /* FORCE_MAXGAP_LEN, banded_smith_waterman.hpp:37 */
#define MAX_ROWS 500000ULL  /* BSW_MAX_ALIGNMENT, banded_smith_waterman.hpp:39 */
#define MIN_HOMOLOGY 95.0   /* PctgBuilder.hpp:63-73 */

/* local SCORING_MATRIX of banded_smith_waterman.cc:80-88 */
static inline int32_t score_of(uint8_t p, uint8_t q)
{
    if (p == q) return 5;
    if (p == 4 || q == 4) return 0;
    return -4;
}

static inline int32_t imax(int32_t x, int32_t y) { return x > y ? x : y; }

void gamdp_oracle_encode(const char* s, uint64_t n, uint8_t* c)
{
    for (uint64_t i = 0; i < n; i++) {
        switch (s[i]) {
        case 'A': case 'a': c[i] = 0; break;
        case 'T': case 't': c[i] = 1; break;
        case 'C': case 'c': c[i] = 2; break;
        case 'G': case 'g': c[i] = 3; break;
        default: c[i] = 4; break;
        }
    }
}

void gamdp_oracle_decode(const uint8_t* c, uint64_t n, char* s)
{
    static const char L[5] = {'A', 'T', 'C', 'G', 'N'};
    for (uint64_t i = 0; i < n; i++) s[i] = L[c[i] > 4 ? 4 : c[i]];
}

void gamdp_oracle_revcomp(uint8_t* c, uint64_t n)
{
    static const uint8_t comp[5] = {1, 0, 3, 2, 4};
    for (uint64_t i = 0; i < n; i++) c[i] = comp[c[i] > 4 ? 4 : c[i]];
    for (uint64_t i = 0, j = n; i + 1 < j; i++) {
        j--;
        uint8_t t = c[i];
        c[i] = c[j];
        c[j] = t;
    }
}

/* first_match_pos (my_alignment.cc:167-193) and last_match_pos (:228-262) over an op array */
static void match_positions(const uint8_t* ops, uint64_t n, gamdp_oracle_result* r)
{
    uint64_t pa = r->begin_a, pb = r->begin_b;
    r->first_found = 0;
    uint64_t i = 0;
    for (; i < n; i++) {
        uint8_t op = ops[i];
        if (op == GAMDP_OP_MATCH) { r->first_found = 1; break; }
        if (op == GAMDP_OP_GAP_A) pb++;
        else if (op == GAMDP_OP_GAP_B) pa++;
        else { pa++; pb++; }
    }
    r->first_a = pa;
    r->first_b = pb;

    r->last_a = r->begin_a;
    r->last_b = r->begin_b;
    r->last_found = 0;
    pa = r->begin_a;
    pb = r->begin_b;
    for (i = 0; i < n; i++) {
        switch (ops[i]) {
        case GAMDP_OP_MATCH:
            r->last_found = 1;
            r->last_a = pa;
            r->last_b = pb;
            pa++; pb++;
            break;
        case GAMDP_OP_GAP_A: pb++; break;
        case GAMDP_OP_GAP_B: pa++; break;
        default: pa++; pb++; break;
        }
    }
}

int gamdp_oracle_align(const uint8_t* a, uint64_t alen, const uint8_t* b, uint64_t blen,
                       uint64_t band, uint64_t begin_a, uint64_t end_a, uint64_t begin_b,
                       uint64_t end_b, int fs, int fe, gamdp_oracle_result* out, uint8_t* ops_out,
                       uint64_t ops_cap)
{
    memset(out, 0, sizeof(*out));
    out->status = GAMDP_ORACLE_EMPTY;

    /* banded_smith_waterman.cc:90-97 */
    if (end_b < begin_b) return out->status;
    /* sizes beyond what any caller produces (keeps the signed arithmetic below exact) */
    if (band > (1u << 20) || alen >= (1ULL << 40) || blen >= (1ULL << 40) || begin_a >= (1ULL << 40)) {
        out->status = GAMDP_ORACLE_INVALID;
        return out->status;
    }
    if (begin_b >= blen) {
        /* b.at(begin_b) (or a.at(pos) under force_start) throws in the row-0 loop as soon as one
         * column qualifies (:116-131); with no qualifying column the reference runs into UB. */
        int any = 0;
        for (uint64_t j = 0; j < 2 * band + 1 && !any; j++) {
            int64_t pos = (int64_t)begin_a - (int64_t)band + (int64_t)j;
            if (pos < 0) continue;
            if (!fs) any = pos < (int64_t)alen;
            else any = pos <= FORCE_MAXGAP || pos < (int64_t)alen;
        }
        out->status = any ? GAMDP_ORACLE_OUT_OF_RANGE : GAMDP_ORACLE_INVALID;
        return out->status;
    }
    if (end_b >= blen) end_b = blen - 1;
    uint64_t X = end_b - begin_b + 1;
    if (alen + band - begin_a < X) X = alen + band - begin_a;
    if (X > MAX_ROWS) X = MAX_ROWS;
    if (X == 0) {
        out->status = GAMDP_ORACLE_INVALID;
        return out->status;
    }
    const uint64_t Y = 2 * band + 1;
    const int64_t w = (int64_t)band, ba = (int64_t)begin_a, la = (int64_t)alen;
    out->cells = X * Y;

    int32_t* H = (int32_t*)calloc(X * Y, sizeof(int32_t)); /* :102-107, zero-initialised */
    if (!H) {
        out->status = GAMDP_ORACLE_INVALID;
        return out->status;
    }
#define HH(i, j) H[(uint64_t)(i) * Y + (uint64_t)(j)]

    /* row 0, :112-132 */
    for (uint64_t j = 0; j < Y; j++) {
        int64_t pos = ba - w + (int64_t)j;
        if ((!fs && pos >= 0 && pos < la) || (fs && pos >= 0 && pos <= FORCE_MAXGAP)) {
            if (pos >= la) { /* only reachable with force_start: a.at(pos) throws */
                free(H);
                out->status = GAMDP_ORACLE_OUT_OF_RANGE;
                return out->status;
            }
            int32_t d = score_of(a[pos], b[begin_b]);
            if (pos > 0 && j > 0) HH(0, j) = imax(imax(d, G_GAP), HH(0, j - 1)); /* left has NO gap */
            else HH(0, j) = imax(G_GAP, d);
        }
        if (fs && pos > FORCE_MAXGAP && pos < la) {
            int32_t d = score_of(a[pos], b[begin_b]);
            HH(0, j) = (pos > 0 && j > 0) ? imax(d, HH(0, j - 1)) : d;
        }
    }

    /* rows >= 1, :135-171 */
    for (uint64_t i = 1; i < X; i++) {
        const uint8_t bb = b[begin_b + i];
        for (uint64_t j = 0; j < Y; j++) {
            int64_t pos = ba + (int64_t)i + (int64_t)j - w;
            if (pos < 0 || pos >= la) continue;
            int32_t d = score_of(a[pos], bb);
            int has_up = (j < Y - 1), has_left = (j > 0);
            int32_t up = has_up ? HH(i - 1, j + 1) + G_GAP : G_GAP;
            if (pos == 0) {
                if (!fs || i <= FORCE_MAXGAP) HH(i, j) = has_up ? imax(imax(d, up), G_GAP) : imax(d, G_GAP);
                else HH(i, j) = has_up ? imax(d, up) : d;
            } else {
                int32_t dg = HH(i - 1, j) + d;
                int32_t left = has_left ? HH(i, j - 1) + G_GAP : G_GAP;
                if (has_up && has_left) HH(i, j) = imax(imax(dg, up), left);
                else if (has_up) HH(i, j) = imax(dg, up);
                else if (has_left) HH(i, j) = imax(dg, left);
                else HH(i, j) = dg;
            }
        }
    }

    /* end cell, :174-212.  Strict '>' : first maximum in scan order wins. */
    int found = 0;
    int64_t mi = 0, mj = 0;
    int32_t best = 0;
    if (!fe) {
        for (uint64_t j = 0; j < Y; j++) {
            int64_t pos = ba + (int64_t)(X - 1) + (int64_t)j - w;
            if (pos >= 0 && (uint64_t)pos <= end_a) { /* [types] signed pos vs unsigned end_a */
                int32_t v = HH(X - 1, j);
                if (!found || v > best) { found = 1; mi = (int64_t)(X - 1); mj = (int64_t)j; best = v; }
            }
        }
    }
    {
        /* [types] `int_type(end_a) >= (begin_a+_band_size)` compares as unsigned long */
        int ge = (end_a >= begin_a + band);
        int64_t i = ge ? (int64_t)end_a - (int64_t)(begin_a + band) : 0;
        int64_t j = ge ? (int64_t)(2 * band) : (int64_t)(2 * band - (begin_a + band - end_a));
        /* [types] `i < x_size` and `i >= x_size-1-FORCE_MAXGAP_LEN` are unsigned comparisons */
        for (; (uint64_t)i < X && j >= 0; i++, j--) {
            if (!fe || ((uint64_t)i >= X - 1 - FORCE_MAXGAP && (uint64_t)i < X)) {
                int32_t v = HH(i, j);
                if (!found || v > best) { found = 1; mi = i; mj = j; best = v; }
            }
        }
    }
    if (!found) { /* :215 */
        free(H);
        out->cells = X * Y;
        return out->status; /* EMPTY */
    }

    /* traceback, :217-311; ops collected backwards */
    uint64_t cap = X + Y + 8, n = 0, nm = 0;
    uint8_t* rops = (uint8_t*)malloc(cap);
    int64_t x = mi, y = mj;
    int64_t pos = ba + x + y - w;
    int status = GAMDP_ORACLE_OK;
    while (x >= 0 && y >= 0 && pos >= 0) {
        if (pos >= la) { status = GAMDP_ORACLE_OUT_OF_RANGE; break; } /* a.at(pos) throws */
        uint8_t pa = a[pos], pb = b[begin_b + (uint64_t)x];
        int32_t s = score_of(pa, pb);
        int32_t h = HH(x, y);
        uint8_t mm = (pa == pb || pa == 4 || pb == 4) ? GAMDP_OP_MATCH : GAMDP_OP_MISMATCH;
        uint8_t op;
        if (pos == 0) {
            int left_ok = !(fs && x > FORCE_MAXGAP); /* left = -inf otherwise */
            if (h == s) { op = mm; x--; }
            else if (y == (int64_t)Y - 1 || (left_ok && h == G_GAP)) { op = GAMDP_OP_GAP_B; y--; }
            else { op = GAMDP_OP_GAP_A; x--; y++; }
        } else {
            int32_t dg = (x > 0 ? HH(x - 1, y) : 0) + s;
            int up_ok = 1;
            int32_t up = (x > 0 && y < (int64_t)Y - 1) ? HH(x - 1, y + 1) + G_GAP : G_GAP;
            if (fs && x == 0) {
                if (pos <= FORCE_MAXGAP) up = G_GAP;
                else up_ok = 0; /* -inf */
            }
            if (h == dg) { op = mm; x--; }
            else if (y < (int64_t)Y - 1 && y > 0 && up_ok && h == up) { op = GAMDP_OP_GAP_A; x--; y++; }
            else if (y < (int64_t)Y - 1 && y > 0) { op = GAMDP_OP_GAP_B; y--; }
            else if (y < (int64_t)Y - 1) { op = GAMDP_OP_GAP_A; x--; y++; }
            else { op = GAMDP_OP_GAP_B; y--; }
        }
        if (op == GAMDP_OP_MATCH) nm++;
        if (n == cap) { cap *= 2; rops = (uint8_t*)realloc(rops, cap); }
        rops[n++] = op;
        pos = ba + x + y - w;
    }
    free(H);
    if (status != GAMDP_ORACLE_OK) {
        free(rops);
        memset(out, 0, sizeof(*out));
        out->cells = X * Y;
        out->status = (uint8_t)status;
        return status;
    }
    for (uint64_t k = 0; k < n / 2; k++) { uint8_t t = rops[k]; rops[k] = rops[n - 1 - k]; rops[n - 1 - k] = t; }

    /* :319-321 */
    out->begin_a = (uint64_t)(pos + 1);
    out->begin_b = (uint64_t)((int64_t)begin_b + x + 1);
    out->score = best;
    out->n_match = nm;
    out->length = n;
    out->homology = n ? (double)(nm * 100) / (double)n : 0.0;
    out->status = GAMDP_ORACLE_OK;
    match_positions(rops, n, out);
    if (ops_out) memcpy(ops_out, rops, n < ops_cap ? n : ops_cap);
    free(rops);
    return GAMDP_ORACLE_OK;
#undef HH
}

/* ---- ABlast::findHits --------------------------------------------------------------------- */

typedef struct { uint64_t code; uint64_t pos; } kmer_t;

static int kmer_cmp(const void* p, const void* q)
{
    const kmer_t* x = (const kmer_t*)p;
    const kmer_t* y = (const kmer_t*)q;
    if (x->code != y->code) return x->code < y->code ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

/* sequence_code, ablast.hpp:53-59: code = (LAST_BASE-1)*code + base, i.e. base 4 with digits 0..4 */
static uint64_t kcode(const uint8_t* s, uint64_t p, uint64_t k)
{
    uint64_t c = 0;
    for (uint64_t i = p; i < p + k; i++) c = 4 * c + s[i];
    return c;
}

int64_t gamdp_oracle_find_hits(const uint8_t* a, uint64_t alen, uint64_t a_start, uint64_t a_end,
                               const uint8_t* b, uint64_t blen, uint64_t b_start, uint64_t b_end,
                               uint64_t word, uint32_t* hits, uint64_t cap)
{
    /* ablast.cc:47-53 */
    if (alen == 0 || blen == 0) return 0;
    if (a_end >= alen) a_end = alen - 1;
    if (b_end >= blen) b_end = blen - 1;
    if (a_start > a_end || b_start > b_end) return 0;
    if (a_end + 1 < word + a_start || b_end + 1 < word + b_start) return 0;

    /* build_hash, ablast.hpp:61-69: positions per code in ascending order */
    uint64_t na = a_end - word + 1 - a_start + 1;
    kmer_t* idx = (kmer_t*)malloc(na * sizeof(kmer_t));
    for (uint64_t i = 0; i < na; i++) { idx[i].code = kcode(a, a_start + i, word); idx[i].pos = a_start + i; }
    qsort(idx, na, sizeof(kmer_t), kmer_cmp);

    /* build_corrispondences_vector + mark_found, ablast.hpp:71-107 */
    uint64_t nf = a_end - a_start + 1;
    uint64_t* f = (uint64_t*)calloc(nf, sizeof(uint64_t));
    for (uint64_t bp = b_start; bp <= b_end - word + 1; bp++) {
        uint64_t c = kcode(b, bp, word);
        uint64_t lo = 0, hi = na;
        while (lo < hi) { uint64_t m = (lo + hi) / 2; if (idx[m].code < c) lo = m + 1; else hi = m; }
        for (; lo < na && idx[lo].code == c; lo++) {
            uint64_t ia = idx[lo].pos - a_start, ib = bp - b_start;
            if (ia >= ib) f[ia - ib]++;
        }
    }
    free(idx);

    /* ablast.cc:57-73 */
    uint64_t best = 0;
    for (uint64_t i = 0; i < nf; i++) if (f[i] > best) best = f[i];
    int64_t nh = 0;
    if (best > 0)
        for (uint64_t i = 0; i < nf; i++)
            if (f[i] == best) { if ((uint64_t)nh < cap && hits) hits[nh] = (uint32_t)(a_start + i); nh++; }
    free(f);
    return nh;
}

/* ---- L1 driver ---------------------------------------------------------------------------- */

typedef struct {
    const uint8_t* master; uint64_t mlen;
    const uint8_t* slave;  uint64_t slen;   /* current orientation */
    uint64_t band;
    gamdp_oracle_result* audit; uint32_t audit_cap; uint32_t n_dp; uint64_t cells;
} l1_ctx;

static int l1_dp(l1_ctx* c, const uint8_t* a, uint64_t alen, uint64_t ba, uint64_t ea, const uint8_t* b,
                 uint64_t blen, uint64_t bb, uint64_t eb, int fs, int fe, gamdp_oracle_result* r)
{
    int st = gamdp_oracle_align(a, alen, b, blen, c->band, ba, ea, bb, eb, fs, fe, r, NULL, 0);
    if (c->audit && c->n_dp < c->audit_cap) c->audit[c->n_dp] = *r;
    c->n_dp++;
    c->cells += r->cells;
    return st;
}

static inline int32_t frame_len(int32_t b, int32_t e) { return e < b ? 0 : e - b + 1; } /* Frame.cc:124-127 */

/* PctgBuilder::alignBlocks, PctgBuilder.cc:1617-1708.  Returns status of the first failing DP. */
static int l1_align_blocks(l1_ctx* c, uint64_t m_start, uint64_t s_start, const gamdp_oracle_block* bl,
                           uint32_t n, gamdp_oracle_result* res)
{
    int forward = bl[0].m_begin <= bl[n - 1].m_begin; /* :1650 */
    int64_t ms = (int64_t)m_start, ss = (int64_t)s_start;
    uint64_t last_a = 0, last_b = 0;
    const gamdp_oracle_block* prev = NULL;
    for (uint32_t k = 0; k < n; k++) {
        const gamdp_oracle_block* cur = forward ? &bl[k] : &bl[n - 1 - k];
        int32_t mlen = frame_len(cur->m_begin, cur->m_end), slen = frame_len(cur->s_begin, cur->s_end);
        if (k > 0) { /* :1660-1667 */
            int32_t mgap = prev->m_begin <= cur->m_begin ? (cur->m_begin - prev->m_end - 1) : (prev->m_begin - cur->m_end - 1);
            int32_t sgap = prev->s_begin <= cur->s_begin ? (cur->s_begin - prev->s_end - 1) : (prev->s_begin - cur->s_end - 1);
            ms = (int64_t)(last_a + (uint64_t)(int64_t)mgap); if (ms < 0) ms = 0;
            ss = (int64_t)(last_b + (uint64_t)(int64_t)sgap); if (ss < 0) ss = 0;
        }
        int st = l1_dp(c, c->master, c->mlen, (uint64_t)ms, (uint64_t)(ms + mlen - 1), c->slave, c->slen,
                       (uint64_t)ss, (uint64_t)(ss + slen - 1), 0, 0, &res[k]);
        if (st == GAMDP_ORACLE_OUT_OF_RANGE || st == GAMDP_ORACLE_INVALID) return st;
        last_a = res[k].last_a; /* last_match_pos; (0,0) for an empty alignment */
        last_b = res[k].last_b;
        prev = cur;
    }
    return GAMDP_ORACLE_OK;
}

/* is_good(vector), :1711-1724 */
static int l1_good_vec(const gamdp_oracle_result* r, uint32_t n, uint64_t min_len)
{
    uint64_t len = 0;
    for (uint32_t i = 0; i < n; i++) { if (r[i].homology < MIN_HOMOLOGY) return 0; len += r[i].length; }
    return len >= min_len;
}
/* is_good(single), :1727-1730 */
static int l1_good_one(const gamdp_oracle_result* r, uint64_t min_len)
{
    return r->homology >= MIN_HOMOLOGY && r->length >= min_len;
}

static inline uint64_t umin(uint64_t x, uint64_t y) { return x < y ? x : y; }

int gamdp_oracle_align_merge_block(const uint8_t* master, uint64_t mlen, const uint8_t* slave_fwd,
                                   uint64_t slen, const gamdp_oracle_block* bl, uint32_t n,
                                   uint64_t band, gamdp_oracle_mb* mb, gamdp_oracle_result* audit,
                                   uint32_t audit_cap)
{
    mb->align_ok = 1; /* :757 */
    mb->align_rev = 0; mb->touched = 0; mb->status = GAMDP_ORACLE_OK;
    mb->m_start = mb->m_end = mb->s_start = mb->s_end = 0; mb->n_dp = 0; mb->cells = 0;
    if (n == 0) { mb->status = GAMDP_ORACLE_INVALID; mb->align_ok = 0; return mb->status; } /* front() of empty list: UB */

    int status = GAMDP_ORACLE_OK;
    uint8_t* slave_rc = (uint8_t*)malloc(slen ? slen : 1);
    memcpy(slave_rc, slave_fwd, slen);
    gamdp_oracle_revcomp(slave_rc, slen);
    gamdp_oracle_result* A = (gamdp_oracle_result*)calloc(n, sizeof(*A));
    l1_ctx c = {master, mlen, slave_fwd, slen, band, audit, audit_cap, 0, 0};

    /* alignMergeBlock :733-744: region from first & last block frames only */
    const gamdp_oracle_block *fb = &bl[0], *lb = &bl[n - 1];
    uint64_t m_start = (uint64_t)(int64_t)(fb->m_begin < lb->m_begin ? fb->m_begin : lb->m_begin);
    uint64_t s_start = (uint64_t)(int64_t)(fb->s_begin < lb->s_begin ? fb->s_begin : lb->s_begin);
    uint64_t s_end = (uint64_t)(int64_t)(fb->s_end > lb->s_end ? fb->s_end : lb->s_end);

    /* findBestAlignment :1380-1408 */
    uint64_t con = 0, dis = 0;
    int32_t min_frame_len = 100;
    for (uint32_t k = 0; k < n; k++) {
        int32_t ml = frame_len(bl[k].m_begin, bl[k].m_end), sl = frame_len(bl[k].s_begin, bl[k].s_end);
        int32_t mn = ml < sl ? ml : sl;
        if (k == 0 || min_frame_len > mn) min_frame_len = mn;
        if (bl[k].m_strand != bl[k].s_strand) dis += (uint64_t)bl[k].n_reads; else con += (uint64_t)bl[k].n_reads;
    }
    double con_prob = (double)con / (double)(con + dis);
    uint64_t mt = (uint64_t)(0.3 * (double)mlen), st = (uint64_t)(0.3 * (double)slen);
    int32_t align_thr = (int32_t)(0.7 * min_frame_len);
    int32_t thr = (int32_t)umin(200, umin(mt, st));
    uint64_t align_thr_u = (uint64_t)(int64_t)align_thr;

    int good = 0, rev = 0;
    /* :1420-1509 -- orientation attempts; the slave is "reversed in place" = switch the view */
    for (int attempt = 0; attempt < 2 && !good && status == GAMDP_ORACLE_OK; attempt++) {
        int try_rev;
        if (con_prob >= 0.5) try_rev = attempt;          /* fwd, then rev */
        else if (con_prob < 0.5) try_rev = 1 - attempt;  /* rev, then fwd */
        else break;                                      /* NaN: neither branch runs */
        /* every reversal maps (start,end) -> (|s|-end-1, |s|-start-1); two reversals restore */
        uint64_t ss = try_rev ? slen - s_end - 1 : s_start;
        c.slave = try_rev ? slave_rc : slave_fwd;
        status = l1_align_blocks(&c, m_start, ss, bl, n, A);
        if (status != GAMDP_ORACLE_OK) break;
        if (l1_good_vec(A, n, align_thr_u)) { good = 1; rev = try_rev; }
    }
    if (status != GAMDP_ORACLE_OK) goto done;

    if (!good) { mb->align_ok = 0; goto done; } /* :1512 + :825-829: coords untouched */

    {
        /* :1515-1526 */
        uint64_t sa = A[0].first_a, sb = A[0].first_b;
        uint64_t ea = A[n - 1].last_a, eb = A[n - 1].last_b;
        uint64_t i1 = sa, i2 = mlen - ea - 1, j1 = sb, j2 = slen - eb - 1;
        const uint8_t* sl = rev ? slave_rc : slave_fwd;
        c.slave = sl;

        gamdp_oracle_result left, right; /* MyAlignment(100): homology 100, everything else 0 */
        memset(&left, 0, sizeof(left)); memset(&right, 0, sizeof(right));
        left.homology = 100.0; right.homology = 100.0;
        int left_rev = 0, right_rev = 0; /* uninitialised in the reference when the tail is skipped */
        uint64_t thr_u = (uint64_t)(int64_t)thr;

        if (!(umin(i1, j1) < thr_u && umin(i2, j2) < thr_u)) {
            uint32_t* hits = (uint32_t*)malloc(sizeof(uint32_t) * (mlen + slen + 1));
            if (umin(i1, j1) >= thr_u) { /* LEFT, :1535-1569 */
                if (i1 < j1) {
                    int64_t nh = gamdp_oracle_find_hits(sl, slen, 0, sb - 1, master, mlen, 0, sa - 1, 20, hits, mlen + slen);
                    uint64_t ba = nh > 0 ? hits[nh - 1] : sb - sa;
                    status = l1_dp(&c, sl, slen, ba, sb - 1, master, mlen, 0, sa - 1, 0, 1, &left);
                    left_rev = 1;
                } else {
                    int64_t nh = gamdp_oracle_find_hits(master, mlen, 0, sa - 1, sl, slen, 0, sb - 1, 20, hits, mlen + slen);
                    uint64_t ba = nh > 0 ? hits[nh - 1] : sa - sb;
                    status = l1_dp(&c, master, mlen, ba, sa - 1, sl, slen, 0, sb - 1, 0, 1, &left);
                    left_rev = 0;
                }
            }
            if ((status == GAMDP_ORACLE_OK || status == GAMDP_ORACLE_EMPTY) && umin(i2, j2) >= thr_u) { /* RIGHT, :1573-1611 */
                status = GAMDP_ORACLE_OK;
                if (i2 < j2) {
                    if (slen <= eb + 1) { status = GAMDP_ORACLE_OUT_OF_RANGE; } /* chop_borders throws */
                    else {
                        const uint8_t* T = sl + eb + 1; uint64_t tl = slen - (eb + 1);
                        int64_t nh = gamdp_oracle_find_hits(T, tl, 0, tl - 1, master, mlen, ea + 1, mlen - 1, 20, hits, mlen + slen);
                        uint64_t ba = nh > 0 ? hits[0] : 0;
                        status = l1_dp(&c, T, tl, ba, tl - 1, master, mlen, ea + 1, mlen - 1, 1, 0, &right);
                        right_rev = 1;
                    }
                } else {
                    if (mlen <= ea + 1) { status = GAMDP_ORACLE_OUT_OF_RANGE; }
                    else {
                        const uint8_t* T = master + ea + 1; uint64_t tl = mlen - (ea + 1);
                        int64_t nh = gamdp_oracle_find_hits(T, tl, 0, tl - 1, sl, slen, eb + 1, slen - 1, 20, hits, mlen + slen);
                        uint64_t ba = nh > 0 ? hits[0] : 0;
                        status = l1_dp(&c, T, tl, ba, tl - 1, sl, slen, eb + 1, slen - 1, 1, 0, &right);
                        right_rev = 0;
                    }
                }
            }
            free(hits);
            if (status == GAMDP_ORACLE_EMPTY) status = GAMDP_ORACLE_OK;
            if (status != GAMDP_ORACLE_OK) goto done;
        }

        /* alignMergeBlock :759-843; main_homology() >= 95 is implied by `good` */
        uint64_t thr2 = umin(100, umin(mt, st));
        uint64_t left_min = (uint64_t)(0.7 * (double)umin(i1, j1));
        uint64_t right_min = (uint64_t)(0.7 * (double)umin(i2, j2));
        int s_lt = rev ? mb->s_rtail : mb->s_ltail;
        int s_rt = rev ? mb->s_ltail : mb->s_rtail;
        if (mb->m_ltail && s_lt && umin(i1, j1) >= thr2) {
            if (l1_good_one(&left, left_min)) {
                sa = left.first_a; sb = left.first_b;
                if (left_rev) { uint64_t t = sa; sa = sb; sb = t; }
            } else mb->align_ok = 0;
        }
        if (mb->m_rtail && s_rt && umin(i2, j2) >= thr2) {
            if (l1_good_one(&right, right_min)) {
                uint64_t ta = right.last_a, tb = right.last_b;
                if (right_rev) { uint64_t t = ta; ta = tb; tb = t; ea = ta; eb += tb + 1; }
                else { ea += ta + 1; eb = tb; }
            } else mb->align_ok = 0;
        }
        if (rev) { uint64_t t = sb; sb = slen - eb - 1; eb = slen - t - 1; }
        mb->align_rev = (uint8_t)rev;
        mb->m_start = (int32_t)sa; mb->m_end = (int32_t)ea;
        mb->s_start = (int32_t)sb; mb->s_end = (int32_t)eb;
        mb->touched = 1;
    }

done:
    mb->n_dp = c.n_dp;
    mb->cells = c.cells;
    mb->status = (uint8_t)status;
    if (status != GAMDP_ORACLE_OK) mb->align_ok = 0;
    free(A);
    free(slave_rc);
    return status;
}

/* ---- synthetic workload + CPU baseline ---------------------------------------------------- */

static inline uint64_t sm64_next(uint64_t* s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

uint64_t gamdp_oracle_synth_pair(uint64_t k, uint64_t len, uint8_t* master, uint8_t* slave)
{
    uint64_t s = 0x47414DULL + k * 0xD1B54A32D192ED03ULL;
    (void)sm64_next(&s);
    for (uint64_t i = 0; i < len; i++) master[i] = (uint8_t)(sm64_next(&s) >> 62);
    const uint64_t T_DEL = 167772, T_SUB = 671088, T_INS = 167772; /* 1%, +3%, 1% of 2^24 */
    uint64_t n = 0;
    for (uint64_t i = 0; i < len; i++) {
        uint64_t r = sm64_next(&s);
        uint64_t u = r >> 40;
        if (u >= T_DEL) {
            uint8_t base = master[i];
            if (u < T_SUB) base = (uint8_t)((base + 1 + (r & 0xFFFF) % 3) & 3);
            slave[n++] = base;
        }
        uint64_t r2 = sm64_next(&s);
        if ((r2 >> 40) < T_INS) slave[n++] = (uint8_t)(r2 & 3);
    }
    return n;
}

typedef struct {
    uint64_t first, n, len, band;
    uint64_t cursor, cells;
    pthread_mutex_t mu;
    gamdp_oracle_result* results;
} bench_t;

static void* bench_worker(void* p)
{
    bench_t* B = (bench_t*)p;
    uint8_t* m = (uint8_t*)malloc(B->len);
    uint8_t* s = (uint8_t*)malloc(B->len + B->len / 8 + 64);
    uint64_t cells = 0;
    for (;;) {
        pthread_mutex_lock(&B->mu); /* extractNextPctg-style mutex-guarded cursor */
        uint64_t k = B->cursor++;
        pthread_mutex_unlock(&B->mu);
        if (k >= B->n) break;
        uint64_t sl = gamdp_oracle_synth_pair(B->first + k, B->len, m, s);
        gamdp_oracle_result r;
        gamdp_oracle_align(m, B->len, s, sl, B->band, 0, B->len - 1, 0, sl - 1, 0, 0, &r, NULL, 0);
        cells += r.cells;
        if (B->results) B->results[k] = r;
    }
    free(m); free(s);
    pthread_mutex_lock(&B->mu);
    B->cells += cells;
    pthread_mutex_unlock(&B->mu);
    return NULL;
}

uint64_t gamdp_oracle_bench_pairs(uint64_t first, uint64_t n, uint64_t len, uint64_t band, int threads,
                                  gamdp_oracle_result* results)
{
    bench_t B = {first, n, len, band, 0, 0, PTHREAD_MUTEX_INITIALIZER, results};
    if (threads < 1) threads = 1;
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, bench_worker, &B);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(th);
    return B.cells;
}
