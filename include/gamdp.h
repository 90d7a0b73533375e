/* gamdp.h -- C ABI of libgamdp: MI355X (gfx950) implementation of gam-merge's contig-pair
 * alignment hot path.  Plain C types only; no C++/torch types cross this boundary.
 *
 * What each entry point replaces in the reference (paths relative to the reference repo):
 *
 *   gamdp_align_batch          N independent calls of
 *                              BandedSmithWaterman(band).find_alignment(a,begin_a,end_a,b,begin_b,end_b,
 *                              force_start,force_end)        lib/include/alignment/banded_smith_waterman.hpp:66-71
 *                              (lib/src/alignment/banded_smith_waterman.cc:69-322) plus the
 *                              first_match_pos / last_match_pos scans of each result
 *                              (lib/src/alignment/my_alignment.cc:167-193, 228-262).
 *   gamdp_find_hits            ABlast(word).findHits(...)    lib/src/alignment/ablast.cc:41-76
 *   gamdp_align_merge_blocks   the loop `for list: for mb: builder.alignMergeBlock(graph,*mb)`
 *                              lib/src/pctg/BuildPctgFunctions.cc:82-84, i.e. N calls of
 *                              PctgBuilder::alignMergeBlock  lib/src/pctg/PctgBuilder.cc:726-844
 *                              (findBestAlignment :1361-1614, alignBlocks :1617-1708, is_good :1711-1730).
 *   gamdp_seqset_*             the RefSequence vectors filled by loadSequences
 *                              (lib/include/assembly/io_contig.code.hpp:568-596) -- uploaded once per GPU.
 *   gamdp_encode / _revcomp    Nucleotide(char) (lib/include/assembly/nucleotide.code.hpp:47-75) and
 *                              reverse_complement (lib/include/assembly/contig.code.hpp:187-229).
 *
 * Conventions: every function returns 0 on success or a negative GAMDP_E* code; nothing throws;
 * the caller owns every buffer it passes; a gamdp_ctx binds one GPU + one HIP stream and must be
 * used by one host thread at a time (one ctx per thread / per GPU).  There is NO CPU fallback:
 * with no usable gfx950 device gamdp_ctx_create fails with GAMDP_ENODEV.
 */
#ifndef GAMDP_H
#define GAMDP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAMDP_VERSION 1

/* error codes (negative return values) */
#define GAMDP_EINVAL  (-1)  /* bad argument                                        */
#define GAMDP_ENODEV  (-2)  /* no usable GPU / HIP runtime failure at context setup */
#define GAMDP_ENOMEM  (-3)  /* host or device allocation failed                    */
#define GAMDP_ENOTSUP (-4)  /* band wider than GAMDP_MAX_BAND                          */
#define GAMDP_EHIP    (-5)  /* a HIP call failed at run time (see gamdp_last_error) */

/* per-task status, mirrors the reference's observable outcomes */
#define GAMDP_ST_OK           0 /* a MyAlignment was produced                                          */
#define GAMDP_ST_EMPTY        1 /* reference returns MyAlignment() (banded_smith_waterman.cc:90, :215) */
#define GAMDP_ST_OUT_OF_RANGE 2 /* reference throws std::out_of_range from Contig::at; gam-merge then
                                   drops the whole graph (lib/src/pctg/ThreadedBuildPctg.cc:322-329)   */
#define GAMDP_ST_INVALID      3 /* arguments for which the reference has undefined behaviour           */
#define GAMDP_ST_DIAG_RANGE   9 /* libgamdp_diag.so only (never the product library): a packed-f16 block of this task held
                                   a value outside the exactly representable range -- the assertion behind the range argument
                                   of DESIGN.md section 4; the parity campaigns run on that build and expect none           */

/* edit-string alphabet (lib/include/alignment/my_alignment.hpp:57-62) */
#define GAMDP_OP_GAP_A 0
#define GAMDP_OP_GAP_B 1
#define GAMDP_OP_MATCH 2
#define GAMDP_OP_MISMATCH 3

#define GAMDP_DEFAULT_BAND 150   /* DEFAULT_BAND_SIZE, banded_smith_waterman.hpp:38 */
/* The reference takes any band (banded_smith_waterman.hpp:66).  So does this library, up to GAMDP_MAX_BAND = 2^20 (beyond it the
 * reference's own index arithmetic is untested and the oracle declines): bands up to GAMDP_MAX_TUNED_BAND run on the systolic
 * kernels (2*band+1 <= 64 lanes x 17 columns), wider ones on a correct-at-any-speed kernel that keeps the whole band matrix
 * (x_size * (2*band+1) int32) in the scratch arena -- a call whose matrix does not fit the arena fails with GAMDP_ENOMEM. */
#define GAMDP_MAX_BAND 1048576
#define GAMDP_MAX_TUNED_BAND 543

typedef struct gamdp_ctx gamdp_ctx;
typedef struct gamdp_seqset gamdp_seqset;
typedef struct gamdp_fasta gamdp_fasta;

/* One find_alignment call.  Sequences are referenced by index into a seqset; *_rc selects the
 * reverse complement of that sequence (the reference reverse-complements the slave contig in
 * place, PctgBuilder.cc:1443, 1467); *_off makes the sequence a suffix view seq[off..] (the
 * reference's chop_begin copy, PctgBuilder.cc:1577, 1596).  Coordinates are relative to the view. */
typedef struct gamdp_task {
    uint32_t a_id, b_id;
    uint64_t a_off, b_off;
    uint8_t a_rc, b_rc, force_start, force_end;
    uint32_t band;
    uint64_t begin_a, end_a, begin_b, end_b;
} gamdp_task;

/* What the callers of find_alignment consume (PctgBuilder.cc:761-762, 787, 801, 1516-1517, 1672).
 * homology is (double)(n_match*100)/(double)length, or 0 when length==0, exactly as
 * banded_smith_waterman.cc:319; it is filled in on the host. */
typedef struct gamdp_result {
    uint64_t begin_a, begin_b;   /* MyAlignment::begin_a / begin_b                          */
    int64_t score;               /* MyAlignment::score                                      */
    uint64_t n_match, length;    /* #MATCH ops, #ops                                        */
    uint64_t first_a, first_b;   /* first_match_pos                                         */
    uint64_t last_a, last_b;     /* last_match_pos                                          */
    uint64_t cells;              /* x_size*y_size of the reference's fill loops (GCUPS unit) */
    double homology;
    uint8_t first_found, last_found; /* bool results of first/last_match_pos               */
    uint8_t status;              /* GAMDP_ST_*                                              */
    uint8_t pad_[5];
} gamdp_result;

/* Optional edit strings: ops[task i] occupies ops_buf[ops_off[i] .. ops_off[i]+length) in forward
 * order, truncated to ops_cap[i].  The caller fills ops_off/ops_cap; pass NULL to skip. */
typedef struct gamdp_ops {
    uint8_t* ops_buf;
    const uint64_t* ops_off;
    const uint64_t* ops_cap;
} gamdp_ops;

/* Block / Frame fields the merge-block driver reads (Frame::getBegin/getEnd/getStrand,
 * Block::getReadsNumber; lib/src/assembly/Frame.cc:100-127, Block.cc:79-82). */
typedef struct gamdp_block {
    int32_t m_begin, m_end, s_begin, s_end;
    char m_strand, s_strand;
    int64_t n_reads;
} gamdp_block;

/* MergeBlock (lib/include/pctg/MergeDescriptor.hpp:40-69): inputs the driver reads ... */
typedef struct gamdp_mb_in {
    int32_t m_id, s_id;                       /* indices into the master / slave seqsets */
    uint8_t m_ltail, m_rtail, s_ltail, s_rtail;
    uint32_t n_blocks;
    const gamdp_block* blocks;                /* graph.getBlocks(mb.vertex), in list order */
} gamdp_mb_in;

/* ... and the fields it writes. coords_set==0 means the reference returned before writing
 * align_rev/m_start/... (PctgBuilder.cc:825-829) so the caller must leave them untouched. */
typedef struct gamdp_mb_out {
    uint8_t align_ok, align_rev;
    uint8_t status;       /* GAMDP_ST_OK, or OUT_OF_RANGE/INVALID: the reference would throw -> drop graph */
    uint8_t coords_set;
    int32_t m_start, m_end, s_start, s_end;
    uint32_t n_dp;        /* find_alignment calls made for this merge block */
    uint64_t cells;       /* sum of their cells                              */
} gamdp_mb_out;

/* ---- context ------------------------------------------------------------------------------ */
int gamdp_ctx_create(int device, gamdp_ctx** out);
void gamdp_ctx_destroy(gamdp_ctx* ctx);
/* bound the scratch arena (direction matrix) in bytes; 0 = default (75 % of free HBM) */
int gamdp_ctx_set_arena_bytes(gamdp_ctx* ctx, uint64_t bytes);
const char* gamdp_last_error(const gamdp_ctx* ctx);
/* the hipStream_t the kernels run on, as an opaque pointer */
void* gamdp_ctx_stream(gamdp_ctx* ctx);
/* HIP-event timing of the DP kernel launches since the last reset: total ms and launch count.  Launches of one call that run side
 * by side (a batch's small N-aware launch beside its big launch, each on a stream of its own) count with the time the GPU was busy
 * with them -- the union of their intervals --, not with the sum of their durations. */
int gamdp_ctx_kernel_time(gamdp_ctx* ctx, double* total_ms, uint64_t* launches, int reset);
/* What the last gamdp_align_batch call on this context launched, in launch order (the pieces of a batch that went through in
 * pieces included): the library's own account of its launch planner's choices (which kernel instantiation a group of calls
 * took, how many wavefront slots, how many rounds) and of what the wavefronts then did on the device (counted there).
 * No reference counterpart: BandedSmithWaterman::find_alignment (banded_smith_waterman.cc:69) is one code path. */
typedef struct gamdp_launch_info {
    char kernel[40];                 /* the instantiation, e.g. "k_align_o<19,15>" */
    uint32_t n_aware;                /* 1: the instantiation handles N (v_dot8 cells) */
    uint32_t tasks_per_wavefront;    /* 1, 2, 4 or 8 */
    uint32_t tasks;                  /* calls in the launch (the copies that fill up its last wavefront not counted) */
    uint32_t units;                  /* wavefront units handed out: tasks / pairs / quads / octets */
    uint32_t slots;                  /* resident wavefronts = scratch slots */
    uint32_t band_max;               /* widest band among the calls */
    uint32_t units_dirfree;          /* units that ran a direction-free range (counted on the device) */
    uint32_t units_packed_top;       /* ... whose blocks with pos <= 0 cells ran packed (device) */
    uint32_t units_packed_top_mixed; /* ... of those, through the per-task form: calls that differ in begin_a / force_start calls (device) */
    uint32_t strips;                 /* strip re-creations (materialise calls) of the launch's walks (device) */
    uint32_t piece;                  /* piece of a batch that went through in pieces (0 = the whole batch / its first piece) */
    uint32_t units_top_wanted;       /* units with a packed range whose calls hold blocks with pos <= 0 cells behind the ramp: what packed top blocks are for (device) */
    double rounds;                   /* units / slots */
    double kernel_ms;                /* HIP events around the launch, on the stream it ran on (this launch alone: side-by-side launches overlap) */
} gamdp_launch_info;
/* copies up to `cap` records to out (may be NULL when cap == 0); *n = how many launches the call made */
int gamdp_ctx_launch_info(const gamdp_ctx* ctx, gamdp_launch_info* out, size_t cap, size_t* n);

/* ---- sequences ---------------------------------------------------------------------------- */
/* seqs[i] points to lens[i] bytes: ASCII bases when is_ascii!=0 (normalised like Nucleotide(char)),
 * else base codes A=0 T=1 C=2 G=3 N=4.  Packs to 2 bit + N mask and uploads to the ctx's GPU. */
int gamdp_seqset_create(gamdp_ctx* ctx, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n,
                        int is_ascii, gamdp_seqset** out);
void gamdp_seqset_destroy(gamdp_seqset* set);
uint32_t gamdp_seqset_size(const gamdp_seqset* set);
uint64_t gamdp_seqset_length(const gamdp_seqset* set, uint32_t id);

/* ---- FASTA loader (input side of the path; lib/include/assembly/io_contig.code.hpp:511-596) ------------- */
/* Reads a (multi-)FASTA file with the reference's rules: name = header up to the first blank, every character
 * other than newline / blank is a base through Nucleotide(char) (unknown -> N).  No GPU needed. */
int gamdp_fasta_open(const char* path, gamdp_fasta** out);
void gamdp_fasta_close(gamdp_fasta* f);
uint32_t gamdp_fasta_count(const gamdp_fasta* f);
const char* gamdp_fasta_name(const gamdp_fasta* f, uint32_t i);
const uint8_t* gamdp_fasta_codes(const gamdp_fasta* f, uint32_t i, uint64_t* len);
/* sequence i of the set = record i of the file */
int gamdp_seqset_create_from_fasta(gamdp_ctx* ctx, const gamdp_fasta* f, gamdp_seqset** out);

/* ---- .blocks files (gam-create's output, gam-merge's input; lib/src/assembly/Block.cc:669-690, 737-747, 795-807,
 *      Frame.cc:197-222) -------------------------------------------------------------------------------------- */
/* One line = numReads, then for the master and for the slave frame: 0 ctgId strand begin end blockReadsLen readsLen
 * (tab separated on output, any blanks on input).  Lines that are empty or start with '#' are skipped, lines that do
 * not parse are dropped silently, blocks with numReads < min_block_size are dropped (loadBlocks, :669-690). */
typedef struct gamdp_block_rec {
    int64_t n_reads;
    uint64_t m_block_reads_len, m_reads_len, s_block_reads_len, s_reads_len;
    int32_t m_ctg, m_begin, m_end, s_ctg, s_begin, s_end;
    char m_strand, s_strand;
    uint8_t pad_[6];
} gamdp_block_rec;
typedef struct gamdp_blocks gamdp_blocks;
int gamdp_blocks_open(const char* path, int64_t min_block_size, gamdp_blocks** out);
void gamdp_blocks_close(gamdp_blocks* b);
uint64_t gamdp_blocks_count(const gamdp_blocks* b);
const gamdp_block_rec* gamdp_blocks_data(const gamdp_blocks* b);
/* writeBlocks (:737-747): the header line, then one block per line */
int gamdp_blocks_write(const char* path, const gamdp_block_rec* recs, uint64_t n);

/* ---- L0: batch of independent banded alignments ------------------------------------------- */
/* a sequences come from set_a, b sequences from set_b (may be the same set). */
int gamdp_align_batch(gamdp_ctx* ctx, const gamdp_seqset* set_a, const gamdp_seqset* set_b,
                      const gamdp_task* tasks, size_t n, gamdp_result* out, const gamdp_ops* ops_or_null);

/* What gamdp_align_batch decides for one call before anything is launched, on plain numbers (view lengths alen / blen):
 * the checks of banded_smith_waterman.cc:90-132 in the reference's order.  Returns GAMDP_ST_OK when the call needs the
 * DP (it would be launched), otherwise the final status (EMPTY / OUT_OF_RANGE / INVALID) the batch call reports without
 * launching.  *cells (may be NULL) = x_size * y_size, the GCUPS unit and the weight the multi-GPU partitioner balances. */
int gamdp_task_preflight(uint64_t alen, uint64_t blen, uint32_t band, uint64_t begin_a, uint64_t end_a, uint64_t begin_b,
                         uint64_t end_b, int force_start, int force_end, uint64_t* cells);

/* bit 0: this library is the diagnostics build (-DGAMDP_DIAG; honours the GAMDP_DIAG_* switches that change kernel
 * paths or invalidate results).  The product build returns 0 and ignores those switches. */
unsigned gamdp_build_info(void);

/* ---- L1: batch of merge blocks ------------------------------------------------------------ */
/* band is DEFAULT_BAND_SIZE (150) in the reference.  audit (optional) receives, per merge block i,
 * up to audit_stride results of its DP calls in call order at audit[i*audit_stride ...]. */
int gamdp_align_merge_blocks(gamdp_ctx* ctx, const gamdp_seqset* master, const gamdp_seqset* slave,
                             const gamdp_mb_in* in, size_t n, uint32_t band, gamdp_mb_out* out,
                             gamdp_result* audit, uint32_t audit_stride);

/* What the last gamdp_align_merge_blocks call on this context did (timing of the driver itself). */
typedef struct gamdp_l1_stats {
    uint64_t merge_blocks, dp_calls, cells;
    uint32_t rounds;          /* most rounds any cohort needed (a round = one batch of pending find_alignment calls) */
    uint32_t cohorts;         /* host threads / streams the merge blocks were spread over on this device          */
    uint32_t launches;        /* kernel launches                                                                   */
    uint32_t pad_;
    double wall_ms;           /* inside the call                                                                   */
    double gpu_busy_ms;       /* union of the kernels' execution intervals (HIP events): time with >= 1 kernel running */
    double kernel_sum_ms;     /* sum of the kernels' durations (can exceed gpu_busy_ms: cohorts overlap)          */
    double host_pending_ms;   /* building pending calls incl. findHits, summed over cohort threads                 */
    double host_feed_ms;      /* feeding results back, summed over cohort threads                                  */
} gamdp_l1_stats;
int gamdp_ctx_l1_stats(const gamdp_ctx* ctx, gamdp_l1_stats* out);

/* ---- several GPUs of one node ------------------------------------------------------------- */
/* Replaces gam-merge's worker pool (lib/src/pctg/ThreadedBuildPctg.cc:143-197: N pthreads, mutex-guarded cursor
 * :50-74) for this path: one host thread + context + resident copy of the sequences per device; the task / merge-block
 * list is partitioned statically (longest-processing-time first by predicted cell updates), every device fills the
 * result slots of its own items, and nothing is exchanged between devices (no collective).  Results are identical to
 * the single-context calls whatever the number of devices.  A device may be listed more than once (one context each). */
typedef struct gamdp_multi gamdp_multi;
typedef struct gamdp_multi_seqset gamdp_multi_seqset;
int gamdp_multi_create(const int* devices, int n, gamdp_multi** out);
void gamdp_multi_destroy(gamdp_multi* m);
int gamdp_multi_size(const gamdp_multi* m);
gamdp_ctx* gamdp_multi_ctx(gamdp_multi* m, int i);            /* borrowed: context of device slot i */
const char* gamdp_multi_last_error(const gamdp_multi* m);
/* the same sequences resident on every device (RefSequence is shared read-only by the reference's workers) */
int gamdp_multi_seqset_create(gamdp_multi* m, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n, int is_ascii,
                              gamdp_multi_seqset** out);
int gamdp_multi_seqset_create_from_fasta(gamdp_multi* m, const gamdp_fasta* f, gamdp_multi_seqset** out);
void gamdp_multi_seqset_destroy(gamdp_multi_seqset* s);
gamdp_seqset* gamdp_multi_seqset_on(gamdp_multi_seqset* s, int i);   /* borrowed: the copy on device slot i */
/* gamdp_align_batch / gamdp_align_merge_blocks over all devices of m; out / audit are indexed like the input */
int gamdp_multi_align_batch(gamdp_multi* m, const gamdp_multi_seqset* set_a, const gamdp_multi_seqset* set_b,
                            const gamdp_task* tasks, size_t n, gamdp_result* out);
int gamdp_multi_align_merge_blocks(gamdp_multi* m, const gamdp_multi_seqset* master, const gamdp_multi_seqset* slave,
                                   const gamdp_mb_in* in, size_t n, uint32_t band, gamdp_mb_out* out, gamdp_result* audit,
                                   uint32_t audit_stride);
/* The partitioner on its own (host only): items in order of decreasing weight, ties by index, go to the least-loaded
 * part, ties to the lower part.  Deterministic: the ranks of a one-process-per-GPU run (bench.py under
 * torch.distributed.run) each derive the same assignment from it without communicating. */
int gamdp_partition_lpt(const uint64_t* weights, size_t n, int parts, uint32_t* part_of);

/* ---- host-side helpers on the path -------------------------------------------------------- */
/* ABlast::findHits on code arrays; returns the number of hits (first cap written) or <0. */
int64_t gamdp_find_hits(const uint8_t* a, uint64_t alen, uint64_t a_start, uint64_t a_end,
                        const uint8_t* b, uint64_t blen, uint64_t b_start, uint64_t b_end,
                        uint64_t word, uint32_t* hits, uint64_t cap);
void gamdp_encode(const char* chars, uint64_t n, uint8_t* codes);
void gamdp_decode(const uint8_t* codes, uint64_t n, char* chars);
void gamdp_revcomp(uint8_t* codes, uint64_t n);

/* ---- post-alignment stage: merge lists -> paired contigs -> .gam.fasta / .pctgs ----- */
/* Host only (no GPU, no ctx).  Replaces, for one assembly graph, BuildPctgFunctions.cc:86-92 after the align loop:
 * splitMergeBlocksByAlign / ByDirection / sortMergeBlocksByDirection / splitMergeBlocksByInclusions
 * (PctgBuilder.cc:667-723, 543-665, 507-541, 291-505) and buildPctgs (PctgBuilder.cc:172-288); and, for the whole
 * run, the id assignment, the single-contig pctgs and the two writers of src/Merge.cc:380-385, 437-465. */

/* a host-side assembly from memory (what gamdp_fasta_open builds from a file): names[i] may be NULL */
int gamdp_fasta_create(const char* const* names, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n,
                       int is_ascii, gamdp_fasta** out);

/* MergeBlock as the post-alignment stage sees it (MergeDescriptor.hpp:40-69 without vertex/valid). */
typedef struct gamdp_mblock {
    int32_t m_id, m_start, m_end;
    int32_t s_id, s_start, s_end;
    uint8_t align_rev, align_ok;
    uint8_t m_ltail, m_rtail, s_ltail, s_rtail;
    uint8_t ext_slave_next, ext_slave_prev;   /* carried and updated like the reference; nothing reads them */
    uint8_t m_rev, s_rev;                     /* written by the direction stage */
    uint8_t pad_[2];
} gamdp_mblock;

/* CtgInPctgInfo (lib/include/pctg/CtgInPctgInfo.hpp:44-69): one row of a paired contig's merge list */
typedef struct gamdp_pctg_row {
    int64_t start, end;        /* first / last base taken from the contig, on the strand it is used on */
    int32_t ctg_id;
    uint8_t reversed, is_master;
    uint8_t pad_[2];
} gamdp_pctg_row;

#define GAMDP_STAGE_ALIGN      1u   /* splitMergeBlocksByAlign      */
#define GAMDP_STAGE_DIRECTION  2u   /* splitMergeBlocksByDirection  */
#define GAMDP_STAGE_SORT       4u   /* sortMergeBlocksByDirection   */
#define GAMDP_STAGE_INCLUSIONS 8u   /* splitMergeBlocksByInclusions */
#define GAMDP_STAGE_ALL        15u

/* The list surgery alone (stages applied in the reference's order, selected by the mask).  Lists are passed flat:
 * blocks holds list 0, then list 1, ...; list_sizes their lengths.  Returns GAMDP_ENOMEM (with *n_out_lists set)
 * when the output does not fit. */
int gamdp_merge_lists_prepare(const gamdp_fasta* master, const gamdp_fasta* slave, const gamdp_mblock* blocks,
                              const uint32_t* list_sizes, uint32_t n_lists, unsigned stages, gamdp_mblock* out_blocks,
                              uint64_t cap_blocks, uint32_t* out_sizes, uint32_t cap_lists, uint32_t* n_out_lists);

/* appendBlocksRegionToPctg asks the BAMs which assembly to trust when the two copies of a block region differ in
 * length by more than 3 % (computeZScore, PctgBuilder.cc:147-168).  The host supplies that decision:
 * return 0 = take the master's copy, 1 = the slave's, < 0 = error (the graph is dropped).  Coordinates are the ones
 * the reference passes (strand coordinates of the merge block). */
typedef int (*gamdp_region_vote_fn)(void* user, int32_t m_id, int32_t m_start, int32_t m_end, int32_t s_id,
                                    int32_t s_start, int32_t s_end);
/* the evidence count of PctgBuilder.cc:155-168 on two z-score vectors: 0 = master, 1 = slave */
int gamdp_zscore_vote(const double* master_z, const double* slave_z, size_t n);

typedef struct gamdp_pctgs gamdp_pctgs;   /* std::list<PairedContig> of one gam-merge run */
int gamdp_pctgs_create(const gamdp_fasta* master, const gamdp_fasta* slave, gamdp_pctgs** out);
void gamdp_pctgs_destroy(gamdp_pctgs* p);
const char* gamdp_pctgs_last_error(const gamdp_pctgs* p);
/* One graph's merge lists (after gamdp_align_merge_blocks filled align_ok/align_rev/m_start..s_end): list surgery +
 * buildPctgs; the resulting paired contigs are appended in list order.  On error nothing of this graph is kept
 * (the reference drops a graph whose worker throws, ThreadedBuildPctg.cc:322-329). */
int gamdp_pctgs_add_graph(gamdp_pctgs* p, const gamdp_mblock* blocks, const uint32_t* list_sizes, uint32_t n_lists,
                          gamdp_region_vote_fn vote, void* user);
/* ids 0.. in insertion order, then one paired contig per master contig no paired contig uses (ascending id, empty
 * contigs skipped): src/Merge.cc:380-385, 437-452, BuildPctgFunctions.cc:111-129 */
int gamdp_pctgs_finish(gamdp_pctgs* p);
uint32_t gamdp_pctgs_count(const gamdp_pctgs* p);
uint32_t gamdp_pctgs_merged_count(const gamdp_pctgs* p);   /* how many came from merge lists (old_pctg_id) */
const uint8_t* gamdp_pctgs_codes(const gamdp_pctgs* p, uint32_t i, uint64_t* len);
/* merge list of paired contig i: returns its length, writes the first cap rows */
uint32_t gamdp_pctgs_rows(const gamdp_pctgs* p, uint32_t i, gamdp_pctg_row* out, uint32_t cap);
/* which contigs any paired contig touches (getMasterCtgIdSet / getSlaveIds; Merge.cc:418-424 needs the slave side
 * for .notmerged.fasta); either pointer may be NULL */
int gamdp_pctgs_contig_use(const gamdp_pctgs* p, uint8_t* master_used, uint8_t* slave_used);
/* ".gam.fasta": >PairedContig_<id>, 60 bases per line (io_contig.code.hpp:246-262 + the endl of Merge.cc:458) */
int gamdp_pctgs_write_fasta(const gamdp_pctgs* p, const char* path);
/* ".pctgs" (PairedContig.cc:305-349) */
int gamdp_pctgs_write_descriptors(const gamdp_pctgs* p, const char* path);

/* ---- gam-merge's side outputs: slave contigs no block lies on, slave contigs no paired contig uses -------------
 * (src/Merge.cc:273-277, 294-297, 335-373, 412-431; host only).  All flag arrays are one byte per contig. */
/* getNoBlocksContigs (Block.cc:810-862): master_nbc[i] / slave_nbc[i] = 1 iff no block lies on contig i.  A block with a
 * contig id outside [0, n_master) / [0, n_slave) is where the reference prints an error and exits: GAMDP_EINVAL. */
int gamdp_no_blocks_contigs(const gamdp_block_rec* blocks, uint64_t n_blocks, uint32_t n_master, uint32_t n_slave,
                            uint8_t* master_nbc, uint8_t* slave_nbc);
/* getNoBlocksAfterFilterContigs (Block.cc:865-925): contigs that had blocks before the coverage filter
 * (master_nbc / slave_nbc = the arrays of the call above on the unfiltered list) and have none in `filtered`. */
int gamdp_no_blocks_after_filter(const gamdp_block_rec* filtered, uint64_t n_blocks, uint32_t n_master, uint32_t n_slave,
                                 const uint8_t* master_nbc, const uint8_t* slave_nbc, uint8_t* master_af, uint8_t* slave_af);
/* Merge.cc:416-429: not_merged[i] = 1 iff slave contig i is in no paired contig and in neither no-blocks set */
int gamdp_pctgs_not_merged(const gamdp_pctgs* p, const uint8_t* slave_nbc_bf, const uint8_t* slave_nbc_af, uint8_t* not_merged);
/* the contigs with select[i] != 0, each as `os << contig << std::endl` (io_contig.code.hpp:246-262: ">" name, 60 bases
 * per line): ".noblocks.BF.fasta", ".noblocks.AF.fasta", ".notmerged.fasta" (Merge.cc:336-373, 414-431) */
int gamdp_fasta_write_selected(const gamdp_fasta* f, const uint8_t* select, const char* path);

/* Synthetic pair k of the benchmark workload (BASELINE.json config 5): master = len uniform ACGT
 * codes, slave = master with 3 % substitutions, 1 % insertions, 1 % deletions (splitmix64 keyed by
 * k).  slave must hold len + len/8 + 64 codes; returns the slave length. */
uint64_t gamdp_synth_pair(uint64_t k, uint64_t len, uint8_t* master, uint8_t* slave);
/* Benchmark helper: a sequence set holding synthetic pairs [first_pair, first_pair+n_pairs) (sequence 2k =
 * master, 2k+1 = slave of pair first_pair+k), generated on host threads straight into the packed planes.
 * Such a set keeps no host copy of the bases: reverse-complement views and merge blocks are refused on it. */
int gamdp_seqset_create_synth(gamdp_ctx* ctx, uint64_t first_pair, uint32_t n_pairs, uint64_t len,
                              gamdp_seqset** out);
/* the same for pairs first_pair + k * stride_pairs, k < n_pairs: the share of one GPU when the fixed pair list of the
 * benchmark is dealt round-robin over several (what gamdp_partition_lpt yields for equal weights) */
int gamdp_seqset_create_synth_strided(gamdp_ctx* ctx, uint64_t first_pair, uint64_t stride_pairs, uint32_t n_pairs,
                                      uint64_t len, gamdp_seqset** out);

#ifdef __cplusplus
}
#endif
#endif /* GAMDP_H */
