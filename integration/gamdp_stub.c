/* Stand-in for libgamdp.so on a host WITHOUT an MI355X / hipcc, for ONE purpose: building the patched gam-merge so that
 * its own CPU path can write the golden-vector dump (GAMDP_DUMP_PREFIX, integration/make_l1_dump.sh).  It exports the
 * entry points integration/GamdpBridge.cc binds (include/gamdp.h) and refuses every one of them with GAMDP_ENODEV, which
 * makes the bridge report "using the CPU alignment" -- gam-merge then runs exactly the reference's code.  Never shipped,
 * never used by the product or its tests: the real library is gam_ngs_amd/libgamdp.so. */
#include "gamdp.h"

int gamdp_multi_create(const int* devices, int n, gamdp_multi** out) { (void)devices; (void)n; if (out) *out = 0; return GAMDP_ENODEV; }
void gamdp_multi_destroy(gamdp_multi* m) { (void)m; }
const char* gamdp_multi_last_error(const gamdp_multi* m) { (void)m; return "gamdp stub library: no GPU implementation on this host"; }
int gamdp_multi_seqset_create(gamdp_multi* m, const uint8_t* const* seqs, const uint64_t* lens, uint32_t n, int is_ascii, gamdp_multi_seqset** out)
{ (void)m; (void)seqs; (void)lens; (void)n; (void)is_ascii; if (out) *out = 0; return GAMDP_ENODEV; }
void gamdp_multi_seqset_destroy(gamdp_multi_seqset* s) { (void)s; }
int gamdp_multi_align_merge_blocks(gamdp_multi* m, const gamdp_multi_seqset* master, const gamdp_multi_seqset* slave, const gamdp_mb_in* in,
                                   size_t n, uint32_t band, gamdp_mb_out* out, gamdp_result* audit, uint32_t audit_stride)
{ (void)m; (void)master; (void)slave; (void)in; (void)n; (void)band; (void)out; (void)audit; (void)audit_stride; return GAMDP_ENODEV; }
