/* GamdpBridge -- the only place where gam-merge talks to libgamdp (the MI355X implementation of its contig-pair
 * alignment, C ABI in gamdp.h).  Deliberately free of GAM-NGS / Boost types: the callers in BuildPctgFunctions.cc and
 * ThreadedBuildPctg.cc translate MergeBlock / Block / RefSequence to the plain records below, so this file compiles
 * against gamdp.h alone.
 *
 *   GAMDP_DEVICES=0            one MI355X            (unset: the bridge is off and gam-merge runs its CPU path)
 *   GAMDP_DEVICES=0,1,...,7    the 8 GPUs of a node  (static partition of the merge blocks, no collective)
 *   GAMDP_DUMP_PREFIX=path     write path.mergeblocks.tsv (what alignMergeBlock reads) and path.mergeblocks.out.tsv
 *                              (what it wrote) -- works on the CPU path too: that is how golden vectors for the
 *                              merge-block driver are produced on a host without a GPU.
 */
#ifndef GAMDPBRIDGE_HPP
#define GAMDPBRIDGE_HPP

#include <stdint.h>

#include <string>
#include <vector>

#include "gamdp.h"

namespace gamdp_bridge
{

/* one MergeBlock (MergeDescriptor.hpp:40-69) with the blocks of its graph vertex (graph.getBlocks(mb.vertex)) */
struct MergeBlockRec
{
	/* in */
	int32_t m_id, s_id;
	bool m_ltail, m_rtail, s_ltail, s_rtail;
	std::vector<gamdp_block> blocks;
	uint32_t graph, list;          /* position in graphs_list / in the graph's mergeLists (dump markers) */
	/* out: the fields alignMergeBlock writes (PctgBuilder.cc:757, 825-843) */
	bool align_ok, align_rev, coords_set;
	bool thrown;                   /* the reference would have thrown: the caller drops the whole graph */
	int32_t m_start, m_end, s_start, s_end;

	MergeBlockRec() : m_id(0), s_id(0), m_ltail(false), m_rtail(false), s_ltail(false), s_rtail(false), graph(0), list(0),
		align_ok(false), align_rev(false), coords_set(false), thrown(false), m_start(0), m_end(0), s_start(0), s_end(0) {}
};

/* true when GAMDP_DEVICES names at least one device */
bool enabled();

/* Upload both assemblies (base codes A=0 T=1 C=2 G=3 N=4, one vector per contig, in RefSequence order) to every
 * device of GAMDP_DEVICES.  Call once, after loadSequences (src/Merge.cc:315-324).  Returns false (and says why on
 * stderr) when the GPUs cannot be used; gam-merge then falls back to ITS OWN CPU path. */
bool init( const std::vector< std::vector<uint8_t> > &master, const std::vector< std::vector<uint8_t> > &slave );
bool ready();
void shutdown();

/* PctgBuilder::alignMergeBlock for the merge blocks of ALL graphs in one batch (band = DEFAULT_BAND_SIZE).
 * Returns false on a library error (message on stderr); records are then untouched. */
bool alignMergeBlocks( std::vector<MergeBlockRec> &mbs, uint32_t band );

/* GAMDP_DUMP_PREFIX: append inputs / outputs of these merge blocks (names = RefName of the contigs) */
bool dumping();
void dump( const std::vector<MergeBlockRec> &mbs, const std::vector<std::string> &masterNames, const std::vector<std::string> &slaveNames );

} /* namespace gamdp_bridge */

#endif /* GAMDPBRIDGE_HPP */
