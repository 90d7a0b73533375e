/* GamdpBridge -- see GamdpBridge.hpp.  Everything gam-merge asks of libgamdp goes through here. */
#include "pctg/GamdpBridge.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

namespace gamdp_bridge
{

namespace
{
	gamdp_multi *g_multi = NULL;
	gamdp_multi_seqset *g_master = NULL, *g_slave = NULL;
	uint32_t g_dump_graph = 0xFFFFFFFFu, g_dump_list = 0xFFFFFFFFu;
	bool g_dump_started = false;

	std::vector<int> devices_from_env()
	{
		std::vector<int> d;
		const char *e = std::getenv("GAMDP_DEVICES");
		if( e == NULL ) return d;
		std::stringstream ss(e);
		std::string tok;
		while( std::getline(ss,tok,',') ) if( !tok.empty() ) d.push_back( std::atoi(tok.c_str()) );
		return d;
	}
}

bool enabled() { return !devices_from_env().empty(); }
bool ready() { return g_multi != NULL; }

bool init( const std::vector< std::vector<uint8_t> > &master, const std::vector< std::vector<uint8_t> > &slave )
{
	std::vector<int> dev = devices_from_env();
	if( dev.empty() ) return false;
	if( gamdp_multi_create( &dev[0], int(dev.size()), &g_multi ) != 0 )
	{
		std::cerr << "[gamdp] cannot open the devices of GAMDP_DEVICES; using the CPU alignment" << std::endl;
		g_multi = NULL;
		return false;
	}
	for( int pass = 0; pass < 2; pass++ )
	{
		const std::vector< std::vector<uint8_t> > &ref = pass == 0 ? master : slave;
		std::vector<const uint8_t*> ptr( ref.size() );
		std::vector<uint64_t> len( ref.size() );
		for( size_t i = 0; i < ref.size(); i++ ) { ptr[i] = ref[i].empty() ? NULL : &ref[i][0]; len[i] = ref[i].size(); }
		gamdp_multi_seqset **dst = pass == 0 ? &g_master : &g_slave;
		if( gamdp_multi_seqset_create( g_multi, ref.empty() ? NULL : &ptr[0], ref.empty() ? NULL : &len[0], uint32_t(ref.size()), /*is_ascii=*/0, dst ) != 0 )
		{
			std::cerr << "[gamdp] upload failed: " << gamdp_multi_last_error(g_multi) << "; using the CPU alignment" << std::endl;
			shutdown();
			return false;
		}
	}
	std::cout << "[gamdp] " << dev.size() << " device(s), " << master.size() << " + " << slave.size() << " contigs resident" << std::endl;
	return true;
}

void shutdown()
{
	if( g_master ) gamdp_multi_seqset_destroy(g_master);
	if( g_slave ) gamdp_multi_seqset_destroy(g_slave);
	if( g_multi ) gamdp_multi_destroy(g_multi);
	g_master = g_slave = NULL;
	g_multi = NULL;
}

bool alignMergeBlocks( std::vector<MergeBlockRec> &mbs, uint32_t band )
{
	if( g_multi == NULL ) return false;
	std::vector<gamdp_mb_in> in( mbs.size() );
	std::vector<gamdp_mb_out> out( mbs.size() );
	for( size_t i = 0; i < mbs.size(); i++ )
	{
		in[i].m_id = mbs[i].m_id; in[i].s_id = mbs[i].s_id;
		in[i].m_ltail = mbs[i].m_ltail; in[i].m_rtail = mbs[i].m_rtail;
		in[i].s_ltail = mbs[i].s_ltail; in[i].s_rtail = mbs[i].s_rtail;
		in[i].n_blocks = uint32_t( mbs[i].blocks.size() );
		in[i].blocks = mbs[i].blocks.empty() ? NULL : &mbs[i].blocks[0];
	}
	if( gamdp_multi_align_merge_blocks( g_multi, g_master, g_slave, mbs.empty() ? NULL : &in[0], in.size(), band,
	                                    mbs.empty() ? NULL : &out[0], NULL, 0 ) != 0 )
	{
		std::cerr << "[gamdp] gamdp_multi_align_merge_blocks: " << gamdp_multi_last_error(g_multi) << std::endl;
		return false;
	}
	for( size_t i = 0; i < mbs.size(); i++ )
	{
		/* OUT_OF_RANGE / INVALID = the reference throws from Contig::at / chop_begin: its worker's catch(...) drops the
		 * graph (ThreadedBuildPctg.cc:322-329); the caller does the same with `thrown` */
		mbs[i].thrown = out[i].status != GAMDP_ST_OK;
		mbs[i].align_ok = out[i].align_ok != 0;
		mbs[i].coords_set = out[i].coords_set != 0;
		if( mbs[i].coords_set )   /* PctgBuilder.cc:825-829 returns before writing these otherwise */
		{
			mbs[i].align_rev = out[i].align_rev != 0;
			mbs[i].m_start = out[i].m_start; mbs[i].m_end = out[i].m_end;
			mbs[i].s_start = out[i].s_start; mbs[i].s_end = out[i].s_end;
		}
	}
	return true;
}

bool dumping() { return std::getenv("GAMDP_DUMP_PREFIX") != NULL; }

void dump( const std::vector<MergeBlockRec> &mbs, const std::vector<std::string> &masterNames, const std::vector<std::string> &slaveNames )
{
	const char *prefix = std::getenv("GAMDP_DUMP_PREFIX");
	if( prefix == NULL ) return;
	const std::ios::openmode mode = g_dump_started ? std::ios::app : std::ios::trunc;
	std::ofstream fin( (std::string(prefix) + ".mergeblocks.tsv").c_str(), std::ios::out | mode );
	std::ofstream fout( (std::string(prefix) + ".mergeblocks.out.tsv").c_str(), std::ios::out | mode );
	if( !g_dump_started )
	{
		fin << "#m_name\ts_name\tm_ltail\tm_rtail\ts_ltail\ts_rtail\tn_blocks\t(m_begin m_end s_begin s_end m_strand s_strand n_reads)*\n";
		fout << "#m_name\ts_name\tthrown\talign_ok\talign_rev\tcoords_set\tm_start\tm_end\ts_start\ts_end\n";
		g_dump_started = true;
	}
	for( size_t i = 0; i < mbs.size(); i++ )
	{
		const MergeBlockRec &mb = mbs[i];
		if( mb.graph != g_dump_graph ) { fin << "#graph\n"; g_dump_graph = mb.graph; g_dump_list = 0xFFFFFFFFu; }
		if( mb.list != g_dump_list ) { fin << "#list\n"; g_dump_list = mb.list; }
		const std::string &mn = masterNames[ size_t(mb.m_id) ], &sn = slaveNames[ size_t(mb.s_id) ];
		fin << mn << '\t' << sn << '\t' << mb.m_ltail << '\t' << mb.m_rtail << '\t' << mb.s_ltail << '\t' << mb.s_rtail << '\t' << mb.blocks.size();
		for( size_t k = 0; k < mb.blocks.size(); k++ )
		{
			const gamdp_block &b = mb.blocks[k];
			fin << '\t' << b.m_begin << '\t' << b.m_end << '\t' << b.s_begin << '\t' << b.s_end << '\t' << b.m_strand << '\t' << b.s_strand << '\t' << (long long)b.n_reads;
		}
		fin << '\n';
		fout << mn << '\t' << sn << '\t' << mb.thrown << '\t' << mb.align_ok << '\t' << mb.align_rev << '\t' << mb.coords_set << '\t'
		     << mb.m_start << '\t' << mb.m_end << '\t' << mb.s_start << '\t' << mb.s_end << '\n';
	}
}

} /* namespace gamdp_bridge */
