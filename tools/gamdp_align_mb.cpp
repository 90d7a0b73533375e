// gamdp-align-mb: run gam-merge's merge-block alignment step (PctgBuilder::alignMergeBlock for a whole list,
// lib/src/pctg/BuildPctgFunctions.cc:82-84) on an MI355X from files, without the rest of gam-merge.
//
//   gamdp-align-mb <master.fasta> <slave.fasta> <mergeblocks.tsv> <out.tsv> [--band N] [--device D | --devices D0,D1,..]
//                  [--repeat K] [--pctgs PREFIX] [--vote master|slave|fail] [--blocks all.blocks [--blocks-filtered kept.blocks]]
//
// --devices runs the step on several GPUs of the node (gamdp_multi_*: merge blocks partitioned statically by predicted
// cells, one host thread + context per device, no collective); a device may be listed twice.  The output is the same
// whatever the device list -- unlike gam-merge --threads N, whose paired-contig order depends on thread timing.
//
// mergeblocks.tsv: one merge block per line, tab separated ('#' lines are comments):
//   m_name s_name m_ltail m_rtail s_ltail s_rtail n_blocks  then n_blocks x (m_begin m_end s_begin s_end m_strand s_strand n_reads)
// i.e. exactly what alignMergeBlock reads from the MergeBlock and from graph.getBlocks(mb.vertex); INTEGRATION.md
// has the 15-line dump to add to the reference at the seam.  out.tsv gets the fields alignMergeBlock writes.
// A line "#graph" starts the next assembly graph and "#list" the next merge list of the current graph.
//
// With --pctgs the rest of buildPctg runs as well (list surgery + buildPctgs, BuildPctgFunctions.cc:86-92, then ids,
// single-contig pctgs and the writers of src/Merge.cc:380-465): PREFIX.gam.fasta and PREFIX.pctgs are written.  A graph
// holding a merge block on which the reference would have thrown contributes nothing, like ThreadedBuildPctg.cc:322-329.
// --vote says what to do when a block region needs the BAM evidence of computeZScore (PctgBuilder.cc:147-168), which
// this tool does not have: take the master's copy, the slave's, or stop (default).
// With --blocks (the .blocks file gam-merge loaded; --blocks-filtered = what its coverage filter kept, default: all of it)
// the side outputs of src/Merge.cc:335-373, 412-431 are written too: PREFIX.noblocks.BF.fasta, PREFIX.noblocks.AF.fasta
// (slave contigs no block lies on, before / after the filter) and PREFIX.notmerged.fasta (slave contigs in neither set and
// in no paired contig).  Contig ids in the .blocks files are positions in the two FASTA files.
//
// It only uses the C ABI of include/gamdp.h (this file is also the C++ usage example of the library).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "gamdp.h"

static void die(const std::string& m)
{
    std::fprintf(stderr, "gamdp-align-mb: %s\n", m.c_str());
    std::exit(1);
}

static int g_vote = -1;  // --vote: what stands in for the BAM evidence
static int region_vote(void*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t) { return g_vote < 0 ? GAMDP_EINVAL : g_vote; }

int main(int argc, char** argv)
{
    if (argc < 5) die("usage: gamdp-align-mb <master.fasta> <slave.fasta> <mergeblocks.tsv> <out.tsv> [--band N] [--device D] [--repeat K] "
                      "[--pctgs PREFIX] [--vote master|slave|fail] [--blocks all.blocks [--blocks-filtered kept.blocks]]");
    unsigned band = GAMDP_DEFAULT_BAND;
    int device = 0, repeat = 1;
    std::vector<int> devices;
    std::string pctg_prefix, blocks_path, filtered_path;
    for (int i = 5; i + 1 < argc; i += 2) {
        if (!std::strcmp(argv[i], "--band")) band = (unsigned)std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--device")) device = std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--devices")) {
            std::stringstream ds(argv[i + 1]);
            std::string tok;
            while (std::getline(ds, tok, ',')) devices.push_back(std::atoi(tok.c_str()));
            if (devices.empty()) die("--devices takes a comma-separated list");
        }
        else if (!std::strcmp(argv[i], "--repeat")) repeat = std::atoi(argv[i + 1]);
        else if (!std::strcmp(argv[i], "--pctgs")) pctg_prefix = argv[i + 1];
        else if (!std::strcmp(argv[i], "--blocks")) blocks_path = argv[i + 1];
        else if (!std::strcmp(argv[i], "--blocks-filtered")) filtered_path = argv[i + 1];
        else if (!std::strcmp(argv[i], "--vote")) {
            if (!std::strcmp(argv[i + 1], "master")) g_vote = 0;
            else if (!std::strcmp(argv[i + 1], "slave")) g_vote = 1;
            else if (!std::strcmp(argv[i + 1], "fail")) g_vote = -1;
            else die("--vote takes master, slave or fail");
        } else die(std::string("unknown option ") + argv[i]);
    }

    gamdp_fasta *fm = nullptr, *fs = nullptr;
    if (gamdp_fasta_open(argv[1], &fm)) die(std::string("cannot load ") + argv[1]);
    if (gamdp_fasta_open(argv[2], &fs)) die(std::string("cannot load ") + argv[2]);
    std::map<std::string, int32_t> mid, sid;
    for (uint32_t i = 0; i < gamdp_fasta_count(fm); i++) mid[gamdp_fasta_name(fm, i)] = (int32_t)i;
    for (uint32_t i = 0; i < gamdp_fasta_count(fs); i++) sid[gamdp_fasta_name(fs, i)] = (int32_t)i;

    std::vector<std::vector<gamdp_block>> blocks;
    std::vector<gamdp_mb_in> in;
    std::vector<uint32_t> graph_of, list_of;  // per merge block
    uint32_t graph = 0, list = 0;
    {
        std::ifstream f(argv[3]);
        if (!f) die(std::string("cannot open ") + argv[3]);
        std::string line;
        size_t ln = 0;
        while (std::getline(f, line)) {
            ln++;
            if (line.compare(0, 6, "#graph") == 0) { if (!in.empty()) { graph++; list = 0; } continue; }
            if (line.compare(0, 5, "#list") == 0) { if (!in.empty() && graph_of.back() == graph) list++; continue; }
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ss(line);
            std::string mn, sn;
            int t[4];
            unsigned nb;
            if (!(ss >> mn >> sn >> t[0] >> t[1] >> t[2] >> t[3] >> nb)) die("bad merge-block line " + std::to_string(ln));
            if (!mid.count(mn) || !sid.count(sn)) die("unknown contig on line " + std::to_string(ln));
            blocks.emplace_back();
            for (unsigned k = 0; k < nb; k++) {
                gamdp_block b;
                std::string ms, sst;
                long long nr;
                if (!(ss >> b.m_begin >> b.m_end >> b.s_begin >> b.s_end >> ms >> sst >> nr)) die("bad block on line " + std::to_string(ln));
                b.m_strand = ms[0];
                b.s_strand = sst[0];
                b.n_reads = nr;
                blocks.back().push_back(b);
            }
            gamdp_mb_in m;
            m.m_id = mid[mn]; m.s_id = sid[sn];
            m.m_ltail = (uint8_t)t[0]; m.m_rtail = (uint8_t)t[1]; m.s_ltail = (uint8_t)t[2]; m.s_rtail = (uint8_t)t[3];
            m.n_blocks = nb;
            m.blocks = nullptr;
            in.push_back(m);
            graph_of.push_back(graph);
            list_of.push_back(list);
        }
        for (size_t i = 0; i < in.size(); i++) in[i].blocks = blocks[i].data();
    }

    gamdp_ctx* ctx = nullptr;
    gamdp_multi* multi = nullptr;
    gamdp_seqset *master = nullptr, *slave = nullptr;
    gamdp_multi_seqset *mmaster = nullptr, *mslave = nullptr;
    if (devices.empty()) {
        if (gamdp_ctx_create(device, &ctx)) die("no usable gfx950 GPU (libgamdp has no CPU fallback)");
        if (gamdp_seqset_create_from_fasta(ctx, fm, &master) || gamdp_seqset_create_from_fasta(ctx, fs, &slave)) die(gamdp_last_error(ctx));
    } else {
        if (gamdp_multi_create(devices.data(), (int)devices.size(), &multi)) die("cannot open the listed gfx950 GPUs (libgamdp has no CPU fallback)");
        if (gamdp_multi_seqset_create_from_fasta(multi, fm, &mmaster) || gamdp_multi_seqset_create_from_fasta(multi, fs, &mslave)) die(gamdp_multi_last_error(multi));
    }

    std::vector<gamdp_mb_out> out(in.size());
    double best_s = 1e30;
    for (int r = 0; r < repeat; r++) {
        const auto t0 = std::chrono::steady_clock::now();
        if (multi) {
            if (gamdp_multi_align_merge_blocks(multi, mmaster, mslave, in.data(), in.size(), band, out.data(), nullptr, 0)) die(gamdp_multi_last_error(multi));
        } else if (gamdp_align_merge_blocks(ctx, master, slave, in.data(), in.size(), band, out.data(), nullptr, 0)) die(gamdp_last_error(ctx));
        best_s = std::min(best_s, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }

    unsigned long long cells = 0, ndp = 0, ok = 0, thrown = 0;
    {
        std::ofstream o(argv[4]);
        o << "#m_name\ts_name\tstatus\talign_ok\talign_rev\tcoords_set\tm_start\tm_end\ts_start\ts_end\tn_dp\tcells\n";
        for (size_t i = 0; i < in.size(); i++) {
            const gamdp_mb_out& r = out[i];
            o << gamdp_fasta_name(fm, (uint32_t)in[i].m_id) << '\t' << gamdp_fasta_name(fs, (uint32_t)in[i].s_id) << '\t' << (int)r.status
              << '\t' << (int)r.align_ok << '\t' << (int)r.align_rev << '\t' << (int)r.coords_set << '\t' << r.m_start << '\t' << r.m_end
              << '\t' << r.s_start << '\t' << r.s_end << '\t' << r.n_dp << '\t' << (unsigned long long)r.cells << '\n';
            cells += r.cells; ndp += r.n_dp; ok += r.align_ok; thrown += (r.status != GAMDP_ST_OK);
        }
    }
    std::fprintf(stderr, "gamdp-align-mb: %zu merge blocks, %llu find_alignment calls, %.3e cell updates, %llu align_ok, %llu would make the "
                 "reference throw; %.3f s -> %.2f GCUPS (band %u)\n", in.size(), ndp, (double)cells, ok, thrown, best_s, cells / best_s / 1e9, band);
    if (!pctg_prefix.empty()) {
        gamdp_pctgs* pc = nullptr;
        if (gamdp_pctgs_create(fm, fs, &pc)) die("gamdp_pctgs_create failed");
        size_t dropped = 0, i = 0;
        while (i < in.size()) {
            size_t j = i;
            bool thrown_here = false;
            std::vector<gamdp_mblock> mbs;
            std::vector<uint32_t> sizes;
            while (j < in.size() && graph_of[j] == graph_of[i]) {
                if (sizes.empty() || list_of[j] != list_of[j - 1]) sizes.push_back(0);
                const gamdp_mb_out& r = out[j];
                thrown_here |= r.status != GAMDP_ST_OK;
                gamdp_mblock b;
                std::memset(&b, 0, sizeof b);
                b.m_id = in[j].m_id; b.s_id = in[j].s_id;
                b.m_ltail = in[j].m_ltail; b.m_rtail = in[j].m_rtail; b.s_ltail = in[j].s_ltail; b.s_rtail = in[j].s_rtail;
                b.align_ok = r.align_ok; b.align_rev = r.align_rev;
                b.m_start = r.m_start; b.m_end = r.m_end; b.s_start = r.s_start; b.s_end = r.s_end;
                b.ext_slave_next = b.ext_slave_prev = 1;  // PctgBuilder.cc:915-916
                mbs.push_back(b);
                sizes.back()++;
                j++;
            }
            if (thrown_here) dropped++;
            else if (gamdp_pctgs_add_graph(pc, mbs.data(), sizes.data(), (uint32_t)sizes.size(), region_vote, nullptr))
                die(std::string("graph ") + std::to_string(graph_of[i]) + ": " + gamdp_pctgs_last_error(pc) + " (see --vote)");
            i = j;
        }
        if (gamdp_pctgs_finish(pc)) die("gamdp_pctgs_finish failed");
        if (gamdp_pctgs_write_fasta(pc, (pctg_prefix + ".gam.fasta").c_str()) || gamdp_pctgs_write_descriptors(pc, (pctg_prefix + ".pctgs").c_str()))
            die("cannot write " + pctg_prefix + ".gam.fasta / .pctgs");
        std::fprintf(stderr, "gamdp-align-mb: %u paired contigs (%u merged, %zu graphs dropped) -> %s.gam.fasta, %s.pctgs\n", gamdp_pctgs_count(pc),
                     gamdp_pctgs_merged_count(pc), dropped, pctg_prefix.c_str(), pctg_prefix.c_str());
        if (!blocks_path.empty()) {   // src/Merge.cc:273-297, 335-373, 412-431
            gamdp_blocks *all = nullptr, *kept = nullptr;
            if (gamdp_blocks_open(blocks_path.c_str(), 1, &all)) die("cannot read " + blocks_path);
            if (!filtered_path.empty() && gamdp_blocks_open(filtered_path.c_str(), 1, &kept)) die("cannot read " + filtered_path);
            const uint32_t nm = gamdp_fasta_count(fm), ns = gamdp_fasta_count(fs);
            std::vector<uint8_t> m_bf(nm + 1), s_bf(ns + 1), m_af(nm + 1), s_af(ns + 1), unused(ns + 1);
            if (gamdp_no_blocks_contigs(gamdp_blocks_data(all), gamdp_blocks_count(all), nm, ns, m_bf.data(), s_bf.data()))
                die(blocks_path + ": a block names a contig the FASTA files do not have (master and slave swapped?)");
            const gamdp_blocks* k = kept ? kept : all;
            if (gamdp_no_blocks_after_filter(gamdp_blocks_data(k), gamdp_blocks_count(k), nm, ns, m_bf.data(), s_bf.data(), m_af.data(), s_af.data()))
                die("the filtered .blocks file names a contig the FASTA files do not have");
            if (gamdp_pctgs_not_merged(pc, s_bf.data(), s_af.data(), unused.data())) die("gamdp_pctgs_not_merged failed");
            if (gamdp_fasta_write_selected(fs, s_bf.data(), (pctg_prefix + ".noblocks.BF.fasta").c_str()) ||
                gamdp_fasta_write_selected(fs, s_af.data(), (pctg_prefix + ".noblocks.AF.fasta").c_str()) ||
                gamdp_fasta_write_selected(fs, unused.data(), (pctg_prefix + ".notmerged.fasta").c_str()))
                die("cannot write the side outputs of " + pctg_prefix);
            gamdp_blocks_close(all);
            gamdp_blocks_close(kept);
        }
        gamdp_pctgs_destroy(pc);
    }
    gamdp_seqset_destroy(master);
    gamdp_seqset_destroy(slave);
    gamdp_multi_seqset_destroy(mmaster);
    gamdp_multi_seqset_destroy(mslave);
    gamdp_multi_destroy(multi);
    gamdp_ctx_destroy(ctx);
    gamdp_fasta_close(fm);
    gamdp_fasta_close(fs);
    return 0;
}
