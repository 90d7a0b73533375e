"""GAGE-shaped merge-block workload (tests/_gage.py: a bacterial-size genome, two assemblies with a realistic contig
length spread, 0.3-3 % divergence, N runs; band 150) through the product at the L1 seam and on to the artefacts:

* every merge block's decision fields and every find_alignment call equal the CPU oracle's driver;
* `gamdp-align-mb --pctgs` from FASTA files + a merge-block dump with several graphs gives byte-identical .gam.fasta /
  .pctgs to the two CPU restatements chained together, the paired contigs come out in graphs_list order
  (ThreadedBuildPctg.cc:57-70, 180-181 with --threads 1), a batch over all graphs equals graph-by-graph calls, and the
  multi-device path (`--devices 0,0`) writes the same bytes.
BASELINE configs 1-4 themselves need data that cannot be fetched here (no network): parity on real GAGE inputs stays
unchecked."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _gage as G  # noqa: E402
from _l1oracle import oracle_mb  # noqa: E402

pytestmark = pytest.mark.gpu


def scenario(pb, mb):
    return dict(master=G.to_ascii(pb["master"][mb["m_id"]]["seq"]).decode(), slave=G.to_ascii(pb["slave"][mb["s_id"]]["seq"]).decode(),
                blocks=mb["blocks"], tails=mb["tails"])


def write_inputs(pb, tmp_path):
    with open(tmp_path / "master.fa", "w") as f:
        for i, c in enumerate(pb["master"]):
            s = G.to_ascii(c["seq"]).decode()
            f.write(">m%d len=%d\n" % (i, len(s)))
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + "\n")
    with open(tmp_path / "slave.fa", "w") as f:
        for i, c in enumerate(pb["slave"]):
            f.write(">s%d\n%s\n" % (i, G.to_ascii(c["seq"]).decode()))

    def dump(path, graphs):
        with open(path, "w") as f:
            for g in graphs:
                f.write("#graph\n")
                for l in g:
                    f.write("#list\n")
                    for mb in l:
                        fields = ["m%d" % mb["m_id"], "s%d" % mb["s_id"]] + [str(int(x)) for x in mb["tails"]] + [str(len(mb["blocks"]))]
                        for b in mb["blocks"]:
                            fields += [str(x) for x in b]
                        f.write("\t".join(fields) + "\n")
    dump(tmp_path / "mb.tsv", pb["graphs"])
    return dump


@pytest.mark.parametrize("seed", [11, 12])
def test_merge_blocks_of_a_gage_shaped_problem_match_the_oracle(seed):
    import gam_ngs_amd as gam
    from _gpu import ctx
    pb = G.problem(seed, genome_len=900_000)
    flat, _ = G.merge_blocks(pb)
    c = ctx()
    ms = gam.SequenceSet(c, [bytes(x["seq"]) for x in pb["master"]], ascii=False)
    ss = gam.SequenceSet(c, [bytes(x["seq"]) for x in pb["slave"]], ascii=False)
    mbs = [gam.MergeBlock(mb["m_id"], mb["s_id"], [gam.Block(*b) for b in mb["blocks"]], *mb["tails"]) for mb in flat]
    gam.PctgBuilder(c, ms, ss).alignMergeBlocks(mbs, audit=40)
    n_tail = n_rev = 0
    for mb, got in zip(flat, mbs):
        o, oaud = oracle_mb(scenario(pb, mb), audit_cap=40)
        assert (got.status, got.align_ok, got.coords_set, got.n_dp, got.cells) == (o.status, bool(o.align_ok), bool(o.touched), o.n_dp, o.cells)
        if o.touched:
            assert (got.align_rev, got.m_start, got.m_end, got.s_start, got.s_end) == (bool(o.align_rev), o.m_start, o.m_end, o.s_start, o.s_end)
        assert [a.key() for a in got.audit] == oaud
        n_tail += got.n_dp > len(mb["blocks"])
        n_rev += bool(got.align_ok and got.align_rev)
    assert len(flat) >= 40 and n_rev >= 5 and n_tail >= 2 and sum(g.align_ok for g in mbs) >= 0.7 * len(flat)
    ms.close(); ss.close()


def test_files_to_gam_fasta_several_graphs_in_graphs_list_order(tmp_path):
    import pctg_oracle as PO
    tool = os.path.join(ROOT, "gam_ngs_amd", "gamdp-align-mb")
    pb = G.problem(21, genome_len=900_000)
    assert len(pb["graphs"]) >= 3
    dump = write_inputs(pb, tmp_path)
    master = [[int(x) for x in c["seq"]] for c in pb["master"]]
    slave = [[int(x) for x in c["seq"]] for c in pb["slave"]]
    # the CPU answer: oracle driver per merge block, then the list surgery / buildPctgs restatement over all graphs
    graphs = []
    for g in pb["graphs"]:
        lists, thrown = [], False
        for l in g:
            lists.append([])
            for mb in l:
                o, _ = oracle_mb(scenario(pb, mb), audit_cap=1)
                thrown |= o.status != 0
                lists[-1].append(dict(m_id=mb["m_id"], s_id=mb["s_id"], m_start=o.m_start, m_end=o.m_end, s_start=o.s_start,
                                      s_end=o.s_end, align_ok=int(o.align_ok), align_rev=int(o.align_rev) if o.touched else 0,
                                      m_ltail=1, m_rtail=1, s_ltail=1, s_rtail=1, ext_slave_next=1, ext_slave_prev=1, m_rev=0, s_rev=0))
        if not thrown:
            graphs.append(lists)
    want, merged = PO.run(graphs, master, slave, lambda mb: 0)
    mn, sn = ["m%d" % i for i in range(len(master))], ["s%d" % i for i in range(len(slave))]
    args = [tool, str(tmp_path / "master.fa"), str(tmp_path / "slave.fa")]
    subprocess.run(args + [str(tmp_path / "mb.tsv"), str(tmp_path / "out.tsv"), "--pctgs", str(tmp_path / "run"), "--vote", "master"],
                   check=True, timeout=600)
    fasta, desc = open(tmp_path / "run.gam.fasta").read(), open(tmp_path / "run.pctgs").read()
    assert merged >= 3
    assert fasta == PO.render_fasta(want)
    assert desc == PO.render_descriptors(want, merged, mn, sn)
    # graphs_list order: the merged paired contigs of the batch = those of graph-by-graph runs, concatenated in order
    per_graph = []
    for gi, g in enumerate(pb["graphs"]):
        dump(tmp_path / ("g%d.tsv" % gi), [g])
        subprocess.run(args + [str(tmp_path / ("g%d.tsv" % gi)), str(tmp_path / "o.tsv"), "--pctgs", str(tmp_path / ("g%d" % gi)),
                               "--vote", "master"], check=True, timeout=600)
        recs = open(tmp_path / ("g%d.gam.fasta" % gi)).read().split(">")[1:]
        rows = [l for l in open(tmp_path / ("g%d.pctgs" % gi)).read().split("\n# ---")[0].splitlines() if not l.startswith("#")]
        n_merged = len(set(r.split("\t")[0] for r in rows))
        per_graph += [r.split("\n", 1)[1] for r in recs[:n_merged]]       # sequences only: ids restart per run
    batch = [r.split("\n", 1)[1] for r in fasta.split(">")[1:]][:merged]
    assert batch == per_graph
    # the multi-device path writes the same bytes (two contexts on device 0)
    subprocess.run(args + [str(tmp_path / "mb.tsv"), str(tmp_path / "out2.tsv"), "--pctgs", str(tmp_path / "run2"), "--vote", "master",
                           "--devices", "0,0"], check=True, timeout=600)
    assert open(tmp_path / "run2.gam.fasta").read() == fasta and open(tmp_path / "run2.pctgs").read() == desc
    assert open(tmp_path / "out2.tsv").read() == open(tmp_path / "out.tsv").read()
    # the side outputs of src/Merge.cc:335-373, 412-431: slave contigs without blocks before / after the coverage filter,
    # slave contigs nothing used -- from the .blocks file gam-merge loaded and the one its filter kept
    from gam_ngs_amd import api

    def rec(m, sid, b):
        return dict(n_reads=int(b[6]), m_block_reads_len=100, m_reads_len=100, s_block_reads_len=100, s_reads_len=100, m_ctg=m,
                    m_begin=int(b[0]), m_end=int(b[1]), s_ctg=sid, s_begin=int(b[2]), s_end=int(b[3]), m_strand=b[4], s_strand=b[5])
    kept = [rec(mb["m_id"], mb["s_id"], b) for g in pb["graphs"] for l in g for mb in l for b in mb["blocks"]]
    in_kept = set(r["s_ctg"] for r in kept)
    lost = [i for i in range(len(slave)) if i not in in_kept][::2]     # these had blocks the coverage filter removed
    api.write_blocks(tmp_path / "all.blocks", kept + [rec(0, i, (0, 99, 0, 99, "+", "+", 3)) for i in lost])
    api.write_blocks(tmp_path / "kept.blocks", kept)
    subprocess.run(args + [str(tmp_path / "mb.tsv"), str(tmp_path / "out3.tsv"), "--pctgs", str(tmp_path / "run3"), "--vote", "master",
                           "--blocks", str(tmp_path / "all.blocks"), "--blocks-filtered", str(tmp_path / "kept.blocks")], check=True, timeout=600)
    used = set()
    for p in want:
        used |= set(p.slave_ids)
    s_bf = [i for i in range(len(slave)) if i not in in_kept and i not in lost]
    unused = [i for i in range(len(slave)) if i not in used and i not in s_bf and i not in lost]
    assert s_bf and lost and unused

    def text(ids):
        out = []
        for i in ids:
            seq = G.to_ascii(pb["slave"][i]["seq"]).decode()
            out.append(">s%d\n" % i + "".join(seq[k:k + 60] + "\n" for k in range(0, len(seq), 60)))
        return "".join(out)
    assert open(tmp_path / "run3.noblocks.BF.fasta").read() == text(s_bf)
    assert open(tmp_path / "run3.noblocks.AF.fasta").read() == text(lost)
    assert open(tmp_path / "run3.notmerged.fasta").read() == text(unused)
    assert open(tmp_path / "run3.gam.fasta").read() == fasta

