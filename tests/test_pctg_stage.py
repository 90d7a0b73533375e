"""Post-alignment stage (SURVEY 8f rows f1 + f2), host only: the C++ product behind the C ABI against the Python
restatement (oracle/pctg_oracle.py) on seeded merge lists, and both writers against text rendered by the reference's
own PairedContig / operator<< / writePctgDescriptors (tests/golden/pctg_writers.json)."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import pctg_oracle as PO  # noqa: E402
import _pctgcases as PC  # noqa: E402
from gam_ngs_amd import pctg as P  # noqa: E402
from gam_ngs_amd.lib import GamdpError  # noqa: E402


def names(prefix, n):
    return ["%s_ctg%d" % (prefix, i) for i in range(n)]


def build_product(master, slave, graphs, vote=PC.vote):
    am = P.Assembly(names=names("m", len(master)), seqs=master)
    asl = P.Assembly(names=names("s", len(slave)), seqs=slave)
    pc = P.PairedContigs(am, asl)
    for lists in graphs:
        pc.add_graph(lists, vote)
    pc.finish()
    return am, asl, pc


@pytest.mark.parametrize("stages", [1, 2, 4, 8, 3, 7, 15])
def test_list_surgery_matches_the_restatement(stages):
    for seed in range(150):
        master, slave, graphs = PC.scenario(seed)
        am = P.Assembly(names=names("m", len(master)), seqs=master)
        asl = P.Assembly(names=names("s", len(slave)), seqs=slave)
        for lists in graphs:
            if stages & 8 and not stages & 2:
                for l in lists:      # the inclusion stage alone: give the blocks strands to work with
                    for b in l:
                        b["m_rev"], b["s_rev"] = b["m_start"] & 1, b["s_start"] & 1
            want = PO.prepare(lists, [len(c) for c in master], [len(c) for c in slave], stages)
            got = P.prepare_merge_lists(am, asl, lists, stages)
            assert got == [[{k: int(b[k]) for k in P.MB_KEYS} for b in l] for l in want], (seed, stages)


def test_paired_contigs_match_the_restatement(tmp_path):
    n_pctg = n_slave_rows = n_multi = 0
    for seed in range(300):
        master, slave, graphs = PC.scenario(seed)
        want, merged = PO.run(graphs, master, slave, PC.vote_mb)
        am, asl, pc = build_product(master, slave, graphs)
        assert len(pc) == len(want) and pc.merged == merged, seed
        for i, w in enumerate(want):
            assert list(pc.codes(i)) == w.codes, (seed, i)
            assert pc.rows(i) == w.rows, (seed, i)
            n_slave_rows += sum(1 for r in w.rows if not r[4])
            n_multi += len(w.rows) > 2
        n_pctg += merged
        m_used, s_used = pc.contig_use()
        assert [i for i, u in enumerate(m_used) if u] == sorted(set().union(*[w.master_ids for w in want]))
        assert [i for i, u in enumerate(s_used) if u] == sorted(set().union(*[w.slave_ids for w in want] + [set()]))
        fa, de = tmp_path / "o.gam.fasta", tmp_path / "o.pctgs"
        pc.write_fasta(fa)
        pc.write_descriptors(de)
        assert fa.read_text() == PO.render_fasta(want), seed
        assert de.read_text() == PO.render_descriptors(want, merged, names("m", len(master)), names("s", len(slave))), seed
    assert n_pctg > 300 and n_slave_rows > 100 and n_multi > 200   # the scenarios do exercise the weave


def test_writers_against_text_rendered_by_the_reference(tmp_path):
    cases = json.load(open(os.path.join(HERE, "golden", "pctg_writers.json")))
    assert len(cases) >= 10
    for c in cases:
        master = [[PO.LETTERS.index(ch) for ch in s] for s in c["master"]]
        slave = [[PO.LETTERS.index(ch) for ch in s] for s in c["slave"]]
        # the restatement's renderers on the recorded pieces ...
        pcs = []
        for rows in c["pieces"]:
            p = PO.Pctg()
            for cid, start, end, rev, is_master, src_rev in rows:
                PO._append(p, is_master, cid, PO._load((master if is_master else slave)[cid], src_rev), start, end, rev)
            pcs.append(p)
        assert PO.render_fasta(pcs) == c["fasta"], c["name"]
        assert PO.render_descriptors(pcs, c["merged"], c["master_names"], c["slave_names"]) == c["pctgs"], c["name"]
        # ... and the product, driven by the merge lists the pieces were recorded from
        am = P.Assembly(names=c["master_names"], seqs=master)
        asl = P.Assembly(names=c["slave_names"], seqs=slave)
        pc = P.PairedContigs(am, asl)
        for lists in c["graphs"]:
            pc.add_graph(lists, PC.vote)
        pc.finish()
        fa, de = tmp_path / "g.gam.fasta", tmp_path / "g.pctgs"
        pc.write_fasta(fa)
        pc.write_descriptors(de)
        assert fa.read_text() == c["fasta"], c["name"]
        assert de.read_text() == c["pctgs"], c["name"]


def mb(m_id, m0, m1, s_id, s0, s1, ok=1, rev=0, tails=(1, 1, 1, 1)):
    return dict(m_id=m_id, m_start=m0, m_end=m1, s_id=s_id, s_start=s0, s_end=s1, align_ok=ok, align_rev=rev,
                m_ltail=tails[0], m_rtail=tails[1], s_ltail=tails[2], s_rtail=tails[3], ext_slave_next=1,
                ext_slave_prev=1, m_rev=0, s_rev=0)


def test_hand_built_behaviours():
    master = [[k % 4 for k in range(200)], [(k // 3) % 4 for k in range(150)], [3] * 90]
    slave = [[(k * 7) % 4 for k in range(300)], [1, 2] * 60]
    # 1. a failed alignment in the middle cuts the list unless both sides sit on the same master contig
    am, asl, pc = build_product(master, slave, [[[mb(0, 10, 50, 0, 5, 45), mb(0, 60, 90, 1, 0, 30, ok=0), mb(0, 100, 150, 0, 200, 250)]]])
    assert pc.merged == 1 and [r[:3] for r in pc.rows(0)] == [(0, 0, 9), (0, 10, 50), (0, 51, 99), (0, 100, 150), (0, 151, 199)]
    am, asl, pc = build_product(master, slave, [[[mb(0, 10, 50, 0, 5, 45), mb(1, 60, 90, 0, 60, 90, ok=0), mb(1, 100, 140, 0, 200, 240)]]])
    assert pc.merged == 2
    # 2. all alignments failed: nothing merged, every master contig comes out alone, in id order
    am, asl, pc = build_product(master, slave, [[[mb(0, 10, 50, 0, 5, 45, ok=0)]]])
    assert pc.merged == 0 and len(pc) == 3 and [pc.rows(i)[0][:3] for i in range(3)] == [(0, 0, 199), (1, 0, 149), (2, 0, 89)]
    # 3. a reversed slave: coordinates move to the reverse strand and the slave piece is reverse-complemented
    lists = [[mb(0, 100, 180, 0, 10, 60, rev=1, tails=(1, 0, 0, 0)), mb(1, 20, 70, 0, 100, 290, rev=1, tails=(0, 1, 0, 0))]]
    am, asl, pc = build_product(master, slave, [lists], vote=lambda *a: 1)
    want, merged = PO.run([lists], master, slave, lambda b: 1)
    assert [pc.rows(i) for i in range(len(pc))] == [w.rows for w in want]
    assert any(r[3] and not r[4] for w in want for r in w.rows)          # a reversed slave row exists
    # 4. a region that needs evidence but no callback: the graph fails loudly and leaves nothing behind
    am = P.Assembly(names=names("m", 3), seqs=master)
    asl = P.Assembly(names=names("s", 2), seqs=slave)
    pc = P.PairedContigs(am, asl)
    with pytest.raises(GamdpError):
        pc.add_graph([[mb(0, 10, 110, 0, 5, 20)]], None)
    assert len(pc) == 0
    # 5. contig ids outside the assemblies are refused
    with pytest.raises(GamdpError):
        pc.add_graph([[mb(7, 0, 5, 0, 0, 5)]], PC.vote)


def test_zscore_vote_counts_like_the_reference():
    import random
    rng = random.Random(5)
    for _ in range(300):
        n = rng.randint(0, 12)
        m = [rng.choice((0.0, 0.0, rng.uniform(-3, 3))) for _ in range(n)]
        s = [rng.choice((0.0, 0.0, rng.uniform(-3, 3))) for _ in range(n)]
        assert P.zscore_vote(m, s) == PO.zscore_vote(m, s)
    assert P.zscore_vote([1.0, -0.5], [0.0, 0.0]) == 0      # zeros on the slave side count for the master
    assert P.zscore_vote([0.0, 0.0, 0.0], [1.0, 2.0, 0.0]) == 1
