"""End-to-end check with a known answer (tests/_genome.py): two synthetic assemblies of one genome -> merge-block
alignment -> list surgery -> buildPctgs.  Every paired contig must be a gap-free, one-directional walk along the genome
and must align to the genome stretch it claims to cover.  The CPU test runs the two restatements (oracle/) chained
together -- it is the semantic check of what they restate; the GPU test runs the product (HIP alignment + C++ stage
behind the C ABI) and must reproduce the restatements' paired contigs exactly."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
import _genome as G  # noqa: E402
import _oracle as O  # noqa: E402
import pctg_oracle as PO  # noqa: E402
from _l1oracle import oracle_mb  # noqa: E402

CODE = {"A": 0, "T": 1, "C": 2, "G": 3}


def oracle_chain(pb):
    master = [[CODE[c] for c in m["seq"]] for m in pb["master"]]
    slave = [[CODE[c] for c in s["seq"]] for s in pb["slave"]]
    ml = []
    for mb in pb["merge_list"]:
        sc = dict(master=pb["master"][mb["m_id"]]["seq"], slave=pb["slave"][mb["s_id"]]["seq"], blocks=mb["blocks"],
                  tails=mb["tails"])
        o, _ = oracle_mb(sc)
        assert o.status == 0
        ml.append(dict(m_id=mb["m_id"], s_id=mb["s_id"], m_start=o.m_start, m_end=o.m_end, s_start=o.s_start,
                       s_end=o.s_end, align_ok=int(o.align_ok), align_rev=int(o.align_rev), m_ltail=1, m_rtail=1,
                       s_ltail=1, s_rtail=1, ext_slave_next=1, ext_slave_prev=1, m_rev=0, s_rev=0))
    pcs, merged = PO.run([[ml]], master, slave, lambda b: 0)
    return master, slave, ml, pcs, merged


def check_against_genome(pb, pcs, merged):
    covered = 0
    for p in pcs[:merged]:
        g_from, g_to, fwd, worst = G.check_walk(pb, p, max_jump=3)
        lo, hi = min(g_from, g_to), max(g_from, g_to)
        assert abs(len(p.codes) - (hi - lo + 1)) <= 0.01 * len(p.codes) + 5
        seq = "".join(PO.LETTERS[c] for c in p.codes)
        if not fwd:
            seq = G.revcomp(seq)
        a, b = O.encode(seq), O.encode(pb["genome"][lo:hi + 1])
        r, _ = O.oracle_align(a, b, 150, 0, len(a) - 1, 0, len(b) - 1, want_ops=False)
        assert r.status == 0 and r.length >= 0.99 * len(a) and r.homology >= 99.0, (r.length, len(a), r.homology)
        covered += hi - lo + 1
    return covered


@pytest.mark.parametrize("seed", range(4))
def test_restatements_rebuild_the_genome(seed):
    pb = G.problem(seed)
    master, slave, ml, pcs, merged = oracle_chain(pb)
    assert all(b["align_ok"] for b in ml)
    assert any(b["align_rev"] for b in ml) or seed not in (0, 1)
    covered = check_against_genome(pb, pcs, merged)
    # the merged paired contigs span more than any single assembly's longest contig and most of the genome
    assert merged >= 1 and covered >= 0.8 * len(pb["genome"])
    assert max(len(p.codes) for p in pcs[:merged]) > max(m["n"] for m in pb["master"])
    assert any(not r[4] for p in pcs[:merged] for r in p.rows)      # slave sequence fills master gaps


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(4, 8))
def test_product_rebuilds_the_genome(seed, tmp_path):
    import gam_ngs_amd as gam
    from gam_ngs_amd import pctg as P
    from _gpu import ctx
    pb = G.problem(seed, genome_len=60000)
    master, slave, ml, want, merged = oracle_chain(pb)
    c = ctx()
    ms = gam.SequenceSet(c, [m["seq"].encode() for m in pb["master"]])
    ss = gam.SequenceSet(c, [s["seq"].encode() for s in pb["slave"]])
    mbs = [gam.MergeBlock(mb["m_id"], mb["s_id"], [gam.Block(*b) for b in mb["blocks"]], *mb["tails"]) for mb in pb["merge_list"]]
    gam.PctgBuilder(c, ms, ss).alignMergeBlocks(mbs)
    lists = [[dict(m_id=mb.m_id, s_id=mb.s_id, m_start=mb.m_start, m_end=mb.m_end, s_start=mb.s_start, s_end=mb.s_end,
                   align_ok=int(mb.align_ok), align_rev=int(mb.align_rev), m_ltail=1, m_rtail=1, s_ltail=1, s_rtail=1,
                   ext_slave_next=1, ext_slave_prev=1) for mb in mbs]]
    assert [[(b["m_start"], b["m_end"], b["s_start"], b["s_end"], b["align_ok"], b["align_rev"]) for b in lists[0]]] == \
           [[(b["m_start"], b["m_end"], b["s_start"], b["s_end"], b["align_ok"], b["align_rev"]) for b in ml]]
    am = P.Assembly(names=["m%d" % i for i in range(len(master))], seqs=master)
    asl = P.Assembly(names=["s%d" % i for i in range(len(slave))], seqs=slave)
    pc = P.PairedContigs(am, asl)
    pc.add_graph(lists, lambda *a: 0)
    pc.finish()
    assert len(pc) == len(want) and pc.merged == merged
    for i, w in enumerate(want):
        assert list(pc.codes(i)) == w.codes and pc.rows(i) == w.rows
    assert check_against_genome(pb, want, merged) >= 0.8 * len(pb["genome"])
