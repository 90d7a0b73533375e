"""gam-merge's side outputs (src/Merge.cc:273-297, 335-373, 412-431), host only: which slave contigs no block lies on
(before / after the coverage filter), which no paired contig uses, and the FASTA text of a selection -- the text pinned
to the reference's own operator<<(ostream&, const Contig&) through tests/golden/side_outputs.json
(generator: tests/golden/make_golden_side.py)."""
import json
import os
import random

import pytest

import gam_ngs_amd as gam
from gam_ngs_amd import api, pctg
from gam_ngs_amd import lib as L

HERE = os.path.dirname(os.path.abspath(__file__))
CODE = {"A": 0, "T": 1, "C": 2, "G": 3, "N": 4}


def blk(m, s, n_reads=10):
    return dict(n_reads=n_reads, m_block_reads_len=100, m_reads_len=100, s_block_reads_len=100, s_reads_len=100,
                m_ctg=m, m_begin=0, m_end=99, s_ctg=s, s_begin=0, s_end=99, m_strand="+", s_strand="+")


def test_selected_contigs_text_equals_the_reference_writer(tmp_path):
    cases = json.load(open(os.path.join(HERE, "golden", "side_outputs.json")))
    assert len(cases) >= 10
    for k, cs in enumerate(cases):
        asm = pctg.Assembly(names=cs["names"], seqs=[[CODE[ch] for ch in s] for s in cs["seqs"]])
        p = tmp_path / ("sel%d.fasta" % k)
        api.write_selected_fasta(asm, cs["select"], p)
        assert p.read_text() == cs["text"], k
        asm.close()


def nbc_restatement(blocks, n_master, n_slave):
    """Block.cc:810-862 read line by line: mark, then flip."""
    m, s = [0] * n_master, [0] * n_slave
    for b in blocks:
        m[b["m_ctg"]] = 1
        s[b["s_ctg"]] = 1
    return [1 - x for x in m], [1 - x for x in s]


def test_no_blocks_sets_against_a_restatement_of_block_cc():
    rng = random.Random(5)
    for _ in range(200):
        nm, ns = rng.randint(1, 30), rng.randint(1, 40)
        before = [blk(rng.randrange(nm), rng.randrange(ns)) for _ in range(rng.randint(0, 60))]
        after = [b for b in before if rng.random() < 0.6]        # what a coverage filter leaves
        m_bf, s_bf = api.no_blocks_contigs(before, nm, ns)
        assert (m_bf, s_bf) == nbc_restatement(before, nm, ns)
        m_af, s_af = api.no_blocks_after_filter(after, nm, ns, m_bf, s_bf)
        # Block.cc:865-925: mark the filtered list's contigs, OR the before-filter sets, flip
        mm, ss = nbc_restatement(after, nm, ns)
        want_m = [int(not ((1 - mm[i]) or m_bf[i])) for i in range(nm)]
        want_s = [int(not ((1 - ss[i]) or s_bf[i])) for i in range(ns)]
        assert (m_af, s_af) == (want_m, want_s)
        # a contig is in exactly one of: has blocks after the filter / lost them in the filter / never had any
        for i in range(ns):
            assert (1 - ss[i]) + s_af[i] + s_bf[i] == 1


def test_block_with_a_contig_outside_the_assemblies_is_refused():
    with pytest.raises(L.GamdpError):
        api.no_blocks_contigs([blk(3, 0)], 3, 5)      # the reference prints an error and exits (Block.cc:832-838)
    with pytest.raises(L.GamdpError):
        api.no_blocks_contigs([blk(0, -1)], 3, 5)
    assert api.no_blocks_contigs([], 2, 3) == ([1, 1], [1, 1, 1])


def test_not_merged_is_what_no_paired_contig_and_no_no_blocks_set_holds(tmp_path):
    rng = random.Random(9)
    master = [[rng.randrange(4) for _ in range(300)] for _ in range(3)]
    slave = [[rng.randrange(4) for _ in range(rng.randint(50, 200))] for _ in range(6)]
    m = pctg.Assembly(names=["m%d" % i for i in range(3)], seqs=master)
    s = pctg.Assembly(names=["s%d" % i for i in range(6)], seqs=slave)
    pc = pctg.PairedContigs(m, s)
    # one merge list whose block joins master 0 and slave 2 over their full lengths
    pc.add_graph([[dict(m_id=0, m_start=0, m_end=199, s_id=2, s_start=0, s_end=len(slave[2]) - 1, align_rev=0, align_ok=1,
                        m_ltail=1, m_rtail=1, s_ltail=1, s_rtail=1)]], vote=lambda *a: 0)
    pc.finish()
    used = pc.contig_use()[1]
    bf, af = [0, 1, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0]
    nm = pc.not_merged(bf, af)
    assert nm == [int(not (used[i] or bf[i] or af[i])) for i in range(6)]
    assert nm[1] == 0 and nm[3] == 0 and sum(nm) >= 2
    out = tmp_path / "x.notmerged.fasta"
    api.write_selected_fasta(s, nm, out)
    names = [l[1:] for l in out.read_text().splitlines() if l.startswith(">")]
    assert names == ["s%d" % i for i in range(6) if nm[i]]
    pc.close(); m.close(); s.close()
