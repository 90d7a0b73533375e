"""The product's multi-GPU path (gamdp_multi_*: one host thread + context + resident sequence copy per device, static
LPT partition, no collective -- the MI355X stand-in for ThreadedBuildPctg.cc:143-197) driven with TWO contexts on
device 0: results must equal the single-context calls item for item, whatever the partition."""
import random

import pytest

import _cases
import _l1cases
from _gpu import ctx, oracle_for
import _oracle as O
import gam_ngs_amd as gam

pytestmark = pytest.mark.gpu


def test_multi_align_batch_equals_single_context():
    rng = random.Random(41)
    cases = _cases.cases(4100, 300, max_len=600, bands=(5, 20, 150, 512))
    for n in (900, 2500, 7000):   # very different weights: the partition is not a round-robin
        a, b = _cases.related_pair(rng, n)
        cases.append(dict(a=a.encode(), b=b.encode(), band=150, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1,
                          fs=False, fe=False))
    seqs = []
    for cs in cases:
        seqs += [cs["a"], cs["b"]]
    calls = lambda sset: [(sset.contig(2 * i), cs["begin_a"], cs["end_a"], sset.contig(2 * i + 1), cs["begin_b"], cs["end_b"],
                           cs["fs"], cs["fe"]) for i, cs in enumerate(cases)]
    bands = [cs["band"] for cs in cases]
    c = ctx()
    s1 = gam.SequenceSet(c, seqs)
    single = gam.BandedSmithWaterman(c).find_alignments(calls(s1), bands=bands)
    m = gam.MultiContext([0, 0])
    sm = gam.MultiSequenceSet(m, seqs)
    multi = gam.BandedSmithWaterman(m).find_alignments(calls(sm), bands=bands)
    assert [r.key() for r in multi] == [r.key() for r in single]
    assert [r.cells for r in multi] == [r.cells for r in single]
    n_ok = 0
    for cs, r in zip(cases, multi):
        o, _ = oracle_for(cs, False)
        if o.status != O.INVALID:
            assert r.key() == o.key()
            n_ok += o.status == O.OK
    assert n_ok > 100
    # the partitioner really split the work: both halves are non-trivial
    weights = [r.cells for r in single]
    part = gam.api.partition_lpt(weights, 2)
    loads = [sum(w for w, p in zip(weights, part) if p == k) for k in (0, 1)]
    assert min(loads) > 0.4 * sum(loads)
    sm.close(); s1.close(); m.close()


def test_multi_align_merge_blocks_equals_single_context():
    scs = _l1cases.scenarios(4200, 150)
    mk = lambda: [gam.MergeBlock(i, i, [gam.Block(*b) for b in sc["blocks"]], *sc["tails"]) for i, sc in enumerate(scs)]
    c = ctx()
    ms, ss = gam.SequenceSet(c, [sc["master"].encode() for sc in scs]), gam.SequenceSet(c, [sc["slave"].encode() for sc in scs])
    single = gam.PctgBuilder(c, ms, ss).alignMergeBlocks(mk(), audit=12)
    m = gam.MultiContext([0, 0, 0])
    mms, mss = gam.MultiSequenceSet(m, [sc["master"].encode() for sc in scs]), gam.MultiSequenceSet(m, [sc["slave"].encode() for sc in scs])
    multi = gam.PctgBuilder(m, mms, mss).alignMergeBlocks(mk(), audit=12)
    key = lambda mb: (mb.status, mb.align_ok, mb.coords_set, mb.align_rev, mb.m_start, mb.m_end, mb.s_start, mb.s_end, mb.n_dp,
                      mb.cells, [a.key() for a in mb.audit])
    assert [key(x) for x in multi] == [key(x) for x in single]
    assert sum(x.align_ok for x in multi) > 30 and sum(not x.align_ok for x in multi) > 10
