"""Shared fixtures/helpers for the -m gpu parity tests (HIP path vs oracle / golden vectors)."""
import functools

import _oracle as O
import gam_ngs_amd as gam
from gam_ngs_amd import api


@functools.lru_cache(maxsize=1)
def ctx():
    return gam.Context(0)  # raises without a gfx950 GPU: the product has no CPU fallback


def run_cases(cases, want_ops=True):
    """cases: list of dicts (a, b ASCII bytes, band, begin/end, fs, fe).  Each case gets its own pair of
    sequences in one SequenceSet; returns the list of MyAlignment from ONE batched C-ABI call."""
    c = ctx()
    seqs = []
    for cs in cases:
        seqs.append(cs["a"])
        seqs.append(cs["b"])
    sset = gam.SequenceSet(c, seqs, ascii=True)
    bsw = gam.BandedSmithWaterman(c)
    calls = [(sset.contig(2 * i), cs["begin_a"], cs["end_a"], sset.contig(2 * i + 1), cs["begin_b"], cs["end_b"],
              cs["fs"], cs["fe"]) for i, cs in enumerate(cases)]
    res = bsw.find_alignments(calls, want_ops=want_ops, bands=[cs["band"] for cs in cases])
    sset.close()
    return res


def oracle_for(cs, want_ops=True):
    return O.oracle_align(O.encode(cs["a"]), O.encode(cs["b"]), cs["band"], cs["begin_a"], cs["end_a"], cs["begin_b"],
                          cs["end_b"], cs["fs"], cs["fe"], want_ops=want_ops)
