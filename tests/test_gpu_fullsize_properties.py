"""Full-size checks on BASELINE.json's workload shape (50 kb x 50 kb pairs, band 512 and the live
band 150) through size-independent properties of a correct banded alignment, plus spot checks against
the oracle:
  * the edit string re-scores to the reported score (5 / -4 / -8; these pairs start at (0,0) so the
    row-0 gap-free rule does not enter), consumes exactly [begin_a, end] x [begin_b, end] and its
    MATCH count / first / last MATCH positions equal the reported ones;
  * summary-only traceback (run-skipping) == edit-string traceback;
  * results do not depend on batch composition, order or repetition (work-queue scheduling);
  * band 150 and band 512 agree whenever the band-150 path stays strictly inside its band.
"""
import pytest

import _oracle as O
from _gpu import ctx
import gam_ngs_amd as gam
from gam_ngs_amd import api

pytestmark = pytest.mark.gpu
N_PAIRS, LEN = 48, 50000


def walk(ops, a, b, begin_a, begin_b):
    pa, pb, score, nm = begin_a, begin_b, 0, 0
    first = last = None
    for ch in ops:
        if ch in "MX":
            is_match = a[pa] == b[pb] or a[pa] == 4 or b[pb] == 4
            assert (ch == "M") == is_match
            score += 5 if a[pa] == b[pb] else (0 if (a[pa] == 4 or b[pb] == 4) else -4)
            if ch == "M":
                nm += 1
                last = (pa, pb)
                if first is None:
                    first = (pa, pb)
            pa += 1
            pb += 1
        elif ch == "A":
            score -= 8
            pb += 1
        else:
            score -= 8
            pa += 1
    return pa, pb, score, nm, first, last


@pytest.fixture(scope="module")
def workload():
    c = ctx()
    pairs = [api.synth_pair(1000 + k, LEN) for k in range(N_PAIRS)]
    seqs = [x for p in pairs for x in p]
    sset = gam.SequenceSet(c, seqs, ascii=False)
    calls = [(sset.contig(2 * k), 0, LEN - 1, sset.contig(2 * k + 1), 0, len(pairs[k][1]) - 1) for k in range(N_PAIRS)]
    return c, pairs, sset, calls


def test_edit_strings_are_self_consistent(workload):
    c, pairs, sset, calls = workload
    res = gam.BandedSmithWaterman(c, 512).find_alignments(calls, want_ops=True)
    for (m, s), r in zip(pairs, res):
        assert r.status == 0 and r.cells == min(len(s), LEN + 512) * 1025
        pa, pb, score, nm, first, last = walk(r.ops, m, s, r.begin_a(), r.begin_b())
        assert score == r.score() and nm == r.n_match and len(r.ops) == r.length()
        assert r.first_found and r.last_found and first == r.first_match and last == r.last_match
        assert r.homology() == (nm * 100) / len(r.ops)
        # semi-global: ends on the last row or on a's last base
        assert pb == len(s) or pa == LEN


def test_summary_traceback_equals_ops_traceback_and_is_schedule_independent(workload):
    c, pairs, sset, calls = workload
    bsw = gam.BandedSmithWaterman(c, 512)
    with_ops = bsw.find_alignments(calls, want_ops=True)
    summary = bsw.find_alignments(calls, want_ops=False)
    assert [r.key() for r in with_ops] == [r.key() for r in summary]
    # reversed order, duplicated tasks, odd batch size
    perm = list(reversed(range(N_PAIRS))) + [0, 1, 2, 0]
    again = bsw.find_alignments([calls[i] for i in perm], want_ops=False)
    assert [r.key() for r in again] == [summary[i].key() for i in perm]


def test_band150_equals_band512_and_oracle_spot_checks(workload):
    c, pairs, sset, calls = workload
    r512 = gam.BandedSmithWaterman(c, 512).find_alignments(calls)
    r150 = gam.BandedSmithWaterman(c, 150).find_alignments(calls)
    for a, b in zip(r512, r150):
        ka, kb = list(a.key()), list(b.key())
        assert ka == kb  # drift of these pairs stays far below 150 columns
    for k in (0, 17):
        m, s = pairs[k]
        o, _ = O.oracle_align(m, s, 512, 0, LEN - 1, 0, len(s) - 1, want_ops=False)
        assert r512[k].key() == o.key()


def test_throughput_kernels_agree_with_each_other_and_the_oracle_at_batch_size():
    """A batch large enough for the throughput kernels of both bands -- 8 192 library-generated 50 kb pairs: band 512
    goes through the packed two-task kernel, band 150 (>= 6 144 long N-free calls) through the packed eight-task kernel with
    its 2-lane strips and side-by-side walks -- must give the same alignment for every pair (these pairs drift far less
    than 150 columns), and both must equal the CPU oracle on a sample."""
    from gam_ngs_amd import lib as L
    c = ctx()
    P = 8192
    sset = gam.SequenceSet.synthetic(c, 5000, P, LEN)
    tasks = (L.Task * P)()
    for k in range(P):
        t = tasks[k]
        t.a_id, t.b_id = 2 * k, 2 * k + 1
        t.begin_a, t.end_a, t.begin_b, t.end_b = 0, LEN - 1, 0, sset.lengths[2 * k + 1] - 1
    keys = {}
    for band in (512, 150):
        for k in range(P):
            tasks[k].band = band
        out = (L.Result * P)()
        assert c.lib.gamdp_align_batch(c.handle, sset.handle, sset.handle, tasks, P, out, None) == 0
        keys[band] = [out[k].key() for k in range(P)]
        assert all(k[0] == L.ST_OK for k in keys[band])
    assert keys[512] == keys[150]
    for k in (0, 4095, 4096, 8191, 1234, 6001):
        m, s = api.synth_pair(5000 + k, LEN)
        o, _ = O.oracle_align(m, s, 150, 0, LEN - 1, 0, len(s) - 1, want_ops=False)
        assert tuple(keys[150][k]) == tuple(o.key()), k
    sset.close()



def _mid_batch_digest():
    """16 384 library-generated 3 kb pairs at band 150 through whatever kernel the launch planner picks: CRC of every result key."""
    import zlib
    from gam_ngs_amd import lib as L
    c = ctx()
    P, n = 16384, 3000
    sset = gam.SequenceSet.synthetic(c, 9000, P, n)
    tasks = (L.Task * P)()
    for k in range(P):
        t = tasks[k]
        t.a_id, t.b_id, t.band = 2 * k, 2 * k + 1, 150
        t.begin_a, t.end_a, t.begin_b, t.end_b = 0, n - 1, 0, sset.lengths[2 * k + 1] - 1
    out = (L.Result * P)()
    assert c.lib.gamdp_align_batch(c.handle, sset.handle, sset.handle, tasks, P, out, None) == 0
    keys = [tuple(out[k].key()) for k in range(P)]
    sset.close()
    return zlib.crc32(repr(keys).encode()), keys


def test_the_launch_planner_s_kernel_choice_does_not_change_results():
    """Round 4 moved the thresholds of the eight-task kernel (batches of 12 288+ N-free band-150 calls of 1 536+ rows): the same 16 384
    pairs of 3 kb through the planner's choice (eight tasks per wavefront) and, in a child process with GAMDP_OCTO_MIN_ROWS out of
    reach, through the one-task kernel must give the same results, and a sample must equal the oracle."""
    import os, subprocess, sys
    crc, keys = _mid_batch_digest()
    for k in (0, 1, 4095, 8192, 16383, 7777):
        m, s = api.synth_pair(9000 + k, 3000)
        o, _ = O.oracle_align(m, s, 150, 0, 2999, 0, len(s) - 1, want_ops=False)
        assert keys[k] == tuple(o.key()), k
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_gpu_fullsize_properties as T\nprint('CRC', T._mid_batch_digest()[0])\n") % (here, os.path.dirname(here))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GAMDP_OCTO_MIN_ROWS="100000000"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert ("CRC %d" % crc) in r.stdout, (crc, r.stdout[-300:])
