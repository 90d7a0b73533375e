"""GPU parity of gamdp_align_batch (HIP kernels, through the C ABI) against the reference-generated
golden vectors and the CPU oracle: bit-exact on every field a caller of find_alignment can observe
(begin_a/b, score, #matches, length, homology, first/last match, status) and on the edit string."""
import ctypes
import random
import zlib

import pytest

import _cases
import _golden as G
import _oracle as O
from _gpu import ctx, oracle_for, run_cases
import gam_ngs_amd as gam
from gam_ngs_amd import api
from gam_ngs_amd import lib as L

pytestmark = pytest.mark.gpu

# libgamdp_diag.so: same sources with -DGAMDP_DIAG (`make -C gam_ngs_amd/csrc diag`); the product library ignores the
# GAMDP_DIAG_* switches these tests flip in child processes
import os as _os
DIAG_LIB = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gam_ngs_amd", "libgamdp_diag.so")


def test_product_build_ignores_diagnostics_switches():
    """The shipped library must not honour the result-invalidating switches: with them set in a child
    process, a band-512 pair still comes back with its traceback done and bit-exact."""
    import subprocess, sys
    code = (
        "import sys, os; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import random, _cases, _oracle as O\n"
        "from _gpu import run_cases, oracle_for\n"
        "import gam_ngs_amd as gam\n"
        "assert gam.load_library().gamdp_build_info() == 0\n"
        "a, b = _cases.related_pair(random.Random(5), 9000)\n"
        "cs = dict(a=a.encode(), b=b.encode(), band=512, begin_a=0, end_a=len(a)-1, begin_b=0, end_b=len(b)-1, fs=False, fe=False)\n"
        "r = run_cases([cs], want_ops=False)[0]; o, _ = oracle_for(cs, False)\n"
        "assert r.key() == o.key() and o.status == 0 and o.length > 8000, (r.key(), o.key())\n"
    ) % (_os.path.dirname(DIAG_LIB).rsplit("/", 1)[0], _os.path.dirname(_os.path.abspath(__file__)))
    env = dict(_os.environ, GAMDP_DIAG_SKIP_TRACEBACK="1", GAMDP_DIAG_COUNT_MAT="1", GAMDP_DIAG_NO_DIRFREE="1",
               GAMDP_DIAG_FORCE_N="1")
    env.pop("GAMDP_LIB", None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_golden_small_cases_with_ops():
    items = G.l0_cases()
    res = run_cases([c for _, c, _ in items], want_ops=True)
    bad = []
    for (name, c, e), r in zip(items, res):
        if r.key() != G.expect_key(e) or (e.get("ops") is not None and r.ops != e["ops"]):
            bad.append((name, r.key(), G.expect_key(e)))
    assert not bad, bad[:5]
    assert len(items) >= 650


def test_golden_small_cases_summary_only_path():
    # without ops the kernel takes the run-skipping traceback: same summaries required
    items = G.l0_cases()
    res = run_cases([c for _, c, _ in items], want_ops=False)
    bad = [(name, r.key(), G.expect_key(e)) for (name, c, e), r in zip(items, res) if r.key() != G.expect_key(e)]
    assert not bad, bad[:5]


@pytest.mark.parametrize("seed", range(6))
def test_random_cases_vs_oracle(seed):
    bands = [(0, 1, 2, 5, 8, 20, 150), (3, 31, 32, 63, 64, 95, 96), (150, 151, 160, 161, 287, 288), (512, 300, 543, 20)][seed % 4]
    cases = _cases.cases(9000 + seed, 500, max_len=400 if seed < 4 else 1500, bands=bands)
    for want_ops in (True, False):
        res = run_cases(cases, want_ops=want_ops)
        bad = []
        n_ok = 0
        for cs, r in zip(cases, res):
            o, ops = oracle_for(cs, want_ops)
            if o.status == O.INVALID:
                continue
            if r.key() != o.key() or (want_ops and r.ops != ops):
                bad.append((cs, r.key(), o.key()))
            n_ok += o.status == O.OK
        assert not bad, bad[:3]
        assert n_ok > 200


def test_begin_a_at_or_past_the_end_of_a():
    """Windows that start at / past the end of a (the reference's row bound wraps, every cell lies outside a): resolved
    on the host from the end-cell scan rules, never launched; the far-out ones sit on the LAST pair of the set."""
    cases = _cases.beyond_cases(32, 1200, bands=(0, 1, 5, 20, 150, 512))
    res = run_cases(cases, want_ops=False)
    bad, stats = [], {}
    for cs, r in zip(cases, res):
        o, _ = oracle_for(cs, False)
        if o.status == O.INVALID:
            assert r.status == O.INVALID
            continue
        stats[o.status] = stats.get(o.status, 0) + 1
        if r.key() != o.key():
            bad.append((cs, r.key(), o.key()))
    assert not bad, bad[:3]
    assert stats.get(O.OUT_OF_RANGE, 0) > 50 and stats.get(O.EMPTY, 0) > 50, stats


def test_medium_pairs_all_kernel_variants():
    rng = random.Random(12)
    cases = []
    for n, band, nfrac in ((3000, 150, 0.0), (3000, 150, 0.01), (2500, 512, 0.0), (2500, 512, 0.02), (4000, 20, 0.0),
                           (3000, 64, 0.0), (2000, 100, 0.01), (3500, 250, 0.0), (1800, 400, 0.005), (6000, 543, 0.0)):
        a, b = _cases.related_pair(rng, n, n_frac=nfrac)
        cases.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1,
                          fs=False, fe=False))
        # a windowed variant with tails on both sides
        cases.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=200, end_a=len(a) - 300, begin_b=190,
                          end_b=len(b) - 310, fs=bool(n % 2), fe=bool(band % 2)))
    for want_ops in (True, False):
        res = run_cases(cases, want_ops)
        for cs, r in zip(cases, res):
            o, ops = oracle_for(cs, want_ops)
            assert r.key() == o.key(), (cs["band"], len(cs["a"]), r.key(), o.key())
            if want_ops:
                assert r.ops == ops
    # edit strings for every other call only: in the two-task kernel one task of a wavefront then walks step by step while
    # its partner takes the side-by-side path alone
    flags = [k % 2 == 0 for k in range(len(cases))]
    for cs, r, f in zip(cases, run_cases(cases, flags), flags):
        o, ops = oracle_for(cs, True)
        assert r.key() == o.key() and r.ops == (ops if f else None), (cs["band"], len(cs["a"]))


def test_batches_with_a_small_scratch_arena():
    """gamdp_ctx_set_arena_bytes: with 16 MB a batch of 36 calls of 2 - 6 kb goes through in many launches of a few slots each
    (long calls peeled off into launches of their own), and the results are the same as with the whole HBM; with 64 KB not one
    task fits and the call fails loudly instead of running anything."""
    from _gpu import ctx
    rng = random.Random(31)
    cases = []
    for k in range(36):
        n = rng.randrange(2000, 6000)
        band = (150, 512, 64, 150)[k % 4]
        a, b = _cases.related_pair(rng, n, n_frac=0.01 if k % 5 == 0 else 0.0)
        cases.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False))
    c = ctx()
    try:
        c.set_arena_bytes(16 << 20)
        res = run_cases(cases, False)
        c.set_arena_bytes(64 << 10)
        with pytest.raises(gam.GamdpError, match="arena"):
            run_cases(cases[:4], False)
    finally:
        c.set_arena_bytes(0)
    for cs, r in zip(cases, res):
        o, _ = oracle_for(cs, False)
        assert r.key() == o.key(), (cs["band"], len(cs["a"]), r.key(), o.key())
    assert [r.key() for r in run_cases(cases, False)] == [r.key() for r in res]   # (and the context is back on its automatic budget)


def test_golden_large_synthetic_pairs():
    """50 kb pairs at band 150 / 512 from the benchmark generator; expected values come from the reference."""
    c = ctx()
    gold = G.load("l0_large.json")
    seqs, calls_meta = [], []
    for d in gold:
        m, s = api.synth_pair(d["k"], d["len"])
        assert len(s) == d["slave_len"]
        assert zlib.crc32(api.decode(m).encode()) == d["a_crc32"]
        seqs += [m, s]
    sset = gam.SequenceSet(c, seqs, ascii=False)
    calls = [(sset.contig(2 * i), 0, d["len"] - 1, sset.contig(2 * i + 1), 0, d["slave_len"] - 1) for i, d in enumerate(gold)]
    bsw = gam.BandedSmithWaterman(c)
    for want_ops in (False, True):
        res = bsw.find_alignments(calls, want_ops=want_ops, bands=[d["band"] for d in gold])
        for d, r in zip(gold, res):
            assert r.key() == G.expect_key(d["expect"]), (d["k"], d["band"])
            assert r.cells == min(d["slave_len"], d["len"] + d["band"]) * (2 * d["band"] + 1)
            if want_ops:
                G.check_ops(d["expect"], r.ops)


def _adversarial_batches():
    """The reference-generated adversarial vectors arranged so that the multi-task kernels pack them: (1) all of them,
    neighbours in the launch's longest-first order = adversarial with adversarial; (2) every one next to an ordinary
    related pair with exactly as many rows (equal predicted cells keep the caller's order: they share a wavefront)."""
    items = G.adversarial_cases()
    rng = random.Random(77)
    mixed = []
    for d, c, e in items:
        a = _cases.rand_seq(rng, len(c["a"]))
        b = (_cases.mutate(rng, a) + _cases.rand_seq(rng, 64))[:len(c["b"])]
        mixed.append((d, c, e))
        mixed.append((None, dict(a=a.encode(), b=b.encode(), band=c["band"], begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1,
                                 fs=False, fe=False), None))
    return items, mixed


def test_adversarial_golden_vectors_through_the_packed_kernels():
    """low-complexity, unrelated and band-edge-hugging pairs of 3-20 kb are what the range argument of
    the packed-f16 blocks has to survive.  Band 512 runs two tasks per wavefront here (k_align_p); band 150 takes the
    eight-task kernel in the child process below (GAMDP_QUAD_MIN=1).  Summaries and edit-string CRCs are the reference's."""
    items, mixed = _adversarial_batches()
    assert len(items) % 2 == 0
    for batch in (items, mixed):
        cases = [c for _, c, _ in batch]
        for want_ops in (False, True):
            res = run_cases(cases, want_ops=want_ops)
            for (d, c, e), r in zip(batch, res):
                if e is None:     # the ordinary partner: against the oracle
                    o, ops = oracle_for(c, want_ops)
                    assert r.key() == o.key() and (not want_ops or r.ops == ops), ("partner", c["band"], len(c["a"]))
                    continue
                assert r.key() == G.expect_key(e), (d["kind"], d["n"], d["band"], r.key(), G.expect_key(e))
                if want_ops:
                    G.check_ops(e, r.ops)


def test_adversarial_golden_vectors_eight_and_four_task_kernels_in_a_fresh_process():
    """The same vectors with every band-150 call forced through the eight-task packed kernel (k_align_o)."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_QUAD_MIN"):
        pytest.skip("already inside the child")
    env = dict(os.environ, GAMDP_QUAD_MIN="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "adversarial_golden_vectors_through"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2000:]


def test_range_assertion_of_the_packed_blocks_in_the_diagnostics_build():
    """libgamdp_diag.so checks, before every re-centring of a packed-f16 block, that each live value lies inside +-2040
    (kernel_pair.inc) and reports a violation as GAMDP_ST_DIAG_RANGE (9) -- which no expected result carries.  The
    adversarial vectors, the long synthetic pairs, the unequal partners and the strip stress cases run on that build,
    at band 512 through the two-task kernel and (GAMDP_QUAD_MIN=1) at band 150 through the eight-task kernel."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_LIB"):
        pytest.skip("already inside a child with a chosen library")
    for extra in ({}, {"GAMDP_QUAD_MIN": "1"}):
        env = dict(os.environ, GAMDP_LIB=DIAG_LIB, **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                            "adversarial_golden_vectors_through or golden_large or pairs_of_unequal or direction_free or band150_stress_cases or packed_top_blocks or window_cases"],
                           env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2000:]


def test_reverse_complement_and_suffix_views():
    """a_rc / b_rc / *_off must equal aligning explicitly reverse-complemented / chopped copies."""
    rng = random.Random(31)
    c = ctx()
    a, b = _cases.related_pair(rng, 1500, n_frac=0.01)
    ca, cb = api.encode(a), api.encode(b)
    sset = gam.SequenceSet(c, [ca, cb, api.reverse_complement(ca), api.reverse_complement(cb), ca[300:], cb[280:]], ascii=False)
    bsw = gam.BandedSmithWaterman(c, 150)
    A, B, Arc, Brc, Achop, Bchop = (sset.contig(i) for i in range(6))
    pairs = [
        ((sset.contig(0, rc=True), 0, len(ca) - 1, sset.contig(1, rc=True), 0, len(cb) - 1), (Arc, 0, len(ca) - 1, Brc, 0, len(cb) - 1)),
        ((sset.contig(0, off=300), 0, len(ca) - 301, sset.contig(1, off=280), 0, len(cb) - 281, True, False),
         (Achop, 0, len(ca) - 301, Bchop, 0, len(cb) - 281, True, False)),
        ((A, 10, 900, sset.contig(1, rc=True), 5, 800), (A, 10, 900, Brc, 5, 800)),
    ]
    for view_call, copy_call in pairs:
        r1, r2 = bsw.find_alignments([view_call, copy_call], want_ops=True)
        assert r1.key() == r2.key() and r1.ops == r2.ops
    # and against the oracle for the rc case
    o, ops = O.oracle_align(api.reverse_complement(ca), api.reverse_complement(cb), 150, 0, len(ca) - 1, 0, len(cb) - 1)
    r = bsw.find_alignment(sset.contig(0, rc=True), 0, len(ca) - 1, sset.contig(1, rc=True), 0, len(cb) - 1, want_ops=True)
    assert r.key() == o.key() and r.ops == ops


_n_edge_cases = _cases.n_edge_cases   # (tests/_cases.py: the L1 seam runs the same cases as merge blocks, test_gpu_l1_parity.py)


def test_one_n_around_every_edge_of_the_window():
    """The N-aware kernels are chosen by the WINDOW a call touches, not by the contig (gamdp_host.cpp prepare_task, SeqSet::window_has_n):
    one N at every distance around the first and the last base the DP touches on a, and around the window on b, with the alignment
    running along the band's first / last column so that those bases are on the path -- a window computed too small would send the
    call to an N-free kernel, which reads an A there.  Plain views, reverse-complement views and chopped views; band 150 (in the
    GAMDP_QUAD_MIN=1 children the four- / eight-task kernels) and band 512; against the oracle."""
    c = ctx()
    for band in ((150,) if _os.environ.get("GAMDP_QUAD_MIN") else (150, 512)):
        cases = _n_edge_cases(band)
        seqs, calls, want = [], [], []
        for k, (a, b, ba, ea, bb, eb, tag) in enumerate(cases):
            ca, cb = api.encode(a), api.encode(b)
            mode = k % 3
            if mode == 0:      # plain
                seqs += [ca, cb]
                view = dict()
            elif mode == 1:    # the set holds the reverse complements: the call asks for the rc view, which is a and b again
                seqs += [api.reverse_complement(ca), api.reverse_complement(cb)]
                view = dict(rc=True)
            else:              # the set holds 37 / 11 more bases in front: the call chops them off
                seqs += [api.encode(b"ACGTN" * 7 + b"AC") + ca, api.encode(b"GNNTACGTACG") + cb]
                view = dict(off=(37, 11))
            calls.append((len(seqs) - 2, view, ba, ea, bb, eb))
            want.append(O.oracle_align(ca, cb, band, ba, ea, bb, eb))
        sset = gam.SequenceSet(c, seqs, ascii=False)
        bsw = gam.BandedSmithWaterman(c, band)
        args = []
        for i, view, ba, ea, bb, eb in calls:
            if "rc" in view:
                A, B = sset.contig(i, rc=True), sset.contig(i + 1, rc=True)
            elif "off" in view:
                A, B = sset.contig(i, off=view["off"][0]), sset.contig(i + 1, off=view["off"][1])
            else:
                A, B = sset.contig(i), sset.contig(i + 1)
            args.append((A, ba, ea, B, bb, eb))
        res = bsw.find_alignments(args, want_ops=True)
        for k, (r, (o, ops)) in enumerate(zip(res, want)):
            assert r.key() == o.key(), (band, cases[k][6], k % 3, r.key(), o.key())
            assert r.ops == ops, (band, cases[k][6], k % 3)
        sset.close()


def test_a_window_too_small_is_noticed():
    """Fault injection (diagnostics build, GAMDP_DIAG_N_WINDOW_SHRINK): with the windows 66 + 256 bases too small on either side -- past
    the margin and the granularity of the N counts -- the test above must fail: it is the one that would catch a wrong window."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_DIAG_N_WINDOW_SHRINK"):
        pytest.skip("already inside the child")
    env = dict(os.environ, GAMDP_LIB=DIAG_LIB, GAMDP_DIAG_N_WINDOW_SHRINK=str(64 + 2 + 256))
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k", "one_n_around"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "AssertionError" in r.stdout, r.stdout[-2500:] + r.stderr[-1500:]
    env = dict(os.environ, GAMDP_LIB=DIAG_LIB)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k", "one_n_around"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-1500:]


def test_n_by_contig_in_a_fresh_process():
    """GAMDP_N_BY_CONTIG=1: a call takes the N-aware kernels whenever one of its contigs holds an N (rounds 1-3): same results."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_N_BY_CONTIG"):
        pytest.skip("already inside the child")
    env = dict(os.environ, GAMDP_N_BY_CONTIG="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "one_n_around or window_cases or random_cases or reverse_complement_and"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2000:]


def test_the_limits_that_remain_for_wide_bands():
    """The reference takes any band (banded_smith_waterman.hpp:66); since round 6 so does the library (k_align_w for bands > 543,
    gamdp_wide.hip).  What is left (include/gamdp.h): a band beyond GAMDP_MAX_BAND = 2^20 is refused (GAMDP_ENOTSUP), and a
    wide-band call whose matrix does not fit the scratch arena fails with GAMDP_ENOMEM -- loudly, both."""
    c = ctx()
    sset = gam.SequenceSet(c, [b"ACGT" * 10, b"ACGT" * 10])
    with pytest.raises(gam.GamdpError, match="GAMDP_MAX_BAND"):
        gam.BandedSmithWaterman(c, (1 << 20) + 1).find_alignment(sset.contig(0), 0, 39, sset.contig(1), 0, 39)
    # the widest band the library takes, and a band no lane geometry divides, on tiny inputs: bit-exact
    cases = [dict(a=b"ACGTTGCA" * 5, b=b"ACGTTGCA" * 5, band=1 << 20, begin_a=0, end_a=39, begin_b=0, end_b=4, fs=False, fe=False),
             dict(a=b"ACGTTGCAAT" * 30, b=b"GTTGCAATAC" * 30, band=100003, begin_a=7, end_a=299, begin_b=2, end_b=290, fs=False, fe=True)]
    for cs, r in zip(cases, run_cases(cases)):
        o, ops = oracle_for(cs)
        assert o.status == O.OK and r.key() == o.key() and r.ops == ops
    assert {x["kernel"] for x in c.launch_info()} == {"k_align_w"}
    rng = random.Random(77)
    a, b = _cases.related_pair(rng, 9000)
    big = gam.SequenceSet(c, [a.encode(), b.encode()])
    try:
        c.set_arena_bytes(64 << 20)   # 9 000 rows x 4 097 columns x 4 B = 147 MB
        with pytest.raises(gam.GamdpError, match="arena"):
            gam.BandedSmithWaterman(c, 2048).find_alignment(big.contig(0), 0, len(a) - 1, big.contig(1), 0, len(b) - 1)
    finally:
        c.set_arena_bytes(0)
    r = gam.BandedSmithWaterman(c, 2048).find_alignment(big.contig(0), 0, len(a) - 1, big.contig(1), 0, len(b) - 1)
    o, _ = oracle_for(dict(a=a.encode(), b=b.encode(), band=2048, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False), False)
    assert r.key() == o.key() and o.status == O.OK


@pytest.mark.parametrize("seed", range(2))
def test_random_cases_vs_oracle_wide_bands(seed):
    """test_random_cases_vs_oracle for bands beyond the systolic kernels (k_align_w): every status, force flags, N, ragged windows."""
    # (seed 1: narrow bands in the same batch -- the wide kernel's launch next to the systolic kernels' in one call)
    cases = _cases.cases(9100 + seed, 300, max_len=400 if seed == 0 else 1500, bands=(544, 600, 1000, 2048) if seed == 0 else (544, 600, 1000, 2048, 150, 20, 512))
    for want_ops in (True, False):
        res = run_cases(cases, want_ops=want_ops)
        bad = []
        n_ok = 0
        for cs, r in zip(cases, res):
            o, ops = oracle_for(cs, want_ops)
            if o.status == O.INVALID:
                continue
            if r.key() != o.key() or (want_ops and r.ops != ops):
                bad.append((cs, r.key(), o.key()))
            n_ok += o.status == O.OK
        assert not bad, bad[:3]
        assert n_ok > 100
    assert "k_align_w" in {r["kernel"] for r in ctx().launch_info()}


def test_empty_batch_and_degenerate_sequences():
    c = ctx()
    assert gam.BandedSmithWaterman(c).find_alignments([]) == []
    cases = [dict(a=b"A", b=b"A", band=150, begin_a=0, end_a=0, begin_b=0, end_b=0, fs=False, fe=False),
             dict(a=b"A", b=b"C", band=0, begin_a=0, end_a=0, begin_b=0, end_b=0, fs=False, fe=False),
             dict(a=b"ACGTACGTAC", b=b"A", band=5, begin_a=3, end_a=9, begin_b=0, end_b=5, fs=True, fe=True),
             dict(a=b"N", b=b"N", band=2, begin_a=0, end_a=3, begin_b=0, end_b=0, fs=False, fe=False)]
    for cs, r in zip(cases, run_cases(cases)):
        o, ops = oracle_for(cs)
        assert r.key() == o.key() and r.ops == ops


def test_library_generated_synthetic_set_equals_python_uploaded_one():
    """gamdp_seqset_create_synth (bench path) must hold exactly the sequences gamdp_synth_pair produces."""
    c = ctx()
    n, length = 6, 3000
    lib_set = gam.SequenceSet.synthetic(c, 40, n, length)
    pairs = [api.synth_pair(40 + k, length) for k in range(n)]
    py_set = gam.SequenceSet(c, [x for p in pairs for x in p], ascii=False)
    assert lib_set.lengths == py_set.lengths
    bsw = gam.BandedSmithWaterman(c, 150)
    r1 = bsw.find_alignments([(lib_set.contig(2 * k), 0, length - 1, lib_set.contig(2 * k + 1), 0, lib_set.lengths[2 * k + 1] - 1) for k in range(n)], want_ops=True)
    r2 = bsw.find_alignments([(py_set.contig(2 * k), 0, length - 1, py_set.contig(2 * k + 1), 0, py_set.lengths[2 * k + 1] - 1) for k in range(n)], want_ops=True)
    assert [(a.key(), a.ops) for a in r1] == [(b.key(), b.ops) for b in r2]
    with pytest.raises(gam.GamdpError):  # no host copy of the bases -> reverse-complement views are refused
        bsw.find_alignment(lib_set.contig(0, rc=True), 0, 10, lib_set.contig(1), 0, 10)


def test_row_cap_and_long_tasks_against_oracle():
    """BSW_MAX_ALIGNMENT: the reference fills at most 500 000 rows (banded_smith_waterman.cc:95).  A 520 kb pair at
    band 5 must stop there, bit for bit like the oracle; plus a 120 kb pair at the live band 150 (7 500 blocks)."""
    import _cases
    rng = random.Random(2)
    a, b = _cases.related_pair(rng, 520000, div=0.6)
    a2, b2 = _cases.related_pair(rng, 120000, n_frac=0.001)
    cases = [dict(a=a.encode(), b=b.encode(), band=5, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False),
             dict(a=a2.encode(), b=b2.encode(), band=150, begin_a=0, end_a=len(a2) - 1, begin_b=0, end_b=len(b2) - 1, fs=False, fe=False)]
    for want_ops in (False, True):
        res = run_cases(cases, want_ops=want_ops)
        for cs, r in zip(cases, res):
            o, ops = oracle_for(cs, want_ops)
            assert r.key() == o.key(), (cs["band"], r.key(), o.key())
            if want_ops:
                assert r.ops == ops
    assert res[0].cells == 500000 * 11


def test_n_aware_kernels_on_every_case_in_a_fresh_process():
    """GAMDP_DIAG_FORCE_N routes N-free inputs through the N-aware (v_dot8) kernels too: the random, medium and
    large-pair tests of this file must still be bit-exact.  The switch is read once per process, hence the child."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_DIAG_FORCE_N"):
        pytest.skip("already inside the forced-N child")
    env = dict(os.environ, GAMDP_DIAG_FORCE_N="1", GAMDP_LIB=DIAG_LIB)  # only the diagnostics build has the switch
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__),
                        "-k", "random_cases or medium_pairs or golden_large or row_cap"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_one_task_per_wavefront_band512_kernel_in_a_fresh_process():
    """Band-512 batches without N run two tasks per wavefront with their fast blocks in packed f16 (kernel_pair.inc) as
    soon as there are two tasks; GAMDP_NO_PAIR=1 keeps the one-task direction-free kernel, which single-task batches
    (and odd leftovers) still use: the band-512 cases of this file must be bit-exact on it too."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_NO_PAIR"):
        pytest.skip("already inside the no-pair child")
    env = dict(os.environ, GAMDP_NO_PAIR="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "direction_free or golden_large or medium_pairs or random_cases_vs_oracle or pairs_of_unequal"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2000:]


def test_two_task_kernel_with_one_after_the_other_walks_in_a_fresh_process():
    """Short band-512 launches (every batch of this file) walk the two tasks of a wavefront side by side; launches of more
    than two rounds keep the one-task walk.  GAMDP_SIDE_WALK_ROUNDS=0 applies that choice to the batches of this file."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_SIDE_WALK_ROUNDS"):
        pytest.skip("already inside the child")
    env = dict(os.environ, GAMDP_SIDE_WALK_ROUNDS="0")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "direction_free or golden_large or medium_pairs or random_cases_vs_oracle or pairs_of_unequal"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2000:]


def test_pieces_over_two_contexts_in_a_fresh_process():
    """Batches of >= 65 536 small calls go through in four pieces on two host threads / contexts (host work of one piece hidden
    behind the other's kernel); GAMDP_CHUNK_MIN=16 applies that to the small batches of this file: same results."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_CHUNK_MIN"):
        pytest.skip("already inside the chunked child")
    env = dict(os.environ, GAMDP_CHUNK_MIN="16")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "random_cases_vs_oracle or medium_pairs or golden_small_cases_summary or begin_a_at or band150_stress or forced_chunked_batch"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2000:]


def test_a_forced_chunked_batch_goes_through_four_pieces():
    """Inside the chunked child only: what the library says it launched carries the pieces' numbers (both contexts were used)."""
    import os
    if not os.environ.get("GAMDP_CHUNK_MIN"):
        pytest.skip("only inside the chunked child")
    c = ctx()
    sset = gam.SequenceSet.synthetic(c, 4242, 48, 1500)
    tasks = (L.Task * 48)()
    for k in range(48):
        t = tasks[k]
        t.a_id, t.b_id, t.band = 2 * k, 2 * k + 1, 150
        t.begin_a, t.end_a, t.begin_b, t.end_b = 0, 1499, 0, sset.lengths[2 * k + 1] - 1
    out = (L.Result * 48)()
    assert c.lib.gamdp_align_batch(c.handle, sset.handle, sset.handle, tasks, 48, out, None) == 0, c.last_error()
    info = c.launch_info()
    assert sorted({r["piece"] for r in info}) == [0, 1, 2, 3], info
    assert sum(r["tasks"] for r in info) == 48 and all(out[k].status == L.ST_OK for k in range(48))
    sset.close()


def test_pairs_of_unequal_tasks_band512():
    """Two tasks per wavefront: partners of very different length / window / flags (the packed range is the common run of
    fast blocks; with none the pair falls back to directions for every cell), an odd task count, 3 to 40 kb."""
    rng = random.Random(512)
    cases = []
    for n, sub, ins, dele in ((40000, 0.03, 0.01, 0.01), (9000, 0.03, 0.01, 0.01), (9100, 0.02, 0.03, 0.001), (3000, 0.03, 0.01, 0.01),
                              (2500, 0.0, 0.0, 0.0), (700, 0.05, 0.01, 0.01), (22000, 0.02, 0.001, 0.03), (21000, 0.04, 0.012, 0.012),
                              (15000, 0.03, 0.01, 0.01)):
        a = _cases.rand_seq(rng, n)
        b = _cases.mutate(rng, a, sub, ins, dele) or "A"
        cases.append(dict(a=a.encode(), b=b.encode(), band=512, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False))
    a = _cases.rand_seq(rng, 30000)
    b = _cases.mutate(rng, a[4000:], 0.02, 0.01, 0.01)
    for ba, ea, bb, eb, fs, fe in ((4000, len(a) - 1, 0, len(b) - 1, True, False), (3900, len(a) - 1, 0, len(b) - 1, False, True),
                                   (4000, 18000, 0, len(b) - 1, False, False), (4200, len(a) + 100, 100, 22000, False, False)):
        cases.append(dict(a=a.encode(), b=b.encode(), band=512, begin_a=ba, end_a=ea, begin_b=bb, end_b=eb, fs=fs, fe=fe))
    assert len(cases) % 2 == 1
    n_ok = 0
    for want_ops in (False, True):
        res = run_cases(cases, want_ops=want_ops)
        for k, (cs, r) in enumerate(zip(cases, res)):
            o, ops = oracle_for(cs, want_ops)
            assert r.key() == o.key(), (k, len(cs["a"]), r.key(), o.key())
            if want_ops:
                assert r.ops == ops, k
            n_ok += o.status == 0
    assert n_ok >= 24


def band150_stress_cases():
    """Band-150 cases for both kernel shapes (one task / four tasks per wavefront): pairs from 40 bases to 60 kb in one
    batch (so the four tasks of a wavefront differ in length, top / end block ranges and direction-free range), drifting
    paths, N runs, deep windows, force flags, unrelated sequences, windows past the end of a."""
    rng = random.Random(150)
    cases = []
    for n, sub, ins, dele, nf in ((60000, 0.02, 0.01, 0.01, 0.0), (59000, 0.02, 0.02, 0.001, 0.0), (30000, 0.03, 0.001, 0.02, 0.0),
                                  (30500, 0.0, 0.0, 0.0, 0.0), (12000, 0.03, 0.01, 0.01, 0.001), (12100, 0.03, 0.01, 0.01, 0.0),
                                  (5000, 0.03, 0.01, 0.01, 0.0), (5010, 0.03, 0.01, 0.01, 0.0), (4990, 0.02, 0.0, 0.0, 0.01),
                                  (2000, 0.03, 0.01, 0.01, 0.0), (700, 0.03, 0.01, 0.01, 0.0), (300, 0.05, 0.02, 0.02, 0.0),
                                  (160, 0.0, 0.0, 0.0, 0.0), (40, 0.1, 0.0, 0.0, 0.0), (20000, 0.04, 0.012, 0.012, 0.0005)):
        a = _cases.rand_seq(rng, n, nf)
        b = _cases.mutate(rng, a, sub, ins, dele) or "A"
        cases.append(dict(a=a.encode(), b=b.encode(), band=150, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False))
    a = _cases.rand_seq(rng, 26000)
    b = _cases.mutate(rng, a[3000:], 0.02, 0.01, 0.01)
    for ba, ea, bb, eb, fs, fe in ((3000, len(a) - 1, 0, len(b) - 1, True, False), (2900, len(a) - 1, 0, len(b) - 1, False, True),
                                   (3000, 14000, 0, len(b) - 1, False, False), (3100, len(a) + 100, 100, 20000, False, False),
                                   (3000, len(a) - 1, 0, 6000, True, True), (0, len(a) - 1, 0, len(b) - 1, False, False)):
        cases.append(dict(a=a.encode(), b=b.encode(), band=150, begin_a=ba, end_a=ea, begin_b=bb, end_b=eb, fs=fs, fe=fe))
    for _ in range(9):   # more short / medium pairs so that several wavefronts are mixed
        n = rng.choice([90, 450, 1500, 3300, 8000])
        a, b = _cases.related_pair(rng, n, n_frac=rng.choice([0.0, 0.0, 0.01]))
        cases.append(dict(a=a.encode(), b=b.encode(), band=150, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False))
    return cases


def test_band150_stress_cases():
    cases = band150_stress_cases()
    n_ok = 0
    for want_ops in (False, True):
        res = run_cases(cases, want_ops=want_ops)
        for k, (cs, r) in enumerate(zip(cases, res)):
            o, ops = oracle_for(cs, want_ops)
            assert r.key() == o.key(), (k, len(cs["a"]), r.key(), o.key())
            if want_ops:
                assert r.ops == ops, k
            n_ok += o.status == 0
    assert n_ok >= 50
    # edit strings for some calls of a batch only: in the eight-task kernel those walk one at a time while their
    # wavefront's other tasks walk side by side (kernel_walk.inc)
    flags = [k % 3 == 1 for k in range(len(cases))]
    res = run_cases(cases, want_ops=flags)
    for k, (cs, r) in enumerate(zip(cases, res)):
        o, ops = oracle_for(cs, True)
        assert r.key() == o.key(), (k, r.key(), o.key())
        assert r.ops == (ops if flags[k] else None), k


def _top_block_batches():
    """Batches built for the packed top blocks (pair_top_range): every call of a batch starts at the same begin_a -- 0, inside
    the band's left triangle, at its edge, just past it (no top blocks at all) -- so that whichever two (eight) calls share a
    wavefront qualify; windows on b, ends past the contigs, alignments whose path starts at pos == 0 far down the triangle
    (a has a long prefix b lacks) or at row 0 far to the right (b has a long prefix a lacks), an early end of a.  Plus batches
    whose calls DIFFER in begin_a and hold force_start / force_end calls (the calls of the live driver, PctgBuilder.cc:1535-1611,
    1662-1669): since round 5 those wavefronts take the per-task form of the packed top blocks (pair_top_range<.., MIXED, FS>)."""
    import _cases
    rng = random.Random(4711)
    batches = []
    for band, begins in ((512, (0, 1, 37, 300, 511, 512, 513, 700)), (150, (0, 1, 37, 149, 150, 151, 400))):
        for begin_a in begins:
            cases = []
            for k in range(8):
                n = rng.choice((2600, 3100, 4100, 6100)) if band == 512 else rng.choice((900, 1300, 2100, 4100))
                a, b = _cases.related_pair(rng, n)
                kind = k % 4
                if kind == 1:
                    a = _cases.rand_seq(rng, rng.randint(40, band - 20)) + a      # the path enters at pos == 0, rows down the triangle
                elif kind == 2:
                    b = _cases.rand_seq(rng, rng.randint(40, band - 20)) + b      # the path enters in row 0, right of the band's centre
                elif kind == 3:
                    a = a[:len(a) - rng.randint(100, 600)]                         # a ends early: the end cell lies on the pos == end_a anti-diagonal
                ba = min(begin_a, len(a) - 1)
                bb = rng.choice((0, 0, 13, 250))
                cases.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=ba, end_a=len(a) - 1 + rng.choice((0, 0, 40)),
                                  begin_b=bb, end_b=len(b) - 1 - rng.choice((0, 0, 7)), fs=False, fe=(k == 5)))
            batches.append(cases)
        # end_a inside the band's first rows: the pos == end_a anti-diagonal (an END capture) runs through the top blocks, which then
        # stay int32 (parity_band512.py found the packed version of this, round 4)
        early = []
        for k in range(8):
            a, b_ = _cases.related_pair(rng, 2600 if band == 512 else 1400)
            ea = rng.choice((band // 2, band, band + 20, 2 * band + 7))
            early.append(dict(a=a.encode(), b=b_.encode(), band=band, begin_a=0, end_a=ea, begin_b=rng.choice((0, 46)), end_b=len(b_) - 1 - rng.choice((0, 79)),
                              fs=False, fe=False))
        batches.append(early)
        mixed = []
        for k in range(8):
            a, b = _cases.related_pair(rng, 3300 if band == 512 else 1500)
            mixed.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=(0, 5, 0, 90)[k % 4], end_a=len(a) - 1, begin_b=0,
                              end_b=len(b) - 1, fs=(k >= 6), fe=False))
        batches.append(mixed)
        # wavefronts of mixed begin_a (inside the triangle, at its edge, far past it: no top block at all), force_start calls whose
        # paths enter at pos == 0 beyond row 10 (where a forced start has no left source) and before it, force_end calls, windows on b
        for rep in range(3):
            mix = []
            for k in range(16):
                n = rng.choice((2600, 3300, 4100)) if band == 512 else rng.choice((900, 1500, 2300, 4100))
                a, b = _cases.related_pair(rng, n)
                kind = rng.randrange(5)
                if kind == 1:
                    a = _cases.rand_seq(rng, rng.randint(3, band - 20)) + a      # the path enters at pos == 0 some rows down (3 .. band - 20: either side of row 10)
                elif kind == 2:
                    b = _cases.rand_seq(rng, rng.randint(3, band - 20)) + b
                ba = rng.choice((0, 0, 1, 7, 12, band // 3, band - 17, band - 1, band, band + 1, band + 40, 3 * band))
                ba = min(ba, len(a) - 1)
                fs = rng.random() < (0.5 if rep == 2 else 0.25)
                mix.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=ba, end_a=len(a) - 1 + rng.choice((0, 0, 40)),
                                begin_b=rng.choice((0, 0, 13, 250)), end_b=len(b) - 1 - rng.choice((0, 0, 7)),
                                fs=fs, fe=(not fs and rng.random() < 0.2)))
            batches.append(mix)
    return batches


def test_packed_top_blocks():
    """The blocks that hold cells with pos <= 0 run packed too when the calls of a wavefront share begin_a (round 4,
    kernel_pair.inc pair_top_range; banded_smith_waterman.cc:111-133,143-155 is what must survive: the pos == 0 rules, the
    zero-initialised matrix left of it).  Against the oracle, summaries and edit strings; the band-150 batches reach the
    eight-task kernel in the GAMDP_QUAD_MIN=1 child of test_four_tasks_per_wavefront_kernels, every batch the int32 top
    blocks once more in the GAMDP_NO_PACKED_TOP=1 child below."""
    n_ok = 0
    for cases in _top_block_batches():
        for want_ops in (False, True):
            res = run_cases(cases, want_ops=want_ops)
            for k, (cs, r) in enumerate(zip(cases, res)):
                o, ops = oracle_for(cs, want_ops)
                assert r.key() == o.key(), (cs["band"], cs["begin_a"], k, len(cs["a"]), r.key(), o.key())
                assert (not want_ops) or r.ops == ops, (cs["band"], cs["begin_a"], k)
                n_ok += o.status == 0
    assert n_ok >= 220


def test_top_blocks_packed_for_a_shared_begin_only_in_a_fresh_process():
    """GAMDP_NO_PACKED_TOP_MIXED=1: wavefronts whose calls differ in begin_a or hold a force_start call keep the int32 top blocks
    (round 4's rule); same results, at both bands and through the eight-task kernel."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_NO_PACKED_TOP_MIXED"):
        pytest.skip("already inside the child")
    for extra in ({}, dict(GAMDP_QUAD_MIN="1")):
        env = dict(os.environ, GAMDP_NO_PACKED_TOP_MIXED="1", **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                            "packed_top_blocks or window_cases"], env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, (extra, r.stdout[-2500:] + r.stderr[-2000:])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_window_cases_on_long_pairs(seed):
    """tests/_cases.py window_cases (the generator of tools/parity_band512.py): random windows on 0.6 - 14 kb pairs at band 512,
    with and without edit strings -- begin_a shared or not, end_a early or late, force flags, N; seed 2 also with a scratch arena
    of a handful of slots.  (Seed 2 holds the call that found round 4's packed top blocks taking a pos == end_a anti-diagonal
    along: begin_a 0, end_a 532, 1 875 rows.)  Band 150 runs the same in the GAMDP_QUAD_MIN=1 child."""
    band = 150 if _os.environ.get("GAMDP_QUAD_MIN") else 512
    cases = _cases.window_cases(seed, band)
    c = ctx()
    if seed == 2:
        c.set_arena_bytes(40 << 20)
    try:
        n = 0
        for want_ops in (False, True):
            res = run_cases(cases, want_ops=want_ops)
            for k, (cs, r) in enumerate(zip(cases, res)):
                o, ops = oracle_for(cs, want_ops)
                if o.status == O.INVALID:
                    continue
                n += 1
                assert r.key() == o.key(), (seed, k, {x: v for x, v in cs.items() if x not in ("a", "b")}, len(cs["a"]), len(cs["b"]), r.key(), o.key())
                assert (not want_ops) or r.ops == ops, (seed, k)
        assert n >= 100
    finally:
        c.set_arena_bytes(0)


@pytest.mark.parametrize("band", [129, 160, 200, 256, 287, 288, 300, 400, 500, 543, 600, 1000, 2048])
def test_generic_bands_on_long_pairs(band):
    """Bands other than the two tuned ones: the generic kernels with 9 (bands 160 - 287) and 17 columns per lane (288 - 543) fill their fast
    blocks without directions since round 5 (band 129: 5 columns per lane, a direction per cell as before) -- a RUNTIME (lane, column) for the band's last column in the direction-free cell
    (do_block_df, kill_c) and in the strips (materialise, CE < 0).  Windowed 0.6 - 14 kb cases (N, force flags, early end_a) and paths
    k columns off the band's middle from one band edge to the other -- the last ones run inside the strip that holds the band's
    last column and the dead lanes behind it; summaries and edit strings against the oracle.  (The short random cases of
    test_random_cases_vs_oracle never reach a direction-free range.)"""
    cases = _cases.window_cases(band, band, count=24) + _cases.displaced_path_cases(band, n=2600)
    n = 0
    for want_ops in (False, True):
        res = run_cases(cases, want_ops=want_ops)
        for k, (cs, r) in enumerate(zip(cases, res)):
            o, ops = oracle_for(cs, want_ops)
            if o.status == O.INVALID:
                continue
            n += 1
            assert r.key() == o.key(), (band, k, {x: v for x, v in cs.items() if x not in ("a", "b")}, len(cs["a"]), len(cs["b"]), r.key(), o.key())
            assert (not want_ops) or r.ops == ops, (band, k)
    assert n >= 60
    info = ctx().launch_info()
    if band > 543:   # round 6: no lane geometry covers it -- the correct-at-any-speed kernel (gamdp_wide.hip), same cases
        assert {r["kernel"] for r in info} == {"k_align_w"}, info
        return
    cols = next(c for c in (2, 3, 5, 9, 17) if 2 * band + 1 <= c * 64)     # gamdp_host.cpp pick_kernel
    assert {r["kernel"] for r in info} == {"k_align<%d,-1,true>" % cols}, info
    assert sum(r["units_dirfree"] for r in info) >= (20 if cols >= 9 else 0), info      # the long ones did run a direction-free range


def _band_with_edge_column(C, k):
    """a band of the C-columns-per-lane generic kernel whose last column sits at in-lane position k = (2 * band) % C"""
    lo, hi = {9: (160, 287), 17: (288, 543)}[C]
    return next(b for b in range(lo + 11, hi + 1) if (2 * b) % C == k)


@pytest.mark.parametrize("C", [9, 17])
def test_every_edge_column_instance_of_the_generic_df_ranges(C):
    """df_range (gamdp_kernel.hip) sends the direction-free blocks of a generic band to the instance of the tuned fast range
    whose COMPILE-TIME edge column is the band's, chosen by (Y - 1) % C: 8 instances for 9 columns per lane, 16 for 17, each in END and
    non-END form (edge column C - 1 takes the runtime path).  One band per residue, paths from one band edge to the other on pairs long
    enough for a direction-free range, edit strings included."""
    seen = set()
    for k in range(C):
        band = _band_with_edge_column(C, k)
        assert (2 * band) % C == k
        cases = _cases.displaced_path_cases(band, n=2300)[::2] + _cases.window_cases(band, band, count=6)
        for want_ops in (False, True):
            res = run_cases(cases, want_ops=want_ops)
            for q, (cs, r) in enumerate(zip(cases, res)):
                o, ops = oracle_for(cs, want_ops)
                if o.status == O.INVALID:
                    continue
                assert r.key() == o.key(), (C, k, band, q, r.key(), o.key())
                assert (not want_ops) or r.ops == ops, (C, k, band, q)
        info = ctx().launch_info()
        assert {r["kernel"] for r in info} == {"k_align<%d,-1,true>" % C}, (band, info)
        assert sum(r["units_dirfree"] for r in info) >= 8, (band, info)
        seen.add(k)
    assert seen == set(range(C))


def test_paths_in_every_strip():
    """tests/_cases.py displaced_path_cases: alignments that run k columns off the middle of the band, from one band edge to the
    other -- every strip of the direction-free kernels gets a walk, the strips at the two edges of a task (whose outer lanes
    belong to the next task or to nobody since the strips are centred on the band's middle column, Tk::sshift) included.
    Band 512 here; band 150 reaches the eight-task and the four-task kernels in the GAMDP_QUAD_MIN=1 children; the strips
    of rounds 1-3 (GAMDP_NO_STRIP_SHIFT=1) in the child below."""
    band = 150 if _os.environ.get("GAMDP_QUAD_MIN") else 512
    cases = _cases.displaced_path_cases(band)
    n = 0
    for want_ops in (False, True):
        res = run_cases(cases, want_ops=want_ops)
        for k, (cs, r) in enumerate(zip(cases, res)):
            o, ops = oracle_for(cs, want_ops)
            assert r.key() == o.key(), (band, k, len(cs["a"]), len(cs["b"]), r.key(), o.key())
            assert (not want_ops) or r.ops == ops, (band, k)
            n += o.status == 0
    assert n >= 50


def test_strips_at_multiples_of_their_width_in_a_fresh_process():
    """GAMDP_NO_STRIP_SHIFT=1: the strips of the direction-free ranges begin at multiples of the strip width (rounds 1-3):
    same results, at band 512 and, through GAMDP_QUAD_MIN=1, in the eight-task kernel."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_NO_STRIP_SHIFT"):
        pytest.skip("already inside the child")
    for extra in ({}, dict(GAMDP_QUAD_MIN="1")):
        env = dict(os.environ, GAMDP_NO_STRIP_SHIFT="1", **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                            "paths_in_every_strip or window_cases or golden_large or medium_pairs"],
                           env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, (extra, r.stdout[-2500:] + r.stderr[-2000:])


def test_wavefronts_filled_up_with_copies():
    """The last wavefront of a multi-task launch is filled up with copies of its last call (TF_PADDING: filled along, no end cell, no
    walk, no result).  301 short N-free band-150 calls through the eight-task kernel (GAMDP_QUAD_MIN=1 child) on a scratch arena of a
    dozen slots -- several rounds, the last wavefront five copies --, and 3 calls alone (one wavefront, five copies); every call
    against the oracle, with and without edit strings."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_TEST_PADDING_CHILD"):
        rng = random.Random(4242)
        cases = []
        for k in range(301):
            a, b = _cases.related_pair(rng, rng.choice([700, 1100, 1600, 2500]))
            cases.append(dict(a=a.encode(), b=b.encode(), band=150, begin_a=rng.choice([0, 0, 3]), end_a=len(a) - 1, begin_b=0,
                              end_b=len(b) - 1, fs=False, fe=False))
        c = ctx()
        c.set_arena_bytes(24 << 20)
        try:
            for batch in (cases, cases[:3]):
                for want_ops in (False, True):
                    res = run_cases(batch, want_ops=want_ops)
                    for k, (cs, r) in enumerate(zip(batch, res)):
                        o, ops = oracle_for(cs, want_ops)
                        assert r.key() == o.key(), (k, len(cs["a"]), r.key(), o.key())
                        assert (not want_ops) or r.ops == ops, k
        finally:
            c.set_arena_bytes(0)
        return
    env = dict(os.environ, GAMDP_QUAD_MIN="1", GAMDP_TEST_PADDING_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k", "wavefronts_filled_up"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2000:]


def test_int32_top_blocks_in_a_fresh_process():
    """GAMDP_NO_PACKED_TOP=1: the packed kernels keep the int32 tagged code for their top blocks (the path of rounds 1-3,
    still taken by wavefronts whose calls differ in begin_a or force their start): same results."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_NO_PACKED_TOP"):
        pytest.skip("already inside the child")
    env = dict(os.environ, GAMDP_NO_PACKED_TOP="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "packed_top_blocks or window_cases or golden_large or medium_pairs or pairs_of_unequal or adversarial_golden_vectors_through"],
                       env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2000:]


def test_four_tasks_per_wavefront_kernels_in_a_fresh_process():
    """GAMDP_QUAD_MIN=1 sends every band-150 call of a batch through the throughput kernels instead of only batches larger
    than the chip's wave slots: eight tasks per wavefront (two quads, fast blocks in packed f16) for contigs without N,
    four tasks per wavefront (16 lanes x 19 columns, direction-free int32 fill) for the rest.  The band-150 cases of this
    file must come out bit-exact; further children keep the four-task int32 kernel on N-free input (GAMDP_NO_PAIR), force
    its N-aware twin (GAMDP_DIAG_FORCE_N, diagnostics build) and switch the direction-free fill off."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_QUAD_MIN"):
        pytest.skip("already inside the four-task child")
    sel = "band150_stress or random_cases or medium_pairs or golden_large or golden_small or begin_a_at or row_cap or packed_top_blocks or window_cases or paths_in_every_strip or one_n_around"
    for extra in ({}, dict(GAMDP_NO_PAIR="1"), dict(GAMDP_DIAG_FORCE_N="1", GAMDP_LIB=DIAG_LIB), dict(GAMDP_DIAG_NO_DIRFREE="1", GAMDP_LIB=DIAG_LIB)):
        env = dict(os.environ, GAMDP_QUAD_MIN="1", **extra)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k", sel],
                           env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, (extra, r.stdout[-2500:] + r.stderr[-2000:])


def test_direction_free_fill_and_strip_materialisation():
    """Band 512 without N runs its fast blocks direction-free and re-creates the directions of a 4-lane strip around
    the path on demand (gamdp_kernel.hip: do_block_df / materialise).  Pairs built to stress that: indel-rich, with a
    net drift of the path across many strips (insertions >> deletions and the reverse), windows that start deep inside
    the sequences, force flags, a second fast run after an anti-diagonal in the middle; each with and without the edit
    string, each also with the feature switched off in a child process."""
    import os, subprocess, sys
    import _cases
    rng = random.Random(99)
    cases = []
    for n, sub, ins, dele in ((30000, 0.02, 0.03, 0.002), (30000, 0.02, 0.002, 0.03), (45000, 0.03, 0.02, 0.02),
                              (12000, 0.0, 0.0, 0.0), (20000, 0.01, 0.04, 0.0005), (20000, 0.05, 0.01, 0.01)):
        a = _cases.rand_seq(rng, n)
        b = _cases.mutate(rng, a, sub, ins, dele)
        cases.append(dict(a=a.encode(), b=b.encode(), band=512, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False))
    a = _cases.rand_seq(rng, 150000, 0.0005)   # long, with a few N runs: the N-aware kernel, 9 000 blocks, 140 groups
    b = _cases.mutate(rng, a, 0.03, 0.01, 0.01)
    cases.append(dict(a=a.encode(), b=b.encode(), band=512, begin_a=0, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=False))
    a = _cases.rand_seq(rng, 26000)
    b = _cases.mutate(rng, a[3000:], 0.02, 0.01, 0.01)
    cases.append(dict(a=a.encode(), b=b.encode(), band=512, begin_a=3000, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=True, fe=False))
    cases.append(dict(a=a.encode(), b=b.encode(), band=512, begin_a=2800, end_a=len(a) - 1, begin_b=0, end_b=len(b) - 1, fs=False, fe=True))
    cases.append(dict(a=a.encode(), b=b.encode(), band=512, begin_a=3000, end_a=14000, begin_b=0, end_b=len(b) - 1, fs=False, fe=False))
    cases.append(dict(a=a.encode(), b=b.encode(), band=512, begin_a=3300, end_a=len(a) + 100, begin_b=100, end_b=20000, fs=False, fe=False))
    n_ok = 0
    for want_ops in (False, True):
        res = run_cases(cases, want_ops=want_ops)
        for k, (cs, r) in enumerate(zip(cases, res)):
            o, ops = oracle_for(cs, want_ops)
            assert r.key() == o.key(), (k, r.key(), o.key())
            if want_ops:
                assert r.ops == ops, k
            n_ok += o.status == 0
    assert n_ok >= 18
    if not os.environ.get("GAMDP_DIAG_NO_DIRFREE"):
        env = dict(os.environ, GAMDP_DIAG_NO_DIRFREE="1", GAMDP_LIB=DIAG_LIB)
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), "-k",
                            "direction_free or golden_large"], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
