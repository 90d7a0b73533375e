"""Pins oracle/gamdp_oracle.c against the reference's own code (oracle/_ref/libgamref.so).

Skipped where the reference build is unavailable (e.g. the GPU box); there the oracle is pinned
by tests/test_oracle_golden.py against the committed vectors generated from the same reference.
"""
import random

import pytest

import _cases
import _oracle as O

pytestmark = pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built (no /root/reference)")


def check_case(c):
    a, b = O.encode(c["a"]), O.encode(c["b"])
    args = (c["band"], c["begin_a"], c["end_a"], c["begin_b"], c["end_b"], c["fs"], c["fe"])
    r, ops = O.oracle_align(a, b, *args)
    if r.status == O.INVALID:
        return "invalid"
    rr, rops = O.ref_align(c["a"], c["b"], *args)
    assert r.key() == O.ref_key(rr), (c, r.key(), O.ref_key(rr))
    assert ops == rops, c
    return r.status


@pytest.mark.parametrize("seed", range(8))
def test_random_cases_match_reference(seed):
    stats = {}
    for c in _cases.cases(1000 + seed, 700):
        s = check_case(c)
        stats[s] = stats.get(s, 0) + 1
    assert stats.get(O.OK, 0) > 300
    assert stats.get("invalid", 0) == 0


def test_out_of_range_cases_exist_and_match():
    n_oor = 0
    for c in _cases.cases(77, 3000, max_len=80, bands=(0, 1, 2, 5, 8)):
        if check_case(c) == O.OUT_OF_RANGE:
            n_oor += 1
    assert n_oor >= 5


def test_begin_a_at_or_past_the_end_of_a_matches_reference():
    """begin_a >= |a| (up to far past it): the reference's row bound wraps; outcomes are EMPTY / OUT_OF_RANGE only"""
    stats = {}
    for c in _cases.beyond_cases(31, 1500):
        s = check_case(c)
        stats[s] = stats.get(s, 0) + 1
    assert stats.get(O.OUT_OF_RANGE, 0) > 100 and stats.get(O.EMPTY, 0) > 100, stats


def test_medium_pairs_band150_and_512():
    rng = random.Random(5)
    for n, band in ((3000, 150), (2500, 512), (4000, 20)):
        a, b = _cases.related_pair(rng, n, n_frac=0.01)
        c = dict(a=a.encode(), b=b.encode(), band=band, begin_a=0, end_a=len(a) - 1, begin_b=0,
                 end_b=len(b) - 1, fs=False, fe=False)
        assert check_case(c) == O.OK


def test_iupac_and_lowercase_normalise_like_reference():
    s = b"acgtnACGTNRYKMxX-*"
    import ctypes
    buf = ctypes.create_string_buffer(s, len(s))
    O.ref().gamref_normalise(buf, len(s))
    assert O.decode(O.encode(s)).encode() == buf.raw[:len(s)]


def test_reverse_complement_matches_reference():
    import ctypes
    rng = random.Random(3)
    for n in (0, 1, 2, 3, 59, 60, 61, 120, 1001):
        s = _cases.rand_seq(rng, n, 0.05).encode()
        buf = ctypes.create_string_buffer(s, max(1, n))
        O.ref().gamref_reverse_complement(buf, n)
        codes = ctypes.create_string_buffer(O.encode(s), max(1, n))
        O.oracle().gamdp_oracle_revcomp(codes, n)
        assert O.decode(codes.raw[:n]).encode() == buf.raw[:n]


@pytest.mark.parametrize("seed", range(3))
def test_find_hits_matches_reference(seed):
    rng = random.Random(4000 + seed)
    n_nonempty = 0
    for _ in range(400):
        word = rng.choice([4, 8, 12, 20])
        la, lb = rng.randint(0, 200), rng.randint(0, 200)
        kind = rng.random()
        if kind < 0.5 and la >= 30:
            a = _cases.rand_seq(rng, la, 0.03 if rng.random() < 0.3 else 0)
            off = rng.randint(0, la // 2)
            b = _cases.mutate(rng, a[off:], 0.01, 0.0, 0.0)
        elif kind < 0.7:
            a = "".join(rng.choice("AC") for _ in range(la))
            b = "".join(rng.choice("AC") for _ in range(lb))
        else:
            a, b = _cases.rand_seq(rng, la), _cases.rand_seq(rng, lb)
        a, b = a.encode(), b.encode()
        a_s, a_e = rng.randint(0, max(0, len(a) // 3)), rng.randint(0, len(a) + 5)
        b_s, b_e = rng.randint(0, max(0, len(b) // 3)), rng.randint(0, len(b) + 5)
        got = O.oracle_find_hits(O.encode(a), a_s, a_e, O.encode(b), b_s, b_e, word)
        want = O.ref_find_hits(a, a_s, a_e, b, b_s, b_e, word)
        assert got == want, (a, b, a_s, a_e, b_s, b_e, word)
        n_nonempty += bool(want)
    assert n_nonempty > 50
