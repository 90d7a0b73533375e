"""Pins oracle/gamdp_oracle.c against the committed golden vectors (generated from the reference
by tests/golden/make_golden.py).  Runs everywhere, including the GPU box."""
import ctypes
import zlib

import pytest

import _golden as G
import _oracle as O


def test_l0_small_cases():
    n = 0
    for name, c, e in G.l0_cases():
        r, ops = O.oracle_align(O.encode(c["a"]), O.encode(c["b"]), c["band"], c["begin_a"], c["end_a"],
                                c["begin_b"], c["end_b"], c["fs"], c["fe"])
        assert r.key() == G.expect_key(e), name
        G.check_ops(e, ops)
        n += 1
    assert n >= 650


@pytest.mark.parametrize("idx", range(6))
def test_l0_large_synthetic(idx):
    d = G.load("l0_large.json")[idx]
    lib = O.oracle()
    m = ctypes.create_string_buffer(d["len"])
    s = ctypes.create_string_buffer(d["len"] + d["len"] // 8 + 64)
    sl = lib.gamdp_oracle_synth_pair(d["k"], d["len"], m, s)
    assert sl == d["slave_len"]
    a, b = m.raw[:d["len"]], s.raw[:sl]
    # the generator itself is part of the contract: same bytes as when the vectors were made
    assert zlib.crc32(O.decode(a).encode()) == d["a_crc32"]
    assert zlib.crc32(O.decode(b).encode()) == d["b_crc32"]
    r, ops = O.oracle_align(a, b, d["band"], 0, d["len"] - 1, 0, sl - 1)
    assert r.key() == G.expect_key(d["expect"])
    G.check_ops(d["expect"], ops)
    assert r.cells == min(sl, d["len"] + d["band"]) * (2 * d["band"] + 1)


def test_l0_adversarial_long_pairs():
    """48 reference-generated long pairs built against the value-range argument of the packed-f16 blocks (homopolymers,
    dinucleotide repeats, unrelated / complementary / half-diverged sequences, indels that pin the path to a band edge,
    tandem repeats of the lane widths): the restatement first."""
    items = G.adversarial_cases()
    assert len(items) >= 48 and {d["band"] for d, _, _ in items} == {150, 512}
    for d, c, e in items:
        r, ops = O.oracle_align(O.encode(c["a"]), O.encode(c["b"]), c["band"], c["begin_a"], c["end_a"], c["begin_b"], c["end_b"])
        assert r.key() == G.expect_key(e), (d["kind"], d["n"], d["band"])
        G.check_ops(e, ops)


def test_find_hits():
    for d in G.load("findhits.json"):
        got = O.oracle_find_hits(O.encode(d["a"]), d["a_s"], d["a_e"], O.encode(d["b"]), d["b_s"], d["b_e"], d["word"])
        assert got == d["hits"], d["name"]


def test_seqops():
    for d in G.load("seqops.json"):
        codes = O.encode(d["input"])
        if d["op"] == "normalise":
            assert O.decode(codes) == d["output"]
        else:
            buf = ctypes.create_string_buffer(codes, max(1, len(codes)))
            O.oracle().gamdp_oracle_revcomp(buf, len(codes))
            assert O.decode(buf.raw[:len(codes)]) == d["output"]
