"""CPU checks of the product library: it loads, exports every symbol include/gamdp.h declares, its
host-only functions (encode / revcomp / findHits / synthetic generator) match the oracle and the golden
vectors, and it refuses to create a context without a GPU (no CPU fallback)."""
import ctypes
import os
import random
import re

import pytest

import _cases
import _golden as G
import _oracle as O
import gam_ngs_amd as gam
from gam_ngs_amd import api, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "gamdp.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(gamdp_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(lib.SYMBOLS)
    l = lib.load_library()
    for name in declared:
        assert hasattr(l, name), name


def test_struct_sizes_match_header_layout():
    assert ctypes.sizeof(lib.Task) == 64
    assert ctypes.sizeof(lib.Result) == 96
    assert ctypes.sizeof(lib.BlockC) == 32
    assert ctypes.sizeof(lib.MbIn) == 24
    assert ctypes.sizeof(lib.MbOut) == 32


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gam.GamdpError):
        gam.Context(0)


def test_encode_revcomp_match_golden():
    for d in G.load("seqops.json"):
        codes = api.encode(d["input"])
        assert codes == O.encode(d["input"])
        if d["op"] == "normalise":
            assert api.decode(codes) == d["output"]
        else:
            assert api.decode(api.reverse_complement(codes)) == d["output"]


def test_find_hits_matches_golden_and_oracle():
    ab = gam.ABlast(20)
    for d in G.load("findhits.json"):
        got = gam.ABlast(d["word"]).findHits(api.encode(d["a"]), d["a_s"], d["a_e"], api.encode(d["b"]), d["b_s"], d["b_e"])
        assert got == d["hits"], d["name"]
    rng = random.Random(11)
    for _ in range(300):
        la = rng.randint(0, 400)
        a = _cases.rand_seq(rng, la, 0.02 if rng.random() < 0.3 else 0)
        off = rng.randint(0, max(0, la // 2))
        b = _cases.mutate(rng, a[off:], 0.02, 0.005, 0.005) if rng.random() < 0.7 else _cases.rand_seq(rng, rng.randint(0, 300))
        ca, cb = api.encode(a), api.encode(b)
        args = (rng.randint(0, 20), rng.randint(0, la + 5), rng.randint(0, 20), rng.randint(0, len(b) + 5))
        assert ab.findHits(ca, args[0], args[1], cb, args[2], args[3]) == O.oracle_find_hits(ca, args[0], args[1], cb, args[2], args[3], 20)


def test_synthetic_generator_matches_oracle_generator():
    lib_o = O.oracle()
    for k, n in ((0, 1000), (5, 50000), (123456, 777)):
        m, s = api.synth_pair(k, n)
        mm = ctypes.create_string_buffer(n)
        ss = ctypes.create_string_buffer(n + n // 8 + 64)
        sl = lib_o.gamdp_oracle_synth_pair(k, n, mm, ss)
        assert (m, s) == (mm.raw[:n], ss.raw[:sl])


def test_fasta_loader_matches_reference_golden(tmp_path):
    """gamdp_fasta_open against what the reference's readNextContigID/readNextSequence load (tests/golden/fasta.json)."""
    for d in G.load("fasta.json"):
        path = tmp_path / (d["name"] + ".fa")
        with open(path, "w", newline="") as f:
            f.write(d["text"])
        names, codes = api.load_fasta(str(path))
        assert names == d["names"], d["name"]
        assert [api.decode(c) for c in codes] == d["seqs"], d["name"]
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n>x\nAC\n")
    with pytest.raises(gam.GamdpError):
        api.load_fasta(str(bad))
