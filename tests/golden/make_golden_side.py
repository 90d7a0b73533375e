#!/usr/bin/env python3
"""Generates tests/golden/side_outputs.json: contigs rendered by the REFERENCE's own operator<<(ostream&, const Contig&)
the way gam-merge writes its ".noblocks.BF.fasta" / ".noblocks.AF.fasta" / ".notmerged.fasta" side outputs
(`stream << *contig << std::endl`, src/Merge.cc:350, 370, 429), through oracle/_ref/libgamref.so
(oracle/ref_shim.cc gamref_render_contigs).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_side.py

What this pins: the text of a selected-contigs FASTA (names, 60 columns, empty contigs, N).  What it cannot pin: the
selection itself -- getNoBlocksContigs / getNoBlocksAfterFilterContigs (Block.cc:810-925) use boost::dynamic_bitset
and Frame.hpp's sparsehash, absent here; they are ten lines of set arithmetic and tests/test_side_outputs.py holds
their restatement."""
import ctypes as C
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402


def main():
    lib = O.ref()
    assert lib is not None, "oracle/_ref/libgamref.so is needed (make -C oracle)"
    lib.gamref_render_contigs.restype = C.c_int64
    lib.gamref_render_contigs.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_char_p, C.c_uint64]
    rng = random.Random(20261003)
    cases = []
    for k in range(12):
        n = rng.randint(1, 9)
        lens = [rng.choice([0, 1, 59, 60, 61, 120, 121, rng.randint(2, 400)]) for _ in range(n)]
        seqs = ["".join(rng.choice("ACGTN" if rng.random() < 0.3 else "ACGT") for _ in range(m)) for m in lens]
        names = [rng.choice(["ctg%d", "scaffold%d.1", "NODE_%d_length_77", "c%d"]) % i for i in range(n)]
        select = [int(rng.random() < 0.6) for _ in range(n)]
        if k == 0:
            select = [0] * n
        if k == 1:
            select = [1] * n
        buf = C.create_string_buffer(1 << 16)
        r = lib.gamref_render_contigs((C.c_char_p * n)(*[x.encode() for x in names]), (C.c_char_p * n)(*[x.encode() for x in seqs]),
                                      n, bytes(select), buf, 1 << 16)
        assert r >= 0
        cases.append(dict(names=names, seqs=seqs, select=select, text=buf.value.decode()))
    json.dump(cases, open(os.path.join(HERE, "side_outputs.json"), "w"), indent=0)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
