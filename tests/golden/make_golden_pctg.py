#!/usr/bin/env python3
"""Generates tests/golden/pctg_writers.json: paired contigs rendered by the REFERENCE's own PairedContig,
operator<<(ostream&, const Contig&) and writePctgDescriptors (compiled into oracle/_ref/libgamref.so, see
oracle/ref_shim.cc gamref_render_pctgs).  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_pctg.py

A case = two small assemblies, the merge lists of one or more graphs, the pieces (merge-list rows + whether the
bases come from the reverse complement of the contig) the restatement oracle/pctg_oracle.py derives from them, and the two texts the reference writes for paired contigs made of exactly
those pieces.  What this pins: sequence rendering (60 columns, names PairedContig_<id>, reverse complements through
the reference's reverse_complement) and every column of the .pctgs rows.  What it cannot pin: which pieces the
reference's PctgBuilder would have chosen (that class needs Boost.Graph)."""
import ctypes as C
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import _oracle as O  # noqa: E402
import _pctgcases as PC  # noqa: E402
import pctg_oracle as PO  # noqa: E402


def render(lib, master, slave, m_names, s_names, pcs, merged):
    def arr(strs):
        return (C.c_char_p * max(1, len(strs)))(*[s.encode() for s in strs])
    pieces = []
    for i, p in enumerate(pcs):
        for (cid, start, end, rev, is_master), src in zip(p.rows, p.src_rev):
            pieces += [i, int(is_master), cid, start, end, int(rev), int(src)]
    buf = (C.c_int64 * max(1, len(pieces)))(*pieces)
    fa = C.create_string_buffer(1 << 20)
    de = C.create_string_buffer(1 << 20)
    lib.gamref_render_pctgs.restype = C.c_int64
    lib.gamref_render_pctgs.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_char_p),
                                        C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_int64), C.c_uint64, C.c_uint32,
                                        C.c_uint64, C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]
    n = lib.gamref_render_pctgs(arr(m_names), arr(["".join(PO.LETTERS[c] for c in s) for s in master]), len(master),
                                arr(s_names), arr(["".join(PO.LETTERS[c] for c in s) for s in slave]), len(slave),
                                buf, len(pieces) // 7, len(pcs), merged, fa, 1 << 20, de, 1 << 20)
    assert n >= 0
    return fa.value.decode(), de.value.decode()


def case(lib, name, master, slave, graphs):
    m_names = ["mctg_%d" % i for i in range(len(master))]
    s_names = ["scaffold%d.1" % i for i in range(len(slave))]
    pcs, merged = PO.run(graphs, master, slave, PC.vote_mb)
    fasta, pctgs = render(lib, master, slave, m_names, s_names, pcs, merged)
    return dict(name=name, master=["".join(PO.LETTERS[c] for c in s) for s in master],
                slave=["".join(PO.LETTERS[c] for c in s) for s in slave], master_names=m_names, slave_names=s_names,
                graphs=graphs, merged=merged, pieces=[[list(r) + [src] for r, src in zip(p.rows, p.src_rev)] for p in pcs], fasta=fasta, pctgs=pctgs)


def main():
    lib = O.ref()
    assert lib is not None, "needs /root/reference"
    out = []
    # line-length edge cases: single-contig paired contigs of 1, 59, 60, 61, 119, 120, 121 bases, nothing merged
    rng = random.Random(7)
    master = [[rng.randrange(5) for _ in range(n)] for n in (1, 59, 60, 61, 119, 120, 121)]
    slave = [[rng.randrange(4) for _ in range(80)]]
    out.append(case(lib, "line_lengths_singles_only", master, slave, []))
    # one merged paired contig whose size is an exact multiple of 60, all master contigs used (no separator row)
    master = [[k % 4 for k in range(120)]]
    slave = [[(k * 3) % 4 for k in range(150)]]
    mb = dict(m_id=0, m_start=20, m_end=99, s_id=0, s_start=10, s_end=88, align_rev=0, align_ok=1, m_ltail=1, m_rtail=1,
              s_ltail=1, s_rtail=1, ext_slave_next=1, ext_slave_prev=1, m_rev=0, s_rev=0)
    out.append(case(lib, "one_merged_no_singles", master, slave, [[[mb]]]))
    # reversed slave taken by vote, with tails
    mb2 = dict(mb, m_end=60, s_start=5, s_end=140, align_rev=1)
    out.append(case(lib, "reversed_slave_region", master + [[3, 2, 1, 0] * 10], slave, [[[mb2]]]))
    for seed in (1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233):
        m, s, graphs = PC.scenario(seed)
        out.append(case(lib, "scenario_%d" % seed, m, s, graphs))
    with open(os.path.join(HERE, "pctg_writers.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("pctg_writers.json", len(out), "cases,", sum(len(c["pieces"]) for c in out), "paired contigs,",
          sum(1 for c in out for p in c["pieces"] for r in p if r[3]), "reversed rows")


if __name__ == "__main__":
    main()
