#!/usr/bin/env python3
"""Generates the committed golden vectors from the REFERENCE's own code.

Run in the build container (needs /root/reference):   python tests/golden/make_golden.py
It builds oracle/_ref/libgamref.so (reference sources compiled where they lie, see
oracle/Makefile) and records, for every case, exactly what the reference returns:
MyAlignment fields, first/last_match_pos, the edit string (or its CRC32 when long), and
ABlast::findHits lists.  Only data (inputs + expected outputs) is written; no reference
source text.  Files written next to this script:

  l0_handbuilt.json   SURVEY Appendix D checklist items 1-9, 11, 14 (named cases)
  l0_random.json      600 seeded random windowed cases (tests/_cases.py, seed 2024)
  l0_large.json       synthetic 50 kb pairs (generator = gamdp_oracle_synth_pair, keys stored)
  l0_adversarial.json 48 long adversarial pairs (tests/_cases.py adversarial_pair: homopolymers, dinucleotide repeats,
                      unrelated / complementary / 50 %-diverged sequences, long indels that pin the path to a band edge,
                      tandem repeats of the lane widths) at bands 512 and 150: recipe + CRC32 of the inputs, the
                      reference's summary and the CRC32 of its edit string
  findhits.json       ABlast::findHits cases (checklist item 10)
  seqops.json         normalisation + reverse_complement cases (items 7, 13)
  fasta.json          FASTA files and what the reference's readNextContigID/readNextSequence load from them
"""
import ctypes
import json
import os
import random
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _cases  # noqa: E402
import _oracle as O  # noqa: E402


def expect(c):
    r, ops = O.ref_align(c["a"], c["b"], c["band"], c["begin_a"], c["end_a"], c["begin_b"], c["end_b"],
                         c["fs"], c["fe"])
    k = O.ref_key(r)
    e = dict(status=k[0], begin_a=k[1], begin_b=k[2], score=k[3], n_match=k[4], length=k[5],
             first_a=k[6], first_b=k[7], first_found=k[8], last_a=k[9], last_b=k[10], last_found=k[11],
             homology=k[12])
    if len(ops) <= 4096:
        e["ops"] = ops
    else:
        e["ops_crc32"] = zlib.crc32(ops.encode())
    return e


def jcase(name, c):
    d = dict(c)
    d["a"] = c["a"].decode()
    d["b"] = c["b"].decode()
    d["name"] = name
    d["expect"] = expect(c)
    return d


def mk(a, b, band, begin_a=None, end_a=None, begin_b=None, end_b=None, fs=False, fe=False):
    return dict(a=a.encode(), b=b.encode(), band=band,
                begin_a=0 if begin_a is None else begin_a, end_a=len(a) - 1 if end_a is None else end_a,
                begin_b=0 if begin_b is None else begin_b, end_b=len(b) - 1 if end_b is None else end_b,
                fs=fs, fe=fe)


def handbuilt():
    rng = random.Random(42)
    out = []
    base = _cases.rand_seq(rng, 120)

    def add(name, *a, **k):
        out.append(jcase(name, mk(*a, **k)))

    # 1. identical / single edits at start, middle, end
    add("identical", base, base, 20)
    for where, p in (("start", 0), ("middle", 60), ("end", 119)):
        sub = base[:p] + ("A" if base[p] != "A" else "C") + base[p + 1:]
        add("subst_" + where, base, sub, 20)
        add("insert_" + where, base, base[:p] + "G" + base[p:], 20)
        add("delete_" + where, base, base[:p] + base[p + 1:], 20)
    # 2. bands; indel run longer than the band
    for band in (0, 1, 5, 150, 512):
        add("band_%d_related" % band, base, _cases.mutate(rng, base), band)
    long_ins = base[:50] + _cases.rand_seq(rng, 12) + base[50:]
    add("indel_run_gt_band_in_b", base, long_ins, 5)
    add("indel_run_gt_band_in_a", long_ins, base, 5)
    # 3. begin_a vs band; end_a variants
    a300 = _cases.rand_seq(rng, 300)
    for w in (8,):
        for ba in (0, 3, 8, 20, 290):
            b = _cases.mutate(rng, a300[ba:ba + 100])
            add("begin_a_%d_w%d" % (ba, w), a300, b, w, begin_a=ba, end_a=min(299, ba + 99))
        add("end_a_lt_begin_plus_w", a300, a300[10:60], w, begin_a=10, end_a=14)
        add("end_a_past_a", a300, a300[250:] + "ACGTACGT", w, begin_a=250, end_a=320)
        add("end_a_past_a_unrelated", a300, _cases.rand_seq(rng, 80), w, begin_a=260, end_a=400)
    # 4. b window edge cases
    add("end_b_lt_begin_b", base, base, 5, begin_b=10, end_b=9)
    add("end_b_past_b", base, base, 5, end_b=500)
    add("x_limited_by_a", base[:40], base + base, 4)
    # 5. force flags
    for fs in (False, True):
        for fe in (False, True):
            add("force_%d%d" % (fs, fe), a300, _cases.mutate(rng, a300[5:200]), 20, begin_a=0, end_a=199, fs=fs, fe=fe)
            add("force_%d%d_begin15" % (fs, fe), a300, _cases.mutate(rng, a300[15:200]), 8, begin_a=15, end_a=199, fs=fs, fe=fe)
            add("force_%d%d_shift_b" % (fs, fe), a300, _cases.rand_seq(rng, 14) + a300[:150], 20, begin_a=0, end_a=170, fs=fs, fe=fe)
    add("force_end_short_x", base, base[:8], 5, end_a=7, fe=True)
    add("force_end_x_11", base, base[:11], 5, end_a=10, fe=True)
    add("force_end_x_12", base, base[:12], 5, end_a=11, fe=True)
    add("force_end_far_antidiag", base, base[:12], 5, fe=True)
    # 6. N content
    n1 = base[:30] + "NNNN" + base[34:]
    add("n_in_a", n1, base, 10)
    add("n_in_b", base, n1, 10)
    add("n_both_same_place", n1, n1, 10)
    add("all_n", "N" * 40, "N" * 40, 5)
    add("all_n_vs_bases", "N" * 40, base[:40], 5)
    # 7. lower case + IUPAC
    add("lower_iupac", base.lower()[:60] + "RYKM" + base[64:], base[:60] + "nnnn" + base.lower()[64:], 10)
    # 8. ties
    add("ties_lowcomplex", "ACACACACACACACACACAC", "ACACACACACACACAC", 6)
    add("ties_homopolymer", "A" * 30, "A" * 24, 8)
    add("ties_homopolymer_window", "A" * 30, "A" * 24, 8, begin_a=3, end_a=20)
    # 9. negative scores (no zero floor)
    add("unrelated", _cases.rand_seq(rng, 100), _cases.rand_seq(rng, 100), 10)
    add("unrelated_band0", _cases.rand_seq(rng, 60), _cases.rand_seq(rng, 60), 0)
    # 11. no MATCH at all
    add("no_match_ops", "AAAAAAAAAA", "CCCCCCCCCC", 0)
    add("no_match_ops_band2", "AAAAAAAAAA", "CCCCCCCCCC", 2)
    # 14. throwing and near-miss inputs with end_a >= |a|
    n_thr = n_ok = 0
    r2 = random.Random(314)
    while n_thr < 8 or n_ok < 8:
        c = _cases.random_case(r2, max_len=60, bands=(0, 1, 2, 5, 8))
        if c["end_a"] < len(c["a"]):
            continue
        e = expect(c)
        if e["status"] == O.OUT_OF_RANGE and n_thr < 8:
            out.append(jcase("throws_%d" % n_thr, c))
            n_thr += 1
        elif e["status"] == O.OK and n_ok < 8:
            out.append(jcase("end_a_past_ok_%d" % n_ok, c))
            n_ok += 1
    return out


def large():
    lib = O.oracle()
    out = []
    for k, length, band in ((0, 50000, 150), (0, 50000, 512), (1, 50000, 512), (2, 50000, 150), (3, 20000, 512),
                            (4, 5000, 150)):
        m = ctypes.create_string_buffer(length)
        s = ctypes.create_string_buffer(length + length // 8 + 64)
        sl = lib.gamdp_oracle_synth_pair(k, length, m, s)
        a, b = O.decode(m.raw[:length]).encode(), O.decode(s.raw[:sl]).encode()
        c = dict(a=a, b=b, band=band, begin_a=0, end_a=length - 1, begin_b=0, end_b=sl - 1, fs=False, fe=False)
        out.append(dict(k=k, len=length, band=band, slave_len=sl, a_crc32=zlib.crc32(a), b_crc32=zlib.crc32(b),
                        expect=expect(c)))
    return out


def adversarial():
    out = []
    for kind, n, band in _cases.adversarial_specs():
        c = _cases.adversarial_case(kind, n, band)
        out.append(dict(kind=kind, n=n, band=band, a_len=len(c["a"]), b_len=len(c["b"]), a_crc32=zlib.crc32(c["a"]),
                        b_crc32=zlib.crc32(c["b"]), expect=expect(c)))
    return out


def findhits():
    rng = random.Random(99)
    out = []

    def add(name, a, b, a_s, a_e, b_s, b_e, word=20):
        hits = O.ref_find_hits(a.encode(), a_s, a_e, b.encode(), b_s, b_e, word)
        out.append(dict(name=name, a=a, b=b, a_s=a_s, a_e=a_e, b_s=b_s, b_e=b_e, word=word, hits=hits))

    a = _cases.rand_seq(rng, 200)
    add("no_hit", a, _cases.rand_seq(rng, 150), 0, 199, 0, 149)
    add("unique_hit", a, a[40:140], 0, 199, 0, 99)
    add("unique_hit_window", a, a[40:140], 10, 180, 0, 99)
    rep = _cases.rand_seq(rng, 30)
    add("multi_equal_votes", rep * 4, rep, 0, 119, 0, 29)
    add("shorter_than_word", a[:15], a[:15], 0, 14, 0, 14)
    add("bounds_clamp", a, a[100:], 0, 5000, 0, 5000)
    add("start_after_end", a, a, 50, 10, 0, 199)
    an = a[:60] + "NNN" + a[63:]
    add("with_n", an, an[30:120], 0, 199, 0, 89)
    add("word8", a, a[70:120], 0, 199, 0, 49, 8)
    for i in range(30):
        la = rng.randint(40, 160)
        x = _cases.rand_seq(rng, la, 0.02 if i % 3 == 0 else 0)
        off = rng.randint(0, la // 2)
        y = _cases.mutate(rng, x[off:], 0.02, 0.005, 0.005)
        add("rand_%d" % i, x, y, rng.randint(0, 10), rng.randint(la // 2, la + 3), 0, rng.randint(len(y) // 2, len(y) + 3),
            rng.choice([8, 12, 20]))
    return out


def seqops():
    rng = random.Random(7)
    out = []
    for s in ["acgtnACGTNRYKMxX-*", "A", "", "ACGTN" * 13]:
        buf = ctypes.create_string_buffer(s.encode(), max(1, len(s)))
        O.ref().gamref_normalise(buf, len(s))
        out.append(dict(op="normalise", input=s, output=buf.raw[:len(s)].decode()))
    for n in (1, 2, 59, 60, 61, 120):
        s = _cases.rand_seq(rng, n, 0.05)
        buf = ctypes.create_string_buffer(s.encode(), n)
        O.ref().gamref_reverse_complement(buf, n)
        out.append(dict(op="revcomp", input=s, output=buf.raw[:n].decode()))
    return out


FASTA_CASES = {
    "plain": ">c1\nACGTACGT\n>c2\nTTTTGGGG\nCCCC\n",
    "desc_and_case": ">ctg_1 length=12 cov=3.5\nacgtnACGTN\nryKM\n>ctg_2\tx\nAC\n",
    "blank_lines_and_spaces": "\n\n>a\nAC GT\n\nAC\n \n>b\n\n>c\nT\n",
    "no_trailing_newline": ">x\nACGT\n>y\nGG",
    "crlf": ">w1\r\nACGT\r\nAC\r\n",
    "single_long": ">only\n" + "ACGTTGCA" * 40 + "\n",
    "empty_record_last": ">p\nAAAA\n>q\n",
}


def fasta():
    import tempfile
    lib = O.ref()
    lib.gamref_load_fasta.restype = ctypes.c_int64
    lib.gamref_load_fasta.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                                      ctypes.c_uint64, ctypes.c_char_p, ctypes.c_uint64]
    out = []
    with tempfile.TemporaryDirectory() as d:
        for name, text in FASTA_CASES.items():
            path = os.path.join(d, name + ".fa")
            with open(path, "w", newline="") as f:
                f.write(text)
            names = ctypes.create_string_buffer(4096)
            seqs = ctypes.create_string_buffer(1 << 16)
            lens = (ctypes.c_uint64 * 64)()
            n = lib.gamref_load_fasta(path.encode(), names, 4096, lens, 64, seqs, 1 << 16)
            assert n >= 0, name
            nm = names.value.decode().split("\n")[:n]
            sq, pos = [], 0
            for i in range(n):
                sq.append(seqs.raw[pos:pos + lens[i]].decode())
                pos += lens[i]
            out.append(dict(name=name, text=text, names=nm, seqs=sq))
    return out


def main():
    assert O.ref() is not None, "needs /root/reference"
    only = sys.argv[1:]   # e.g. `make_golden.py l0_adversarial.json` regenerates that file alone

    def dump(name, make):
        if only and name not in only:
            return
        obj = make()
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(obj, f, indent=0, sort_keys=True)
        print(name, len(obj))

    dump("l0_handbuilt.json", handbuilt)
    dump("l0_random.json", lambda: [jcase("r%d" % i, c) for i, c in enumerate(_cases.cases(2024, 600))])
    dump("l0_large.json", large)
    dump("l0_adversarial.json", adversarial)
    dump("findhits.json", findhits)
    dump("seqops.json", seqops)
    dump("fasta.json", fasta)


if __name__ == "__main__":
    main()
