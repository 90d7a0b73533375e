"""Reader / writer of the merge-block dump format (integration/GamdpBridge.cc `dump`, tools/gamdp_align_mb.cpp input):
<prefix>.mergeblocks.tsv + <prefix>.mergeblocks.out.tsv, contigs referred to by FASTA name."""
import os


def read_dump(prefix):
    """-> list of dict(m_name, s_name, tails, blocks, graph, list, out=dict(...) or None)"""
    recs, graph, lst = [], -1, -1
    for line in open(prefix + ".mergeblocks.tsv"):
        line = line.rstrip("\n")
        if line.startswith("#graph"):
            graph += 1
            lst = -1
            continue
        if line.startswith("#list"):
            lst += 1
            continue
        if not line or line.startswith("#"):
            continue
        f = line.split("\t")
        nb = int(f[6])
        blocks = []
        for k in range(nb):
            b = f[7 + 7 * k: 14 + 7 * k]
            blocks.append((int(b[0]), int(b[1]), int(b[2]), int(b[3]), b[4], b[5], int(b[6])))
        recs.append(dict(m_name=f[0], s_name=f[1], tails=tuple(int(x) for x in f[2:6]), blocks=blocks, graph=max(graph, 0),
                         list=max(lst, 0), out=None))
    out_path = prefix + ".mergeblocks.out.tsv"
    if os.path.exists(out_path):
        rows = [l.rstrip("\n").split("\t") for l in open(out_path) if l.strip() and not l.startswith("#")]
        assert len(rows) == len(recs), "dump and its .out.tsv differ in length"
        for r, row in zip(recs, rows):
            assert (row[0], row[1]) == (r["m_name"], r["s_name"])
            v = [int(x) for x in row[2:]]
            r["out"] = dict(thrown=v[0], align_ok=v[1], align_rev=v[2], coords_set=v[3], m_start=v[4], m_end=v[5], s_start=v[6], s_end=v[7])
    return recs


def write_dump(prefix, recs):
    """the same two files from records that carry `out` (used by the harness self-test)"""
    with open(prefix + ".mergeblocks.tsv", "w") as fin, open(prefix + ".mergeblocks.out.tsv", "w") as fout:
        fin.write("#m_name\ts_name\tm_ltail\tm_rtail\ts_ltail\ts_rtail\tn_blocks\t(blocks)*\n")
        fout.write("#m_name\ts_name\tthrown\talign_ok\talign_rev\tcoords_set\tm_start\tm_end\ts_start\ts_end\n")
        g = l = None
        for r in recs:
            if r["graph"] != g:
                fin.write("#graph\n")
                g, l = r["graph"], None
            if r["list"] != l:
                fin.write("#list\n")
                l = r["list"]
            f = [r["m_name"], r["s_name"]] + [str(int(x)) for x in r["tails"]] + [str(len(r["blocks"]))]
            for b in r["blocks"]:
                f += [str(x) for x in b]
            fin.write("\t".join(f) + "\n")
            o = r["out"]
            fout.write("\t".join([r["m_name"], r["s_name"]] + [str(int(o[k])) for k in
                                 ("thrown", "align_ok", "align_rev", "coords_set", "m_start", "m_end", "s_start", "s_end")]) + "\n")


def compare(rec, got):
    """got: dict(thrown, align_ok, coords_set, align_rev, m_start, m_end, s_start, s_end) from the oracle / the GPU;
    rec['out']: what the reference wrote.  The reference's dump cannot say whether alignMergeBlock returned before
    touching the coordinates (PctgBuilder.cc:825-829), so they are compared only where OUR driver says they were
    written.  Returns None or a description of the difference."""
    want = rec["out"]
    if bool(want["thrown"]) != bool(got["thrown"]):
        return "thrown %r vs %r" % (want["thrown"], got["thrown"])
    if want["thrown"]:
        return None
    if bool(want["align_ok"]) != bool(got["align_ok"]):
        return "align_ok %r vs %r" % (want["align_ok"], got["align_ok"])
    if got["coords_set"]:
        for k in ("align_rev", "m_start", "m_end", "s_start", "s_end"):
            if int(want[k]) != int(got[k]):
                return "%s %r vs %r" % (k, want[k], got[k])
    return None
