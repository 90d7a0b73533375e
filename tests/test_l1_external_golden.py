"""Pins the merge-block (L1) driver against dumps of the REAL reference, when such dumps exist under
tests/golden/l1_reference_dump/<name>/ (see the README there; produced by gam-merge + integration/gam-merge-gamdp.patch
with GAMDP_DUMP_PREFIX on a Boost host).  Without data the pin tests skip and L1 parity stays "unpinned"; the harness
itself is exercised on a dump written from the oracle's results."""
import glob
import os

import pytest

import _gage as G
import _l1dump as D
from _l1oracle import oracle_mb
from gam_ngs_amd import api

HERE = os.path.dirname(os.path.abspath(__file__))
DATASETS = sorted(d for d in glob.glob(os.path.join(HERE, "golden", "l1_reference_dump", "*")) if os.path.isdir(d))


def load_dataset(d):
    mn, mc = api.load_fasta(os.path.join(d, "master.fasta"))
    sn, sc = api.load_fasta(os.path.join(d, "slave.fasta"))
    recs = D.read_dump(os.path.join(d, "dump"))
    mid, sid = {n: i for i, n in enumerate(mn)}, {n: i for i, n in enumerate(sn)}
    return mc, sc, mid, sid, recs


def oracle_result(mc, sc, mid, sid, r):
    sc_ = dict(master=api.decode(mc[mid[r["m_name"]]]), slave=api.decode(sc[sid[r["s_name"]]]), blocks=r["blocks"], tails=r["tails"])
    o, _ = oracle_mb(sc_, audit_cap=1)
    return dict(thrown=o.status != 0, align_ok=o.align_ok, coords_set=o.touched, align_rev=o.align_rev, m_start=o.m_start,
                m_end=o.m_end, s_start=o.s_start, s_end=o.s_end)


def check_oracle(d):
    mc, sc, mid, sid, recs = load_dataset(d)
    bad = []
    for i, r in enumerate(recs):
        diff = D.compare(r, oracle_result(mc, sc, mid, sid, r))
        if diff:
            bad.append((i, r["m_name"], r["s_name"], diff))
    return len(recs), bad


def check_gpu(d):
    import gam_ngs_amd as gam
    from _gpu import ctx
    mc, sc, mid, sid, recs = load_dataset(d)
    c = ctx()
    ms, ss = gam.SequenceSet(c, mc, ascii=False), gam.SequenceSet(c, sc, ascii=False)
    mbs = [gam.MergeBlock(mid[r["m_name"]], sid[r["s_name"]], [gam.Block(*b) for b in r["blocks"]], *[bool(x) for x in r["tails"]]) for r in recs]
    gam.PctgBuilder(c, ms, ss).alignMergeBlocks(mbs)
    bad = []
    for i, (r, mb) in enumerate(zip(recs, mbs)):
        got = dict(thrown=mb.status != 0, align_ok=mb.align_ok, coords_set=mb.coords_set, align_rev=mb.align_rev, m_start=mb.m_start,
                   m_end=mb.m_end, s_start=mb.s_start, s_end=mb.s_end)
        diff = D.compare(r, got)
        if diff:
            bad.append((i, r["m_name"], r["s_name"], diff))
    ms.close(); ss.close()
    return len(recs), bad


@pytest.mark.skipif(not DATASETS, reason="no reference dump under tests/golden/l1_reference_dump/ (L1 parity unpinned)")
@pytest.mark.parametrize("d", DATASETS or ["-"])
def test_oracle_against_reference_dump(d):
    n, bad = check_oracle(d)
    assert n > 0 and not bad, bad[:5]


@pytest.mark.gpu
@pytest.mark.skipif(not DATASETS, reason="no reference dump under tests/golden/l1_reference_dump/ (L1 parity unpinned)")
@pytest.mark.parametrize("d", DATASETS or ["-"])
def test_gpu_against_reference_dump(d):
    n, bad = check_gpu(d)
    assert n > 0 and not bad, bad[:5]


def synthetic_dataset(tmp_path):
    """a dataset in the dump format whose `out` comes from the oracle (NOT a pin -- it only exercises the harness)"""
    pb = G.problem(5, genome_len=250_000)
    d = tmp_path / "synthetic"
    d.mkdir()
    for name, ctgs, tag in (("master.fasta", pb["master"], "m"), ("slave.fasta", pb["slave"], "s")):
        with open(d / name, "w") as f:
            for i, c in enumerate(ctgs):
                f.write(">%s%d\n%s\n" % (tag, i, G.to_ascii(c["seq"]).decode()))
    recs = []
    for gi, g in enumerate(pb["graphs"]):
        for li, l in enumerate(g):
            for mb in l:
                sc = dict(master=G.to_ascii(pb["master"][mb["m_id"]]["seq"]).decode(), slave=G.to_ascii(pb["slave"][mb["s_id"]]["seq"]).decode(),
                          blocks=mb["blocks"], tails=mb["tails"])
                o, _ = oracle_mb(sc, audit_cap=1)
                recs.append(dict(m_name="m%d" % mb["m_id"], s_name="s%d" % mb["s_id"], tails=mb["tails"], blocks=mb["blocks"], graph=gi, list=li,
                                 out=dict(thrown=o.status != 0, align_ok=o.align_ok, align_rev=o.align_rev, coords_set=1, m_start=o.m_start,
                                          m_end=o.m_end, s_start=o.s_start, s_end=o.s_end)))
    D.write_dump(str(d / "dump"), recs)
    return str(d), recs


def test_harness_on_a_dump_written_from_the_oracle(tmp_path):
    d, recs = synthetic_dataset(tmp_path)
    assert D.read_dump(os.path.join(d, "dump"))[0]["blocks"] == [tuple(b) for b in recs[0]["blocks"]]
    n, bad = check_oracle(d)
    assert n == len(recs) >= 10 and not bad
    # The harness would catch a misreading of PctgBuilder.cc: a dump that deviates from ours in ONE field of ONE merge
    # block fails, whichever field it is -- an off-by-one end coordinate, the orientation, the verdict, a throw.
    import copy
    k = next(i for i, r in enumerate(recs) if r["out"]["align_ok"])
    k_rev = next((i for i, r in enumerate(recs) if r["out"]["align_ok"] and r["out"]["align_rev"]), k)
    for field, idx, change in (("m_end", k, lambda v: v + 1), ("m_end", k, lambda v: v - 1), ("s_start", k, lambda v: v + 1),
                               ("align_rev", k, lambda v: 1 - int(v)), ("align_rev", k_rev, lambda v: 1 - int(v)),
                               ("align_ok", k, lambda v: 1 - int(v)), ("thrown", k, lambda v: 1)):
        bent = copy.deepcopy(recs)
        bent[idx]["out"][field] = change(bent[idx]["out"][field])
        D.write_dump(os.path.join(d, "dump"), bent)
        n, bad = check_oracle(d)
        assert [b[0] for b in bad] == [idx], (field, idx, bad[:3])
    D.write_dump(os.path.join(d, "dump"), recs)
    assert check_oracle(d)[1] == []


@pytest.mark.gpu
def test_gpu_harness_on_a_dump_written_from_the_oracle(tmp_path):
    d, recs = synthetic_dataset(tmp_path)
    n, bad = check_gpu(d)
    assert n == len(recs) and not bad, bad[:3]
    # ... and the GPU half of the harness catches a flipped orientation and an off-by-one end just the same
    k = next(i for i, r in enumerate(recs) if r["out"]["align_ok"])
    for field, change in (("align_rev", lambda v: 1 - int(v)), ("m_end", lambda v: v + 1)):
        bent = [dict(r, out=dict(r["out"])) for r in recs]
        bent[k]["out"][field] = change(bent[k]["out"][field])
        D.write_dump(os.path.join(d, "dump"), bent)
        n, bad = check_gpu(d)
        assert [b[0] for b in bad] == [k], (field, bad[:3])
