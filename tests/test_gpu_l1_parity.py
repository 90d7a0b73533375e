"""GPU parity of gamdp_align_merge_blocks (round-batched merge-block driver + HIP kernels) against the
oracle's restatement of PctgBuilder::alignMergeBlock: align_ok / align_rev / m_start..s_end bit for
bit, same number of DP calls, and every DP call's result identical (audit trail)."""
import ctypes as C

import pytest

import _l1cases
import _oracle as O
from _gpu import ctx
import gam_ngs_amd as gam

pytestmark = pytest.mark.gpu


from _l1oracle import oracle_mb  # noqa: E402,F401


def make_mbs(scs):
    return [gam.MergeBlock(i, i, [gam.Block(*b) for b in sc["blocks"]], *sc["tails"]) for i, sc in enumerate(scs)]


def _check_against_oracle(c, scs):
    masters = gam.SequenceSet(c, [sc["master"].encode() for sc in scs])
    slaves = gam.SequenceSet(c, [sc["slave"].encode() for sc in scs])
    mbs = make_mbs(scs)
    gam.PctgBuilder(c, masters, slaves).alignMergeBlocks(mbs, audit=16)
    stats = {"ok": 0, "rev": 0, "bad": 0, "tails": 0}
    for sc, mb in zip(scs, mbs):
        o, oaud = oracle_mb(sc)
        got = (mb.status, mb.align_ok, mb.coords_set, mb.n_dp, mb.cells)
        want = (o.status, bool(o.align_ok), bool(o.touched), o.n_dp, o.cells)
        assert got == want, (sc["kind"], got, want)
        if o.touched:
            assert (mb.align_rev, mb.m_start, mb.m_end, mb.s_start, mb.s_end) == \
                   (bool(o.align_rev), o.m_start, o.m_end, o.s_start, o.s_end), sc["kind"]
        assert [a.key() for a in mb.audit] == oaud, sc["kind"]
        stats["ok"] += mb.align_ok
        stats["rev"] += mb.align_ok and mb.align_rev
        stats["bad"] += not mb.align_ok
        stats["tails"] += mb.align_ok and mb.n_dp > len(sc["blocks"])
    return stats


@pytest.mark.parametrize("seed", range(4))
def test_merge_blocks_match_oracle(seed):
    stats = _check_against_oracle(ctx(), _l1cases.scenarios(500 + seed, 60))
    # the generator must exercise every branch of the driver
    assert stats["ok"] >= 15 and stats["rev"] >= 3 and stats["bad"] >= 5 and stats["tails"] >= 5, stats


@pytest.mark.parametrize("arena_kb", [24576, 6144, 2048, 700])
def test_merge_blocks_with_a_small_scratch_arena(arena_kb):
    """gamdp_ctx_set_arena_bytes bounds what a call may claim for scratch.  Half of it goes to the chain launch (per merge block
    three scratch slots sized for its own longest frame, up to 133 KB each here, 8 MB in all): with 24 MB the launch takes the
    60 merge blocks at once, the longest with twin workgroups; with 6 MB and 2 MB in three and eight pieces without twins; and
    with 700 KB the longest chain's workgroup does not fit at all and the call falls back to the round loop, a few calls per
    launch -- whatever the path, the results are the oracle's."""
    c = gam.Context(0)
    c.set_arena_bytes(arena_kb << 10)
    try:
        stats = _check_against_oracle(c, _l1cases.scenarios(503, 60))
    finally:
        c.set_arena_bytes(0)
    assert stats["ok"] >= 15 and stats["bad"] >= 5, stats


def test_merge_blocks_with_an_empty_slave_frame_at_the_start_of_the_slave():
    """A frame with s_end < s_begin has length 0 (Frame.cc:124-127).  When the call of such a block starts at slave base 0 the
    reference's end_b = begin_b + 0 - 1 wraps around (unsigned long, PctgBuilder.cc:1669-1677), find_alignment clips it to
    |b| - 1, and the call aligns the WHOLE slave -- far more rows than any frame of the merge block has (the chain
    kernel's scratch slots are sized by the longest frame).  Such merge blocks stay with the round loop; here they sit in one
    batch with ordinary ones, as first block, as second block (start clamped to 0) and alone: every decision and every DP
    record is the oracle's."""
    import random
    import _cases
    rng = random.Random(31)
    scs = _l1cases.scenarios(611, 24)
    odd = []
    for variant in range(6):
        core = _cases.rand_seq(rng, 2600)
        core_s = _cases.mutate(rng, core, 0.01, 0.003, 0.003)
        master = _cases.rand_seq(rng, rng.randint(0, 40)) + core + _cases.rand_seq(rng, 200)
        slave = core_s + _cases.rand_seq(rng, 150)
        off = len(master) - 200 - len(core)
        normal = (off + 300, off + 620, 290, 615, "+", "+", 20)
        normal2 = (off + 900, off + 1300, 890, 1290, "+", "+", 12)
        empty = (off + 10, off + 200, 0, -1, "+", "+", 7)          # s_begin = 0, s_end = -1: an empty slave frame
        empty_late = (off + 700, off + 800, 0, -1, "+", "+", 7)
        blocks = [[empty, normal, normal2], [normal, empty_late, normal2], [empty], [empty, normal], [normal, normal2, empty_late],
                  [(off + 10, off + 200, 5, 2, "+", "-", 9), normal]][variant]
        odd.append(dict(kind="empty_frame", master=master, slave=slave, blocks=blocks, tails=(True, True, True, True)))
    mixed = scs[:12] + odd + scs[12:]
    stats = _check_against_oracle(ctx(), mixed)
    assert stats["ok"] >= 5
    # and on their own (no ordinary merge block in the call: no chain launch at all)
    _check_against_oracle(ctx(), odd)


def test_single_merge_block_calls_match_batched():
    c = ctx()
    scs = _l1cases.scenarios(77, 6)
    masters = gam.SequenceSet(c, [sc["master"].encode() for sc in scs])
    slaves = gam.SequenceSet(c, [sc["slave"].encode() for sc in scs])
    pb = gam.PctgBuilder(c, masters, slaves)
    batched = pb.alignMergeBlocks(make_mbs(scs))
    for i, sc in enumerate(scs):
        one = pb.alignMergeBlock(make_mbs(scs)[i])
        b = batched[i]
        assert (one.align_ok, one.align_rev, one.m_start, one.m_end, one.s_start, one.s_end, one.n_dp) == \
               (b.align_ok, b.align_rev, b.m_start, b.m_end, b.s_start, b.s_end, b.n_dp)


def test_reference_exception_is_reported_per_merge_block():
    """A frame running past the master's end on unrelated sequences makes find_alignment pick a zero cell
    outside `a`: the reference throws and gam-merge drops the graph.  The driver must
    report OUT_OF_RANGE for that merge block only."""
    import random
    import _cases
    rng = random.Random(9)
    c = ctx()
    m0, s0 = _cases.rand_seq(rng, 300), _cases.rand_seq(rng, 300)
    good = _l1cases.scenario(random.Random(4), "overlap")
    masters = gam.SequenceSet(c, [m0.encode(), good["master"].encode()])
    slaves = gam.SequenceSet(c, [s0.encode(), good["slave"].encode()])
    scs = [dict(master=m0, slave=s0, blocks=[(250, 400, 10, 160, "+", "+", 9)], tails=(True, True, True, True)), good]
    mbs = make_mbs(scs)
    gam.PctgBuilder(c, masters, slaves, band=5).alignMergeBlocks(mbs, audit=8)
    o0, _ = oracle_mb(scs[0], band=5)
    assert o0.status == O.OUT_OF_RANGE
    assert mbs[0].status == O.OUT_OF_RANGE and not mbs[0].align_ok and not mbs[0].coords_set
    o1, aud1 = oracle_mb(scs[1], band=5)
    assert (mbs[1].status, mbs[1].align_ok, mbs[1].n_dp) == (o1.status, bool(o1.align_ok), o1.n_dp)
    assert [a.key() for a in mbs[1].audit] == aud1


def test_standalone_tool_from_files(tmp_path):
    """gamdp-align-mb (C++ over the C ABI): FASTA files + dumped merge blocks in, alignMergeBlock results out."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tool = os.path.join(root, "gam_ngs_amd", "gamdp-align-mb")
    scs = _l1cases.scenarios(901, 40)
    with open(tmp_path / "master.fa", "w") as f:
        for i, sc in enumerate(scs):
            f.write(">m%d some description\n" % i)
            for k in range(0, len(sc["master"]), 60):
                f.write(sc["master"][k:k + 60] + "\n")
    with open(tmp_path / "slave.fa", "w") as f:
        for i, sc in enumerate(scs):
            f.write(">s%d\n%s\n" % (i, sc["slave"]))
    with open(tmp_path / "mb.tsv", "w") as f:
        f.write("# dumped merge blocks\n")
        for i, sc in enumerate(scs):
            fields = ["m%d" % i, "s%d" % i] + [str(int(x)) for x in sc["tails"]] + [str(len(sc["blocks"]))]
            for b in sc["blocks"]:
                fields += [str(x) for x in b]
            f.write("\t".join(fields) + "\n")
    subprocess.run([tool, str(tmp_path / "master.fa"), str(tmp_path / "slave.fa"), str(tmp_path / "mb.tsv"), str(tmp_path / "out.tsv")],
                   check=True, timeout=300)
    rows = [l.rstrip("\n").split("\t") for l in open(tmp_path / "out.tsv") if not l.startswith("#")]
    assert len(rows) == len(scs)
    for sc, r in zip(scs, rows):
        o, _ = oracle_mb(sc)
        status, ok, rev, cset, ms, me, ss, se, ndp, cells = [int(x) for x in r[2:]]
        assert (status, ok, cset, ndp, cells) == (o.status, o.align_ok, o.touched, o.n_dp, o.cells)
        if o.touched:
            assert (rev, ms, me, ss, se) == (o.align_rev, o.m_start, o.m_end, o.s_start, o.s_end)


def test_standalone_tool_through_to_gam_fasta(tmp_path):
    """gamdp-align-mb --pctgs: merge blocks grouped into graphs and merge lists -> alignment on the GPU -> list surgery,
    buildPctgs, ids, single-contig pctgs -> .gam.fasta / .pctgs, against the two CPU restatements chained together."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    import pctg_oracle as PO
    tool = os.path.join(root, "gam_ngs_amd", "gamdp-align-mb")
    scs = _l1cases.scenarios(77, 60)
    code = {"A": 0, "T": 1, "C": 2, "G": 3}
    master = [[code.get(ch, 4) for ch in sc["master"].upper()] for sc in scs]
    slave = [[code.get(ch, 4) for ch in sc["slave"].upper()] for sc in scs]
    with open(tmp_path / "master.fa", "w") as f:
        for i, sc in enumerate(scs):
            f.write(">m%d\n%s\n" % (i, sc["master"]))
    with open(tmp_path / "slave.fa", "w") as f:
        for i, sc in enumerate(scs):
            f.write(">s%d\n%s\n" % (i, sc["slave"]))
    graphs, dropped = [], 0
    with open(tmp_path / "mb.tsv", "w") as f:
        for g0 in range(0, len(scs), 6):          # a graph = 6 merge blocks in lists of 1..3
            f.write("#graph\n")
            lists, thrown = [], False
            for i in range(g0, min(g0 + 6, len(scs))):
                if (i - g0) in (0, 1, 3):
                    f.write("#list\n")
                    lists.append([])
                sc = scs[i]
                fields = ["m%d" % i, "s%d" % i] + [str(int(x)) for x in sc["tails"]] + [str(len(sc["blocks"]))]
                for b in sc["blocks"]:
                    fields += [str(x) for x in b]
                f.write("\t".join(fields) + "\n")
                o, _ = oracle_mb(sc)
                thrown |= o.status != 0
                t = [int(x) for x in sc["tails"]]
                lists[-1].append(dict(m_id=i, s_id=i, m_start=o.m_start, m_end=o.m_end, s_start=o.s_start, s_end=o.s_end,
                                      align_ok=int(o.align_ok), align_rev=int(o.align_rev) if o.touched else 0,
                                      m_ltail=t[0], m_rtail=t[1], s_ltail=t[2], s_rtail=t[3], ext_slave_next=1,
                                      ext_slave_prev=1, m_rev=0, s_rev=0))
            if thrown:
                dropped += 1
            else:
                graphs.append(lists)
    subprocess.run([tool, str(tmp_path / "master.fa"), str(tmp_path / "slave.fa"), str(tmp_path / "mb.tsv"), str(tmp_path / "out.tsv"),
                    "--pctgs", str(tmp_path / "run"), "--vote", "master"], check=True, timeout=300)
    want, merged = PO.run(graphs, master, slave, lambda mb: 0)
    assert merged > 10
    assert open(tmp_path / "run.gam.fasta").read() == PO.render_fasta(want)
    assert open(tmp_path / "run.pctgs").read() == PO.render_descriptors(want, merged, ["m%d" % i for i in range(len(scs))],
                                                                      ["s%d" % i for i in range(len(scs))])


@pytest.mark.parametrize("switch", ["GAMDP_L1_ROUNDS", "GAMDP_L1_ONE_WAVE", "GAMDP_L1_NO_TWINS"])
def test_other_ways_through_a_merge_block_call_in_a_fresh_process(switch):
    """The main chains of the merge blocks run on the device in one launch -- a workgroup per merge block, one wavefront filling
    and two walking (k_chain2), the long chains with a twin workgroup that runs the other orientation at the same time -- and
    the host replays its state machines over the records.  GAMDP_L1_NO_TWINS=1 runs the two orientations one after the other,
    GAMDP_L1_ONE_WAVE=1 keeps the one-wavefront chain kernel (k_chain), GAMDP_L1_ROUNDS=1 every call in the round loop (what
    other bands than 150 take anyway).  All of them must give what the oracle gives: the merge-block tests of this file and the GAGE-shaped ones once more
    in a child per switch (the switches are read once per process)."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_L1_ROUNDS") or os.environ.get("GAMDP_L1_ONE_WAVE") or os.environ.get("GAMDP_L1_NO_TWINS"):
        pytest.skip("already inside the child")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, **{switch: "1"})
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.abspath(__file__), os.path.join(here, "test_gpu_gage.py"),
                        "-k", "merge_blocks_match_oracle or single_merge_block or reference_exception or gage or one_n_around_every_edge_of_a_chain"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2500:] + r.stderr[-2000:]


def _n_edge_scenarios():
    """tests/_cases.py n_edge_cases as merge blocks of ONE block whose frames are the call's windows: the main chain's call is then
    exactly that call (PctgBuilder.cc:1652-1677: begin = frame begin, end = begin + frame length - 1) -- one N at every distance around
    the first and the last base the DP touches on the master and around the slave frame, the alignment running along the band's first
    or last column so that those bases are on the path."""
    import _cases
    scs = []
    for a, b, ba, ea, bb, eb, tag in _cases.n_edge_cases(150):
        scs.append(dict(kind="n-edge %s" % (tag,), master=a.decode(), slave=b.decode(), blocks=[(ba, ea, bb, eb, "+", "+", 10)],
                        tails=(True, True, True, True)))
    return scs


def test_one_n_around_every_edge_of_a_chain_calls_window():
    """N by window for the merge-block chains (round 5): the chain kernel picks the cell of every call by the bases the call touches
    (gamdp_kernel.hip chain_filler, gamdp_dev.h call_touches_n), not by the contigs' flags.  A window computed too small would fill a
    call whose path crosses an N with the N-free cell, which reads an A there: every DP record of every merge block against the oracle's
    driver, through the device chains (default), the one-wavefront chain kernel and the round loop (children of
    test_chain_kernel_variants_in_fresh_processes run this test too)."""
    stats = _check_against_oracle(ctx(), _n_edge_scenarios())
    assert stats["ok"] + stats["bad"] >= 50, stats


def test_a_chain_window_too_small_is_noticed():
    """Fault injection (diagnostics build, GAMDP_DIAG_N_WINDOW_SHRINK): the chain kernel tests windows 66 + 256 bases too small on either
    side.  The host's replay derives the choice of cell of every call by itself, with the full margin, and the call must fail loudly
    ("the device's choice of the N-aware cell is ...") -- never hand back the alignment an N-free cell made of an N."""
    import os, subprocess, sys
    if os.environ.get("GAMDP_DIAG_N_WINDOW_SHRINK"):
        pytest.skip("already inside the child")
    from test_gpu_l0_parity import DIAG_LIB
    code = ("import sys; sys.path.insert(0, %r); import test_gpu_l1_parity as T, gam_ngs_amd as gam\n"
            "from gam_ngs_amd import lib as L\n"
            "scs = T._n_edge_scenarios(); c = gam.Context(0)\n"
            "masters = gam.SequenceSet(c, [s['master'].encode() for s in scs]); slaves = gam.SequenceSet(c, [s['slave'].encode() for s in scs])\n"
            "try:\n"
            "    gam.PctgBuilder(c, masters, slaves).alignMergeBlocks(T.make_mbs(scs), audit=16)\n"
            "    print('NO ERROR')\n"
            "except L.GamdpError as e:\n"
            "    print('ERROR:', e)\n") % os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, GAMDP_LIB=DIAG_LIB, GAMDP_DIAG_N_WINDOW_SHRINK=str(64 + 2 + 256))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "ERROR:" in r.stdout and "choice of the N-aware cell" in r.stdout, r.stdout[-2000:] + r.stderr[-1500:]
    env = dict(os.environ, GAMDP_LIB=DIAG_LIB)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "NO ERROR" in r.stdout, r.stdout[-2000:] + r.stderr[-1500:]
