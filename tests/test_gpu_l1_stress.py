"""The merge-block call's device chain under repetition and fault injection.

The main chain of every merge block runs on the device (k_chain2: a filling and two walking wavefronts per merge block, the
long chains with a twin workgroup for the other orientation; mailboxes in LDS, cancel flags, a pinned host mirror polled by
cohort threads).  Two things are checked here that one passing run does not show:

* every window the device derived is compared by the host's replay (gamdp_l1.cpp replay_chain) with its own derivation: the
  diagnostics build can skew one device-derived start by one base (GAMDP_DIAG_CHAIN_SKEW=k) and the call must then FAIL,
  naming the merge block, not return a plausible result;
* the protocol is timing dependent, so a GAGE-shaped call (about 2 000 merge blocks, the 30 Mb shape of bench.py's l1
  record) is repeated 50 times with twins, and again with 1 / 3 / 16 cohort threads and without twins: every call's
  gamdp_mb_out[] and audit trail must be byte-identical to the first call's, in every process.

Reference: PctgBuilder.cc:1652-1677 (next start = last match + gap), :1420-1509 (retry in the other orientation)."""
import ctypes as C
import hashlib
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DIAG_LIB = os.path.join(ROOT, "gam_ngs_amd", "libgamdp_diag.so")

pytestmark = pytest.mark.gpu

AUDIT = 40


def _raw_call(gam, L, ctx, ms, ss, flat, band=150):
    """one gamdp_align_merge_blocks call; returns (rc, bytes of gamdp_mb_out[], bytes of the audit array)"""
    n = len(flat)
    ins = (L.MbIn * n)()
    keep = []
    for i, mb in enumerate(flat):
        nb = len(mb["blocks"])
        arr = (L.BlockC * max(1, nb))()
        for k, b in enumerate(mb["blocks"]):
            arr[k].m_begin, arr[k].m_end, arr[k].s_begin, arr[k].s_end = b[0], b[1], b[2], b[3]
            arr[k].m_strand, arr[k].s_strand, arr[k].n_reads = b[4].encode(), b[5].encode(), b[6]
        keep.append(arr)
        x = ins[i]
        x.m_id, x.s_id = mb["m_id"], mb["s_id"]
        x.m_ltail, x.m_rtail, x.s_ltail, x.s_rtail = [int(t) for t in mb["tails"]]
        x.n_blocks = nb
        x.blocks = C.cast(arr, C.POINTER(L.BlockC))
    outs = (L.MbOut * n)()
    aud = (L.Result * (n * AUDIT))()

    def call():
        C.memset(outs, 0, C.sizeof(outs))
        C.memset(aud, 0, C.sizeof(aud))
        rc = ctx.lib.gamdp_align_merge_blocks(ctx.handle, ms.handle, ss.handle, ins, n, band, outs, aud, AUDIT)
        return rc, bytes(outs), bytes(aud)
    return call, keep


def _child_main(argv):
    """python test_gpu_l1_stress.py <genome_len> <repeats>: prints one line per call, `rc digest`, then `err <text>` on failure"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    import _gage as G
    import gam_ngs_amd as gam
    from gam_ngs_amd import lib as L
    genome_len, repeats = int(argv[1]), int(argv[2])
    pb = G.problem(int(argv[3]) if len(argv) > 3 else 7, genome_len=genome_len)
    flat, _ = G.merge_blocks(pb)
    ctx = gam.Context(0)
    ms = gam.SequenceSet(ctx, [bytes(x["seq"]) for x in pb["master"]], ascii=False)
    ss = gam.SequenceSet(ctx, [bytes(x["seq"]) for x in pb["slave"]], ascii=False)
    call, _keep = _raw_call(gam, L, ctx, ms, ss, flat)
    print("n_mb %d" % len(flat))
    for _ in range(repeats):
        rc, o, a = call()
        print("%d %s" % (rc, hashlib.sha256(o + a).hexdigest()))
        if rc != 0:
            print("err " + ctx.last_error())
            break
    sys.stdout.flush()


def _run_child(genome_len, repeats, seed=7, **env):
    e = dict(os.environ)
    for k in ("GAMDP_L1_COHORTS", "GAMDP_L1_NO_TWINS", "GAMDP_L1_ROUNDS", "GAMDP_L1_ONE_WAVE", "GAMDP_LIB", "GAMDP_DIAG_CHAIN_SKEW"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), str(genome_len), str(repeats), str(seed)], env=e, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = r.stdout.strip().splitlines()
    n_mb = int(lines[0].split()[1])
    calls = [l.split() for l in lines[1:] if not l.startswith("err ")]
    errs = [l[4:] for l in lines[1:] if l.startswith("err ")]
    return n_mb, calls, errs


def test_repeated_merge_block_calls_are_byte_identical_whatever_the_cohorts_and_twins():
    n_mb, base, errs = _run_child(30_000_000, 50)
    assert not errs and n_mb >= 1500, (n_mb, errs)
    assert len(base) == 50 and all(rc == "0" for rc, _ in base)
    want = base[0][1]
    assert all(d == want for _, d in base), "a repeated call differs from the first: %s" % sorted(set(d for _, d in base))
    # the same call through other splits of the host work and without twins: the same bytes
    for env in (dict(GAMDP_L1_COHORTS="1"), dict(GAMDP_L1_COHORTS="3"), dict(GAMDP_L1_COHORTS="16"), dict(GAMDP_L1_NO_TWINS="1"),
                dict(GAMDP_L1_ROUNDS="1")):
        reps = 3 if "GAMDP_L1_ROUNDS" in env else 12
        _, calls, errs = _run_child(30_000_000, reps, **env)
        assert not errs, (env, errs)
        assert len(calls) == reps and all(rc == "0" and d == want for rc, d in calls), (env, calls[:3], want)


def test_the_result_is_what_the_oracle_gives_after_many_calls():
    """the digest the stress test compares is of a RIGHT answer: the last of 20 calls on a smaller problem, field by field"""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import _gage as G
    import gam_ngs_amd as gam
    from gam_ngs_amd import lib as L
    from _gpu import ctx
    from _l1oracle import oracle_mb
    pb = G.problem(21, genome_len=700_000)
    flat, _ = G.merge_blocks(pb)
    c = ctx()
    ms = gam.SequenceSet(c, [bytes(x["seq"]) for x in pb["master"]], ascii=False)
    ss = gam.SequenceSet(c, [bytes(x["seq"]) for x in pb["slave"]], ascii=False)
    call, _keep = _raw_call(gam, L, c, ms, ss, flat)
    first = None
    for _ in range(20):
        rc, o, a = call()
        assert rc == 0, c.last_error()
        first = first or (o, a)
        assert (o, a) == first
    outs = (L.MbOut * len(flat)).from_buffer_copy(first[0])
    aud = (L.Result * (len(flat) * AUDIT)).from_buffer_copy(first[1])
    for i, mb in enumerate(flat):
        sc = dict(master=G.to_ascii(pb["master"][mb["m_id"]]["seq"]).decode(), slave=G.to_ascii(pb["slave"][mb["s_id"]]["seq"]).decode(),
                  blocks=mb["blocks"], tails=mb["tails"])
        o, oaud = oracle_mb(sc, audit_cap=AUDIT)
        g = outs[i]
        assert (g.status, bool(g.align_ok), bool(g.coords_set), g.n_dp, g.cells) == (o.status, bool(o.align_ok), bool(o.touched), o.n_dp, o.cells)
        if o.touched:
            assert (bool(g.align_rev), g.m_start, g.m_end, g.s_start, g.s_end) == (bool(o.align_rev), o.m_start, o.m_end, o.s_start, o.s_end)
        assert [gam.MyAlignment.from_result(aud[i * AUDIT + k]).key() for k in range(min(AUDIT, g.n_dp))] == oaud
    ms.close(); ss.close()


@pytest.mark.parametrize("skew_call", [0, 1])
def test_a_skewed_device_window_fails_the_call_loudly(skew_call):
    """Fault injection (diagnostics build): the chain kernel starts call `skew_call` of every first attempt one slave base late.
    The record it leaves is a perfectly plausible alignment; the host's replay derives the window by itself, sees the
    difference and fails the call with an internal error that names the merge block and the call."""
    n_mb, calls, errs = _run_child(700_000, 1, seed=21, GAMDP_LIB=DIAG_LIB, GAMDP_DIAG_CHAIN_SKEW=str(skew_call))
    assert n_mb >= 30
    assert len(calls) == 1 and calls[0][0] != "0", calls
    assert errs and "merge block" in errs[0] and ("call %d of its chain" % skew_call) in errs[0] and "begin_b" in errs[0], errs
    # the same build without the switch: the call succeeds (the switch, not the build, is what fails it)
    _, calls, errs = _run_child(700_000, 2, seed=21, GAMDP_LIB=DIAG_LIB)
    assert not errs and all(rc == "0" for rc, _ in calls) and calls[0][1] == calls[1][1]


if __name__ == "__main__":
    _child_main(sys.argv)
