"""A large heterogeneous batch through the launch planner's OWN choice (no environment switch), against the oracle: 16 384 band-150
calls shaped like the live driver's (tests/_mixed.py: chain calls with random windows on both contigs, force_end left tails from base 0,
force_start right tails on chop_begin views, contigs of log-normal length, a few with runs of N) -- mixed lengths, mixed begin_a, force
flags and N-by-window splits in ONE gamdp_align_batch call (the committed suite had no such batch).  And the library's own
account of what it launched (gamdp_ctx_launch_info) for shapes on either side of the planner's thresholds.

Reference: BandedSmithWaterman::find_alignment (lib/src/alignment/banded_smith_waterman.cc:69-322) as PctgBuilder::alignBlocks /
findBestAlignment call it (lib/src/pctg/PctgBuilder.cc:1535-1611, 1652-1677).
"""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor
import os

import pytest

import _mixed
import _oracle as O
from _gpu import ctx
import gam_ngs_amd as gam
from gam_ngs_amd import lib as L

pytestmark = pytest.mark.gpu


def run_batch(c, sset, calls):
    tasks = (L.Task * len(calls))()
    _mixed.fill_tasks(tasks, calls)
    out = (L.Result * len(calls))()
    rc = c.lib.gamdp_align_batch(c.handle, sset.handle, sset.handle, tasks, len(calls), out, None)
    assert rc == 0, c.last_error()
    return out


def oracle_keys(seqs, calls):
    def one(cl):
        a, b = seqs[cl["a_id"]][cl["a_off"]:], seqs[cl["b_id"]]
        r, _ = O.oracle_align(a, b, cl["band"], cl["begin_a"], cl["end_a"], cl["begin_b"], cl["end_b"], cl["fs"], cl["fe"], want_ops=False)
        return r.key()
    with ThreadPoolExecutor(min(32, os.cpu_count() or 1)) as ex:   # (the oracle's C code runs outside the GIL)
        return list(ex.map(one, calls))


def test_driver_shaped_batch_of_16384_calls_through_the_planners_choice():
    c = ctx()
    seqs, calls = _mixed.mixed_batch(20261004, 2048, 8)
    assert len(calls) == 16384
    sset = gam.SequenceSet(c, seqs, ascii=False)
    out = run_batch(c, sset, calls)
    info = c.launch_info()
    want = oracle_keys(seqs, calls)
    bad = [i for i in range(len(calls)) if tuple(out[i].key()) != tuple(want[i])]
    assert not bad, (len(bad), calls[bad[0]], out[bad[0]].key(), want[bad[0]])
    # the batch is what it claims to be: force flags, windows that start inside the band's left triangle, calls on contigs with N,
    # and alignments (not only settled calls)
    assert sum(cl["fs"] for cl in calls) > 400 and sum(cl["fe"] for cl in calls) > 400
    assert sum(1 for cl in calls if cl["begin_a"] < 135) > 2000
    assert sum(1 for k in want if k[0] == O.OK) > 15000
    # ... and went where the planner sends such a batch: the eight-task packed kernel for the N-free calls, an N-aware launch for the rest
    kernels = {r["kernel"] for r in info}
    assert "k_align_o<19,15>" in kernels, info
    assert any(r["n_aware"] for r in info), info
    octo = [r for r in info if r["kernel"] == "k_align_o<19,15>"]
    units = sum(r["units"] for r in octo)
    assert sum(r["units_dirfree"] for r in octo) >= 0.9 * units, octo
    sset.close()


def test_driver_shaped_batch_with_many_tail_calls():
    """Half of the calls force their start or their end, contigs of 0.2 - 12 kb: the packed top blocks of force_start calls (a second
    v_pk_max per cell for the chain, pair_top_range<FS>) and calls that start deep inside the band's left triangle, in wavefronts
    whose tasks differ in begin_a.  (tools/mixed_stress.py runs the same comparison over more seeds and mixes.)"""
    c = ctx()
    seqs, calls = _mixed.mixed_batch(777, 2048, 8, force_frac=0.5, len_lo=200, len_hi=12000)
    sset = gam.SequenceSet(c, seqs, ascii=False)
    out = run_batch(c, sset, calls)
    octo = [r for r in c.launch_info() if r["kernel"] == "k_align_o<19,15>"]
    assert octo and sum(r["units_packed_top_mixed"] for r in octo) > 500, c.launch_info()   # (the eight-task kernel, top blocks of wavefronts whose calls differ)
    want = oracle_keys(seqs, calls)
    bad = [i for i in range(len(calls)) if tuple(out[i].key()) != tuple(want[i])]
    assert not bad, (len(bad), calls[bad[0]], out[bad[0]].key(), want[bad[0]])
    assert sum(cl["fs"] for cl in calls) > 3000 and sum(cl["fe"] for cl in calls) > 3000
    assert sum(1 for k in want if k[0] == O.OK) > 12000
    sset.close()


@pytest.mark.parametrize("n_calls,length,band,want", [
    (12288, 2000, 150, "k_align_o<19,15>"),      # a full batch of N-free band-150 calls of >= 1 k rows: eight tasks per wavefront
    (12288 - 64, 2000, 150, "k_align<5,0,false>"),   # below 12 288 calls (and 2.5 k rows): one task per wavefront
    (12288, 900, 150, "k_align<5,0,false>"),     # 12 288 calls, but under 1 k rows on average
    (16384, 600, 150, "k_align_o<19,15>"),       # from 16 384 calls on: at every length
    (6144, 9000, 150, "k_align_o<19,15>"),       # long contigs: from 6 144 calls on
    (8192, 3000, 150, "k_align_o<19,15>"),       # ... from 8 192 calls of >= 2.5 k rows
    (8192, 2000, 150, "k_align<5,0,false>"),
    (6144, 5000, 150, "k_align_o<19,15>"),       # ... from 6 144 of >= 4.5 k
    (6144, 4000, 150, "k_align<5,0,false>"),
    (64, 3000, 512, "k_align_p<17,4>"),          # band 512 without N: two tasks per wavefront
    (64, 3000, 500, "k_align<17,-1,true>"),      # any other band: the generic kernels
])
def test_the_library_says_what_it_launched(n_calls, length, band, want):
    c = ctx()
    n_pairs = 64
    sset = gam.SequenceSet.synthetic(c, 777, n_pairs, length)
    tasks = (L.Task * n_calls)()
    for k in range(n_calls):
        p = k % n_pairs
        t = tasks[k]
        t.a_id, t.b_id, t.band = 2 * p, 2 * p + 1, band
        t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length - 1, 0, sset.lengths[2 * p + 1] - 1
    out = (L.Result * n_calls)()
    assert c.lib.gamdp_align_batch(c.handle, sset.handle, sset.handle, tasks, n_calls, out, None) == 0, c.last_error()
    info = c.launch_info()
    assert [r["kernel"] for r in info] == [want], info
    r = info[0]
    assert r["tasks"] == n_calls and r["units"] * r["tasks_per_wavefront"] >= n_calls and r["slots"] >= 1 and r["kernel_ms"] > 0
    assert abs(r["rounds"] - r["units"] / r["slots"]) < 1e-9 and r["band_max"] == band
    if want.startswith(("k_align_o", "k_align_p")):
        assert r["units_dirfree"] == r["units"] and r["units_packed_top"] == r["units"], r   # whole-contig calls from base 0: packed top blocks
        assert r["strips"] >= n_calls, r                                                  # every walk re-creates at least one strip
    # results of equal calls are equal (and real)
    assert all(out[k].status == L.ST_OK and tuple(out[k].key()) == tuple(out[k % n_pairs].key()) for k in range(n_calls))
    sset.close()


def test_small_n_aware_launch_beside_the_big_one():
    """Round 6: a batch big enough for two rounds or more of the eight-task kernel whose few calls on contigs with N make one small
    launch of the one-task N-aware kernel -- that launch runs BESIDE the big one (a second stream, its own region of the scratch arena;
    Ctx::align in gamdp_host.cpp).  Every call of the small launch and a sample of the others against the oracle; and the same batch once
    more in a child process with GAMDP_NO_AUX_LAUNCH=1 (one launch after the other): identical results, call for call."""
    import hashlib, subprocess, sys
    c = ctx()
    seqs, calls = _mixed.mixed_batch(606, 8704, 8)   # 69 632 calls: 2.1 rounds of eight-task wavefronts
    sset = gam.SequenceSet(c, seqs, ascii=False)
    out = run_batch(c, sset, calls)
    info = c.launch_info()
    small = [r for r in info if r["tasks_per_wavefront"] == 1 and r["n_aware"]]
    big = [r for r in info if r["kernel"] == "k_align_o<19,15>"]
    assert len(info) == 2 and small and big and big[0]["rounds"] >= 2.0 and small[0]["rounds"] <= 1.0, info
    has_n = [4 in s for s in seqs]
    picked = [i for i, cl in enumerate(calls) if has_n[cl["a_id"]] or has_n[cl["b_id"]] or i % 97 == 0]
    assert sum(1 for i in picked if has_n[calls[i]["a_id"]] or has_n[calls[i]["b_id"]]) >= small[0]["tasks"] > 100
    want = oracle_keys(seqs, [calls[i] for i in picked])
    bad = [i for i, w in zip(picked, want) if tuple(out[i].key()) != tuple(w)]
    assert not bad, (len(bad), calls[bad[0]], out[bad[0]].key())
    digest = hashlib.sha256(repr([tuple(out[i].key()) for i in range(len(calls))]).encode()).hexdigest()
    sset.close()
    if os.environ.get("GAMDP_MIXED_DIGEST_ONLY"):
        print("DIGEST", digest)
        return
    env = dict(os.environ, GAMDP_NO_AUX_LAUNCH="1", GAMDP_MIXED_DIGEST_ONLY="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-s", "-m", "gpu", os.path.abspath(__file__), "-k", "small_n_aware_launch"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert ("DIGEST " + digest) in r.stdout, "results differ between the side-by-side launches and one after the other"
