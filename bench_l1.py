#!/usr/bin/env python3
"""The merge-block (L1) record of bench.py: gamdp_align_merge_blocks at band 150 (gam-merge's only live band,
banded_smith_waterman.hpp:38) on the GAGE-shaped synthetic two-assembly workload of tests/_gage.py (stand-in for
BASELINE configs 1-4, whose data cannot be fetched here).  One step = ONE call over all merge blocks of all graphs.

    python bench_l1.py [--genome 2900000] [--steps 5]          # on its own; bench.py embeds run() as "l1": {...}
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def marshal(flat):
    """merge blocks (dicts of tests/_gage.py) -> (gamdp_mb_in array, keep-alive list)"""
    from gam_ngs_amd import lib as L
    n = len(flat)
    ins = (L.MbIn * max(1, n))()
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
    return ins, keep


B_ALG = 0.2507          # algorithmic HBM bytes per cell update
HBM_PEAK_GBS = 8000.0


def cpu_baseline(pb, flat, band):
    """The CPU path beside it: the oracle's restatement of alignMergeBlock (oracle/gamdp_oracle.c, kind "port" -- PctgBuilder.cc
    itself cannot be built here) on the SAME merge blocks, a pool of host threads pulling merge blocks from a shared list the
    way ThreadedBuildPctg.cc:50-74 deals graphs.  At most ~600 M cells of it (the 2.9 Mb workload whole; a spread sample of a
    bigger one), so that the default bench run stays within minutes.  Test infrastructure timed as a reported baseline,
    never part of the product path."""
    from concurrent.futures import ThreadPoolExecutor
    import _gage
    import _oracle as O
    threads = min(os.cpu_count() or 1, 32)
    n = len(flat)
    budget = 600e6
    order = list(range(n))
    total = sum(sum(b[3] - b[2] + 1 for b in mb["blocks"]) * (2 * band + 1) for mb in flat)
    step = max(1, int(total / budget + 0.999))
    order = order[::step]
    enc_m, enc_s = {}, {}
    jobs = []
    for i in order:
        mb = flat[i]
        if mb["m_id"] not in enc_m:
            enc_m[mb["m_id"]] = O.encode(_gage.to_ascii(pb["master"][mb["m_id"]]["seq"]))
        if mb["s_id"] not in enc_s:
            enc_s[mb["s_id"]] = O.encode(_gage.to_ascii(pb["slave"][mb["s_id"]]["seq"]))
        nb = len(mb["blocks"])
        arr = (O.OracleBlock * max(1, nb))()
        for k, b in enumerate(mb["blocks"]):
            arr[k].m_begin, arr[k].m_end, arr[k].s_begin, arr[k].s_end = b[0], b[1], b[2], b[3]
            arr[k].m_strand, arr[k].s_strand, arr[k].n_reads = b[4].encode(), b[5].encode(), b[6]
        o = O.OracleMB()
        o.m_ltail, o.m_rtail, o.s_ltail, o.s_rtail = [int(x) for x in mb["tails"]]
        jobs.append((enc_m[mb["m_id"]], enc_s[mb["s_id"]], arr, nb, o))
    fn = O.oracle().gamdp_oracle_align_merge_block

    def one(j):
        m, s, arr, nb, o = j
        fn(m, len(m), s, len(s), arr, nb, band, C.byref(o), None, 0)   # (ctypes releases the GIL for the call)
        return o.cells
    with ThreadPoolExecutor(max_workers=threads) as ex:
        t0 = time.perf_counter()
        cells1 = sum(ex.map(one, jobs))
        dt1 = time.perf_counter() - t0
        # ~15 s of CPU work in all: the sample over and over (a pass of the 2.9 Mb workload is a tenth of a second on 32 threads)
        reps = max(1, min(100, int(15.0 / max(dt1 * threads, 1e-3))))
        t0 = time.perf_counter()
        cells = 0
        for _ in range(reps):
            cells += sum(ex.map(one, jobs))
        dt = time.perf_counter() - t0
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "merge_blocks_per_s": reps * len(jobs) / dt, "cores": threads, "kind": "port",
            "sample": "%d of the %d merge blocks (every %d-th) x %d passes, %d threads, %.1f s" % (len(jobs), n, step, reps, threads, dt)}


def replayed_counters(genome):
    """PMC counters of the chain kernel for this workload from the newest committed profile set (profiles/*_l1_<w>_pmc_counters.json,
    tools/collect_l1_profiles.sh): replayed, and labelled with the sources they were collected on."""
    w = {2_900_000: "2p9mb", 30_000_000: "30mb"}.get(genome)
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(pdir)) if (w and os.path.isdir(pdir)) else []:
        if name.endswith("_l1_%s_pmc_counters.json" % w):
            best = name
    if best is None:
        return None
    pj = json.load(open(os.path.join(pdir, best)))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import srchash
    return {"source": "profiles/%s @%s" % (best, pj.get("commit") or "unrecorded"), "profile_matches_source": pj.get("source_hash") == srchash.source_hash(ROOT),
            "hbm_bytes_per_launch": pj.get("hbm_bytes_per_launch"), "traffic_over_algorithmic": pj.get("traffic_over_algorithmic"),
            "valu_insts_per_cell": pj.get("valu_insts_per_cell"), "measured_in_this_run": False}


def run(ctx, genome=2_900_000, steps=5, warmup=1, seed=1, verify=24, band=150, cpu=True):
    """Returns the "l1" object of the bench line.  `verify` merge blocks are checked against the CPU oracle's driver
    (oracle/, checker only) outside the timed region; a difference fails the run."""
    import _gage
    import gam_ngs_amd as gam
    from gam_ngs_amd import lib as L
    t0 = time.perf_counter()
    pb = _gage.problem(seed, genome_len=genome)
    flat, _ = _gage.merge_blocks(pb)
    masters = gam.SequenceSet(ctx, [bytes(c["seq"]) for c in pb["master"]], ascii=False)
    slaves = gam.SequenceSet(ctx, [bytes(c["seq"]) for c in pb["slave"]], ascii=False)
    ins, keep = marshal(flat)
    n = len(flat)
    outs = (L.MbOut * max(1, n))()
    t_setup = time.perf_counter() - t0

    def step():
        rc = ctx.lib.gamdp_align_merge_blocks(ctx.handle, masters.handle, slaves.handle, ins, n, band, outs, None, 0)
        if rc != 0:
            raise SystemExit("gamdp_align_merge_blocks failed: %d %s" % (rc, ctx.lib.gamdp_last_error(ctx.handle)))

    for _ in range(warmup):
        step()
    acc = dict(wall=0.0, busy=0.0, ksum=0.0, pend=0.0, feed=0.0)
    st = L.L1Stats()
    t1 = time.perf_counter()
    for _ in range(steps):
        ts = time.perf_counter()
        step()
        ctx.lib.gamdp_ctx_l1_stats(ctx.handle, C.byref(st))
        if os.environ.get("GAMDP_BENCH_L1_STEPS"):
            print("step: %.3f ms around the call, %.3f ms inside the library" % ((time.perf_counter() - ts) * 1e3, st.wall_ms), file=sys.stderr)
        acc["wall"] += st.wall_ms; acc["busy"] += st.gpu_busy_ms; acc["ksum"] += st.kernel_sum_ms
        acc["pend"] += st.host_pending_ms; acc["feed"] += st.host_feed_ms
    dt = (time.perf_counter() - t1) / steps
    rec = {
        "workload": "GAGE-shaped synthetic: %.1f Mb genome, %d + %d contigs, %d merge blocks in %d graphs, band %d"
                    % (genome / 1e6, len(pb["master"]), len(pb["slave"]), n, len(pb["graphs"]), band),
        "merge_blocks": n, "dp_calls": int(st.dp_calls), "cells": int(st.cells), "steps": steps,
        "ms_per_step": dt * 1e3, "merge_blocks_per_s": n / dt, "gcups": st.cells / dt / 1e9,
        "gpu_busy_frac": acc["busy"] / acc["wall"] if acc["wall"] else 0.0,
        "gpu_busy_ms": acc["busy"] / steps, "kernel_sum_ms": acc["ksum"] / steps,
        "host_pending_ms": acc["pend"] / steps, "host_feed_ms": acc["feed"] / steps,
        "rounds": int(st.rounds), "cohorts": int(st.cohorts), "launches": int(st.launches),
        "align_ok": sum(1 for i in range(n) if outs[i].align_ok), "setup_s": round(t_setup, 2),
        # the HBM roofline of the whole call (0.2507 B per cell update against 8 TB/s) and of the time its kernels ran: a call
        # lasts as long as its longest chain at a lone wavefront's speed, so both are tiny -- stated, not hidden
        "roofline_frac": st.cells * B_ALG / dt / 1e9 / HBM_PEAK_GBS,
        "roofline_frac_gpu_busy": (st.cells * B_ALG / (acc["busy"] / steps / 1e3) / 1e9 / HBM_PEAK_GBS) if acc["busy"] else None,
        "counters": replayed_counters(genome),
    }
    if cpu:
        rec["cpu_baseline"] = cpu_baseline(pb, flat, band)
    if verify:
        from _l1oracle import oracle_mb
        # a sample across the whole size distribution (every (n-1)/(verify-1)-th merge block of the list sorted by block
        # span: the cheapest, the largest -- multi-round retries, tails, the long band-150 calls -- and what lies
        # between); every decision field is compared
        by_span = sorted(range(n), key=lambda i: sum(b[1] - b[0] for b in flat[i]["blocks"]))
        k = min(verify, n)
        order = sorted({by_span[round(j * (n - 1) / max(1, k - 1))] for j in range(k)}) if n else []
        for i in order:
            mb = flat[i]
            sc = dict(master=_gage.to_ascii(pb["master"][mb["m_id"]]["seq"]).decode(),
                      slave=_gage.to_ascii(pb["slave"][mb["s_id"]]["seq"]).decode(), blocks=mb["blocks"], tails=mb["tails"])
            o, _ = oracle_mb(sc, band=band, audit_cap=1)
            got = (outs[i].status, bool(outs[i].align_ok), bool(outs[i].coords_set), outs[i].n_dp, outs[i].cells)
            want = (o.status, bool(o.align_ok), bool(o.touched), o.n_dp, o.cells)
            if o.touched:
                got += (bool(outs[i].align_rev), outs[i].m_start, outs[i].m_end, outs[i].s_start, outs[i].s_end)
                want += (bool(o.align_rev), o.m_start, o.m_end, o.s_start, o.s_end)
            if got != want:
                raise SystemExit("bench_l1: merge block %d differs from the CPU oracle: %r vs %r" % (i, got, want))
        rec["verified_merge_blocks"] = len(order)
        rec["verified_sample"] = "every %d-th merge block of the list sorted by block span, smallest and largest included" % max(1, (n - 1) // max(1, k - 1))
    masters.close()
    slaves.close()
    return rec


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=2_900_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--verify", type=int, default=24)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import gam_ngs_amd as gam
    print(json.dumps(run(gam.Context(0), genome=a.genome, steps=a.steps, seed=a.seed, verify=a.verify, cpu=not a.no_cpu_baseline)))
