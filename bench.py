#!/usr/bin/env python3
"""Benchmark of the hot path on BASELINE.json's workload: GCUPS of the banded semi-global DP on
synthetic 50 kb x 50 kb contig pairs (5 % divergence, band 512; generator of SURVEY.md 8d).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P] [--scaling strong|weak]

One "step" = one gamdp_align_batch call over this rank's pairs (fill + end-cell search + traceback summary for
every pair), sequences already packed and resident in HBM.  N > 1 = one process per GPU (the driver starts them
with torch.distributed.run; `python bench.py --gpus N` on its own spawns the same thing as a child process).

  --scaling strong (default)  BASELINE config 5 as written: ONE fixed list of P pairs (100 000), dealt over the N
                              GPUs by the library's partitioner (gamdp_partition_lpt on predicted cells: equal
                              weights -> round-robin); value = P pairs' cells / max-over-ranks time.
  --scaling weak              every rank aligns its own P pairs; value = N*P pairs' cells / max-over-ranks time.
With N > 1 the strong-scaling line also carries a short weak-scaling measurement ("weak": {...}).
There is no data-path collective in either mode; RCCL only carries the barrier and three timing scalars.
BENCH_SHARE_GPU=1 (tests/test_gpu_bench_ranks.py): all ranks on GPU 0 over gloo, a bounded scratch arena per rank (--arena-gb) -- the
N > 1 path on a one-GPU box, not a multi-GPU measurement.  With N > 1 rank 0 compares --verify-pairs pairs of ITS share with the CPU
path (verified_pairs); cpu_baseline itself, a timing of the host cores, stays an N = 1 record.

The printed JSON line also carries
  roofline        algorithmic HBM bytes (0.2507 B per cell update) of one kernel launch divided by
                  that launch's HIP-event duration (the library's own stream), against the 8 TB/s HBM peak;
  cpu_baseline    the reference's own find_alignment (oracle/_ref, kind "reference") or, where that build is
                  absent, our C restatement (oracle/, kind "port"), timed on this box's host cores on a bounded
                  sample of the same pairs (rank 0, N = 1 only);
  verified_pairs  how many of those CPU results were compared field for field with the GPU results of the same
                  pairs (outside the timed region); any difference fails the run;
  l1              the merge-block driver (gamdp_align_merge_blocks, band 150) on a GAGE-shaped synthetic
                  two-assembly workload: merge blocks/s, GCUPS, share of the call spent in GPU kernels, rounds;
  band150         the headline's own pairs once more at band 150, gam-merge's live band (N = 1, default workload only):
                  GCUPS, kernel, roofline fraction, and a sample verified against the CPU path like the headline's.
  mixed150        a batch shaped like the live driver's calls (100 000 band-150 calls: random windows, force-flag tails, mixed lengths,
                  a few contigs with N) through the planner's own choice: GCUPS over 8 steps (+ median / min step), the kernel mix the
                  library reports, the time the GPU was busy (the small N-aware launch runs beside the big one), roofline fractions, a verified sample.
  launch_info     what the library says it launched in the last step (gamdp_ctx_launch_info): roofline.kernel comes from there.
  strong8_proxy   (and strong4_proxy) the share ONE GPU gets of the fixed list in a strong-scaling run at N = 8 (4) --
                  pairs 0, 8, 16, ... -- timed on this GPU: GCUPS and projected_Ngpu_factor = N * gcups / value.  A
                  one-GPU stand-in for the scaling curve the driver measures when it has an 8-GPU node; a sample is
                  verified against the CPU path.
roofline.valu is the secondary bound: the kernel is bound by vector-ALU issue, not by HBM.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 0.2507          # algorithmic HBM bytes per cell update (SURVEY.md 8d / BASELINE.md section 4)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s


def reduce_step_stats(dt, cells, failed, device=None):
    """(max over ranks of dt, sum of cells, sum of failed) -- identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return dt, cells, failed
    t = torch.tensor([dt, cells, failed], dtype=torch.float64, device=device)
    tmax = t.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tsum = t.clone()
    dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
    return tmax[0].item(), tsum[1].item(), tsum[2].item()


def rank_share(n_items, rank, world, weights=None):
    """Indices of the fixed list this rank owns: the library's deterministic LPT partition (every rank computes the
    same one).  The benchmark's pairs have equal predicted cells, so this is a round-robin deal."""
    from gam_ngs_amd import api
    part = api.partition_lpt(weights if weights is not None else [1] * n_items, world)
    return [i for i, p in enumerate(part) if p == rank]


def cpu_baseline(length, band, pair_ids):
    """Times the CPU path on the given pairs with all host cores and returns (record, results).  Test infrastructure
    (oracle/) is used here only as the thing being timed for the reported baseline and as the checker of the GPU
    results -- never by the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from gam_ngs_amd import api
    n = len(pair_ids)
    threads, _ = cpu_threads(length, band)
    ref = O.ref()
    keys = []
    if ref is not None and hasattr(ref, "gamref_bench_pairs"):
        pairs = [api.synth_pair(k, length) for k in pair_ids]
        a = [api.decode(m).encode() for m, _ in pairs]
        b = [api.decode(s).encode() for _, s in pairs]
        arr_a, arr_b = (C.c_char_p * n)(*a), (C.c_char_p * n)(*b)
        la, lb = (C.c_uint64 * n)(*[len(x) for x in a]), (C.c_uint64 * n)(*[len(x) for x in b])
        res = (O.RefResult * n)()
        ref.gamref_bench_pairs.restype = C.c_uint64
        ref.gamref_bench_pairs.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_char_p),
                                           C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.c_int, C.POINTER(O.RefResult)]
        t0 = time.time()
        cells = ref.gamref_bench_pairs(arr_a, la, arr_b, lb, n, band, threads, res)
        dt = time.time() - t0
        kind = "reference"
        keys = [O.ref_key(res[i]) for i in range(n)]
    else:
        contiguous = all(pair_ids[i] == pair_ids[0] + i for i in range(n))
        res = (O.OracleResult * n)()
        fn = O.oracle().gamdp_oracle_bench_pairs
        fn.restype = C.c_uint64
        fn.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(O.OracleResult)]
        t0 = time.time()
        if contiguous:
            cells = fn(pair_ids[0], n, length, band, threads, res)
        else:   # the oracle's bench entry takes a contiguous range: a strided sample goes through pair by pair
            one = (O.OracleResult * 1)()
            cells = 0
            for i, k in enumerate(pair_ids):
                cells += fn(k, 1, length, band, 1, one)
                C.memmove(C.byref(res, i * C.sizeof(O.OracleResult)), one, C.sizeof(O.OracleResult))
            threads = 1
        dt = time.time() - t0
        kind = "port"
        keys = [res[i].key() for i in range(n)]
    rec = {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": threads, "hardware_threads": os.cpu_count() or 1,
           "cores_cap": cpu_threads(length, band)[1], "kind": kind,
           "sample": "%d of the same synthetic %d bp pairs (band %d), %d threads, %.1f s" % (n, length, band, threads, dt)}
    return rec, keys


CPU_THREADS_MEASURED = 16   # see cpu_threads()


def cpu_threads(length, band):
    """Threads of the CPU baseline and, when that is fewer than the box has, why.  SURVEY.md 8d asks for T = hardware_concurrency();
    measured on the GPU box (256 hardware threads, round 6, gpurun_out/r6_bench1.json): the reference's find_alignment on 512 of the
    50 kb x band-512 pairs runs at 0.124 GCUPS on 256 threads (211.7 s) against 1.9 - 2.0 GCUPS on 16 (13 s) -- every in-flight pair
    zero-fills a 410 MB matrix of its own (banded_smith_waterman.cc:102-107) and 256 of them fight over page faults and memory
    bandwidth.  The baseline is meant to show the reference at its best on this box, so the pool stays at 16 (BENCH_CPU_THREADS=n
    overrides), never more than half of the available memory allows; the cap and its reason are printed next to `cores`."""
    hw = os.cpu_count() or 1
    want = int(os.environ.get("BENCH_CPU_THREADS", "0")) or min(hw, CPU_THREADS_MEASURED)
    per_pair = 8 * (length + 1) * (2 * band + 1) + (64 << 20)
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        avail = 0
    fit = max(1, int(avail * 0.5 // per_pair)) if avail else want
    threads = max(1, min(want, fit))
    if threads >= hw:
        return threads, None
    if threads < want:
        return threads, "%d of %d hardware threads: the reference holds a %d MB matrix per in-flight pair, half of the available memory fits %d" % (
            threads, hw, per_pair >> 20, fit)
    return threads, ("%d of %d hardware threads: the reference does not scale on this workload -- measured on this pool's 256-thread box: "
                     "0.124 GCUPS on 256 threads against 1.9 - 2.0 on 16 (a %d MB zero-filled matrix per in-flight pair)" % (threads, hw, per_pair >> 20))


def launch_summary(ctx):
    """What the library says the last batch call launched (gamdp_ctx_launch_info): (dominant kernel, records for the line).
    The dominant kernel is the instantiation that took most of the call's kernel time."""
    info = ctx.launch_info()
    by_kernel = {}
    for r in info:
        by_kernel[r["kernel"]] = by_kernel.get(r["kernel"], 0.0) + r["kernel_ms"]
    dominant = max(by_kernel, key=by_kernel.get) if by_kernel else None
    keep = ("kernel", "tasks", "units", "slots", "rounds", "band_max", "units_dirfree", "units_top_wanted", "units_packed_top",
            "units_packed_top_mixed", "strips", "piece", "kernel_ms")
    return dominant, [{k: (round(r[k], 3) if isinstance(r[k], float) else r[k]) for k in keep} for r in info]


def measured_traffic(P_launch, length, band, launches_ok, kernel):
    """HBM bytes per launch from the PMC counters.  FETCH_SIZE / WRITE_SIZE need their own rocprofv3 passes, so the
    figure is REPLAYED from the newest committed profile of exactly this workload (its file and the commit it was
    collected on are named in the line); null when the workload differs or no profile exists."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if name.endswith("_traffic.json"):
            try:
                tj = json.load(open(os.path.join(pdir, name)))
                w = tj["workload"]
                same_kernel = kernel.replace(" ", "").split("<")[0] + "<" in tj.get("kernel", "").replace(" ", "")
                if (w["pairs_per_launch"], w["len"], w["band"]) == (P_launch, length, band) and launches_ok and same_kernel:
                    best = (name, tj)
            except (OSError, KeyError, ValueError):
                pass
    if best is None:
        return None, None, None, None
    name, tj = best
    return tj["hbm_bytes_per_launch"], "profiles/" + name, tj.get("commit"), tj.get("source_hash")


def source_hash():
    """hash of the library's sources as this run sees them (tools/srchash.py)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import srchash
    return srchash.source_hash(ROOT)


PACKED_STREAM_CEILING_GCUPS = 20400.0   # tools/valu_rate5.hip: the bare packed-f16 cell stream on the whole chip


def valu_record(P_launch, length, band, kernel, gcups):
    """The secondary (binding) bound: vector-ALU instructions per cell update from the newest committed SQ_INSTS_VALU pass
    of exactly this workload and kernel (replayed, like `traffic`), and this run's throughput against the measured
    ceiling of the bare packed-f16 cell stream."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for name in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not name.endswith("_traffic.json"):
            continue
        try:
            tj = json.load(open(os.path.join(pdir, name)))
            cj = json.load(open(os.path.join(pdir, name.replace("_traffic.json", "_pmc_counters.json"))))
            w = tj["workload"]
            same_kernel = kernel.replace(" ", "").split("<")[0] + "<" in tj.get("kernel", "").replace(" ", "")
            if (w["pairs_per_launch"], w["len"], w["band"]) == (P_launch, length, band) and same_kernel and "SQ_INSTS_VALU" in cj["counters"]:
                best = (name.replace("_traffic.json", "_pmc_counters.json"), tj, cj)
        except (OSError, KeyError, ValueError):
            pass
    rec = {"bound": "vector-ALU issue", "packed_stream_ceiling_gcups": PACKED_STREAM_CEILING_GCUPS,
           "frac_of_valu_ceiling": gcups / PACKED_STREAM_CEILING_GCUPS, "insts_per_cell": None, "source": None,
           "measured_in_this_run": False}
    if best is not None:
        name, tj, cj = best
        cells = P_launch * (2 * band + 1) * length   # x_size * y_size per pair: the b window is the ~len-base slave
        rec["insts_per_cell"] = cj["counters"]["SQ_INSTS_VALU"] * 64.0 / cells
        rec["source"] = "profiles/%s @%s (SQ_INSTS_VALU x 64 lanes / cell updates of one launch)" % (name, tj.get("commit") or "unrecorded")
        rec["profile_matches_source"] = tj.get("source_hash") == source_hash()
    return rec


def strided_sample(n_total, n_sample):
    """Positions of a sample spread evenly over a list of n_total items (first and last included)."""
    n = min(n_sample, n_total)
    return sorted({round(j * (n_total - 1) / max(1, n - 1)) for j in range(n)}) if n_total else []


def strong_proxy_record(ctx, n_gpus, pairs_total, length, band, value, steps=2, warmup=1, verify=128):
    """Rank 0's share of the fixed list in a strong-scaling run over n_gpus GPUs (pairs 0, n_gpus, 2 n_gpus, ...: what
    gamdp_partition_lpt deals it on equal weights), timed on this GPU."""
    import gam_ngs_amd as gam
    from gam_ngs_amd import lib as L
    P = len(range(0, pairs_total, n_gpus))
    sset = gam.SequenceSet.synthetic(ctx, 0, P, length, stride=n_gpus)
    tasks = (L.Task * max(1, P))()
    for k in range(P):
        t = tasks[k]
        t.a_id, t.b_id, t.band = 2 * k, 2 * k + 1, band
        t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length - 1, 0, sset.lengths[2 * k + 1] - 1
    out = (L.Result * max(1, P))()

    def step():
        rc = ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None)
        if rc != 0:
            raise SystemExit("gamdp_align_batch (strong proxy) failed: %d %s" % (rc, ctx.lib.gamdp_last_error(ctx.handle)))

    for _ in range(warmup):
        step()
    ctx.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    kernel_ms, launches = ctx.kernel_time()
    cells = sum(out[k].cells for k in range(P))
    if any(out[k].status != L.ST_OK for k in range(P)):
        raise SystemExit("bench.py: strong proxy: pairs came back without an alignment")
    gcups = cells / dt / 1e9
    rec = {"pairs": P, "of": pairs_total, "gcups": gcups, "steps": steps, "ms_per_step": dt * 1e3,
           "kernel_ms_per_launch": kernel_ms / max(1, launches), "launches": int(launches),
           "projected_%dgpu_factor" % n_gpus: n_gpus * gcups / value if value else None,
           "note": "one GPU's share of the strong-scaling list at N = %d, measured on ONE GPU; not a multi-GPU measurement" % n_gpus}
    if verify:
        pos = strided_sample(P, verify)
        cpu, cpu_keys = cpu_baseline(length, band, [k * n_gpus for k in pos])
        diff = [j for j, k in enumerate(pos) if tuple(out[k].key()) != tuple(cpu_keys[j])]
        if diff:
            raise SystemExit("bench.py: strong proxy: GPU result of pair %d differs from the CPU %s" % (pos[diff[0]] * n_gpus, cpu["kind"]))
        rec["verified_pairs"] = len(pos)
    sset.close()
    return rec


def band150_record(ctx, m, length, steps=2, warmup=1, verify=128):
    """The pairs of the headline run (still resident) aligned at band 150, gam-merge's only live band
    (banded_smith_waterman.hpp:38): the "band150" object of the line.  `verify` pairs are compared with the CPU path."""
    from gam_ngs_amd import lib as L
    sset, tasks, out = m["_keep"]
    P, band = m["P"], 150
    for k in range(P):
        tasks[k].band = band

    def step():
        rc = ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None)
        if rc != 0:
            raise SystemExit("gamdp_align_batch (band 150) failed: %d %s" % (rc, ctx.lib.gamdp_last_error(ctx.handle)))

    for _ in range(warmup):
        step()
    ctx.kernel_time(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    kernel_ms, launches = ctx.kernel_time()
    dominant, linfo = launch_summary(ctx)   # (of the last step: every step launches the same)
    cells = sum(out[k].cells for k in range(P))
    bad = sum(1 for k in range(P) if out[k].status != L.ST_OK)
    if bad:
        raise SystemExit("bench.py: %d pairs came back without an alignment at band 150" % bad)
    per_launch_s = kernel_ms / 1e3 / max(1, launches)
    rec = {"workload": "the same %d pairs at band 150" % P, "gcups": cells / dt / 1e9, "steps": steps, "ms_per_step": dt * 1e3,
           "kernel": dominant, "launch_info": linfo, "kernel_ms_per_launch": per_launch_s * 1e3, "launches": int(launches),
           "roofline_frac": (cells * steps / max(1, launches)) * B_ALG / per_launch_s / 1e9 / HBM_PEAK_GBS if per_launch_s > 0 else 0.0}
    # counters of this workload, replayed from its committed profile set like the headline's (null when there is none)
    tb, tsrc, tcommit, thash = measured_traffic(P, length, band, launches == steps, rec["kernel"])
    rec["traffic"] = (tb / per_launch_s / 1e9) if tb and per_launch_s > 0 else None
    rec["traffic_bytes_per_launch"] = tb
    rec["traffic_over_algorithmic"] = (tb / (cells * steps / max(1, launches) * B_ALG)) if tb else None
    rec["traffic_source"] = ("replayed from %s @%s" % (tsrc, tcommit or "unrecorded")) if tsrc else None
    rec["profile_matches_source"] = (thash == source_hash()) if tsrc else None
    rec["valu"] = valu_record(P, length, band, rec["kernel"], rec["gcups"])
    if verify:
        pos = strided_sample(P, verify)   # spread over the whole list (its order is also the launch's pairing order)
        ids = [m["first"] + k * m["stride"] for k in pos]
        cpu, cpu_keys = cpu_baseline(length, band, ids)
        diff = [j for j, k in enumerate(pos) if tuple(out[k].key()) != tuple(cpu_keys[j])]
        if diff:
            raise SystemExit("bench.py: band 150: GPU result of pair %d differs from the CPU %s: %r vs %r"
                             % (ids[diff[0]], cpu["kind"], out[pos[diff[0]]].key(), cpu_keys[diff[0]]))
        rec["verified_pairs"] = len(pos)
        rec["verified_sample"] = "every %d-th pair of the list" % max(1, (P - 1) // max(1, len(pos) - 1))
    for k in range(P):
        tasks[k].band = 512
    return rec


def mixed150_record(ctx, n_pairs=12500, calls_per_pair=8, steps=8, warmup=2, verify=256, seed=20261004):
    """A batch shaped like the live driver's calls (tests/_mixed.py: >= 100 000 band-150 calls on contigs of log-normal length 0.3 - 20 kb,
    random windows on both contigs, a tenth of the calls force_start / force_end tails, ~1 % of the contigs with runs of N), through
    the launch planner's own choice: GCUPS, the kernel mix as the library reports it, `verify` calls spread over the list compared
    with the CPU path (the reference's own find_alignment where its build exists, else the oracle)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _mixed
    import _oracle as O
    import gam_ngs_amd as gam
    from gam_ngs_amd import api, lib as L
    seqs, calls = _mixed.mixed_batch(seed, n_pairs, calls_per_pair)
    n = len(calls)
    sset = gam.SequenceSet(ctx, seqs, ascii=False)
    tasks = (L.Task * n)()
    _mixed.fill_tasks(tasks, calls)
    out = (L.Result * n)()

    def step():
        rc = ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, n, out, None)
        if rc != 0:
            raise SystemExit("gamdp_align_batch (mixed150) failed: %d %s" % (rc, ctx.lib.gamdp_last_error(ctx.handle)))

    for _ in range(warmup):
        step()
    ctx.kernel_time(reset=True)
    walls = []
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        step()
        walls.append((time.perf_counter() - t1) * 1e3)
    dt = (time.perf_counter() - t0) / steps
    kernel_ms, launches = ctx.kernel_time()
    dominant, linfo = launch_summary(ctx)
    cells = sum(out[k].cells for k in range(n))
    rows = [c["end_b"] - c["begin_b"] + 1 for c in calls]
    octo = [r for r in linfo if r["kernel"].startswith("k_align_o")]
    units = sum(r["units"] for r in octo)
    rec = {"workload": "%d band-150 calls on %d contig pairs of log-normal length 0.3 - 20 kb: chain calls with random windows, "
                       "%d force_start and %d force_end tail calls, %d calls that start inside the band's left triangle (tests/_mixed.py, seed %d)"
                       % (n, n_pairs, sum(c["fs"] for c in calls), sum(c["fe"] for c in calls), sum(1 for c in calls if c["begin_a"] < 135), seed),
           "calls": n, "mean_rows": sum(rows) / float(n), "gcups": cells / dt / 1e9, "steps": steps, "ms_per_step": dt * 1e3,
           # (a 12 ms step on a 256-thread host: one step in ten or twenty is 4 - 6 ms late; gcups is over the mean of all steps)
           "ms_per_step_median": sorted(walls)[len(walls) // 2], "ms_per_step_min": min(walls),
           # kernel time = the time the GPU was busy with the call's launches: the small N-aware launch runs BESIDE the eight-task
           # launch (gamdp_host.cpp: Ctx::align), so this is the union of their intervals, not the sum of launch_info's kernel_ms
           "kernel_ms_per_step": kernel_ms / steps, "launches_per_step": launches / float(steps), "kernel": dominant, "launch_info": linfo,
           "roofline_frac_in_kernel": (cells * B_ALG / (kernel_ms / steps / 1e3) / 1e9 / HBM_PEAK_GBS) if kernel_ms > 0 else None,
           "roofline_frac_whole_step": cells * B_ALG / dt / 1e9 / HBM_PEAK_GBS,
           # of the eight-task wavefronts whose calls hold blocks with pos <= 0 cells behind the ramp, how many ran them packed
           "packed_top_share": (sum(r["units_packed_top"] for r in octo) / float(max(1, sum(r["units_top_wanted"] for r in octo)))) if units else None,
           "packed_top_units": sum(r["units_packed_top"] for r in octo), "octo_units": units,
           "statuses": {str(st): sum(1 for k in range(n) if out[k].status == st) for st in sorted({out[k].status for k in range(n)})}}
    if verify:
        pos = strided_sample(n, verify)
        ref = O.ref()
        diff = []
        for k in pos:
            c = calls[k]
            a, b = seqs[c["a_id"]][c["a_off"]:], seqs[c["b_id"]]
            if ref is not None:
                r, _ = O.ref_align(api.decode(a).encode(), api.decode(b).encode(), c["band"], c["begin_a"], c["end_a"], c["begin_b"], c["end_b"], c["fs"], c["fe"])
                want = O.ref_key(r)
            else:
                r, _ = O.oracle_align(a, b, c["band"], c["begin_a"], c["end_a"], c["begin_b"], c["end_b"], c["fs"], c["fe"], want_ops=False)
                want = r.key()
            if tuple(out[k].key()) != tuple(want):
                diff.append((k, out[k].key(), want))
        if diff:
            raise SystemExit("bench.py: mixed150: call %d differs from the CPU %s: %r vs %r (%d of %d differ)"
                             % (diff[0][0], "reference" if ref is not None else "port", diff[0][1], diff[0][2], len(diff), len(pos)))
        rec["verified_calls"] = len(pos)
        rec["verified_against"] = "reference" if ref is not None else "port"
    sset.close()
    return rec


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process (this parent has
    not touched the GPU and never will) and leave with its exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=100000,
                    help="strong: pairs in the fixed list (default: all 100 k pairs of BASELINE config 5); weak: per GPU")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--len", type=int, default=50000)
    ap.add_argument("--band", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU-baseline sample (0 = 32 per core)")
    ap.add_argument("--no-l1", action="store_true", help="skip the merge-block (L1, band 150) record")
    ap.add_argument("--l1-genome", type=int, default=2_900_000, help="genome size of the L1 workload (S. aureus: 2.9 Mb)")
    ap.add_argument("--no-band150", action="store_true", help="skip the band-150 record of the same pairs")
    ap.add_argument("--no-proxy", action="store_true", help="skip the strong8_proxy / strong4_proxy records")
    ap.add_argument("--arena-gb", type=float, default=0.0,
                    help="bound the library's scratch arena per rank (GB); default: the library's own budget, or, with "
                         "BENCH_SHARE_GPU=1, free HBM / (2 x ranks) so that ranks sharing one GPU cannot starve each other")
    ap.add_argument("--verify-pairs", type=int, default=128,
                    help="N > 1: pairs of rank 0's share compared with the CPU path (0 = none); N = 1 verifies the cpu_baseline sample")
    ap.add_argument("--no-mixed150", action="store_true", help="skip the mixed150 record (a driver-shaped batch of 100 000 band-150 calls)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # BENCH_SHARE_GPU=1 (testing only): all ranks use GPU 0 and talk over gloo, to exercise the N>1 code path on
    # a single-GPU box; the reported number is then NOT a multi-GPU measurement.
    share_gpu = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if share_gpu else "nccl")
    torch.cuda.set_device(local_rank)

    import gam_ngs_amd as gam
    from gam_ngs_amd import lib as L

    if L.load_library().gamdp_build_info() & 1:
        print("bench.py: WARNING: running on the diagnostics build (GAMDP_LIB); not a product measurement", file=sys.stderr)
    ctx = gam.Context(local_rank)
    length, band = args.len, args.band
    arena_bytes = int(args.arena_gb * 2**30)
    if share_gpu and world > 1 and not arena_bytes:
        arena_bytes = int(torch.cuda.mem_get_info(local_rank)[0] // (2 * world))
    if arena_bytes:
        ctx.set_arena_bytes(arena_bytes)

    def barrier():
        # (RCCL wants to be told the device of a barrier; gloo takes none)
        kw = {} if share_gpu else {"device_ids": [local_rank]}
        if world > 1:
            dist.barrier(**kw)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier(**kw)

    def measure(mode, steps, warmup):
        """One timed run in `mode`; returns the numbers of the JSON line (rank 0 uses them)."""
        if mode == "strong":
            ids = rank_share(args.pairs, rank, world)     # equal weights: pair i -> rank i % world
            first, stride, P = (ids[0] if ids else 0), world, len(ids)
            assert ids == [first + k * stride for k in range(P)]
        else:
            first, stride, P = rank * args.pairs, 1, args.pairs
        t_setup = time.perf_counter()
        # pairs are generated + packed inside the library (same generator as gamdp_synth_pair / the oracle) and
        # uploaded once: sequence 2k = master, 2k+1 = slave of this rank's k-th pair
        sset = gam.SequenceSet.synthetic(ctx, first, P, length, stride=stride)
        t_setup = time.perf_counter() - t_setup
        tasks = (L.Task * max(1, P))()
        # GAMDP_BENCH_BEGIN=n (timing experiment, not the benchmark): every window starts n bases into both contigs -- with n > band no
        # cell of the band lies before the start of `a`, i.e. no block takes the int32 code for the reference's pos == 0 rules
        shift = int(os.environ.get("GAMDP_BENCH_BEGIN", "0"))
        for k in range(P):
            t = tasks[k]
            t.a_id, t.b_id, t.band = 2 * k, 2 * k + 1, band
            t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length - 1, 0, sset.lengths[2 * k + 1] - 1
            if shift:
                t.begin_a, t.begin_b = shift, shift
        out = (L.Result * max(1, P))()

        def step():
            rc = ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None)
            if rc != 0:
                raise SystemExit("gamdp_align_batch failed: %d %s" % (rc, ctx.lib.gamdp_last_error(ctx.handle)))

        for _ in range(warmup):
            step()
        ctx.kernel_time(reset=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        kernel_ms, launches = ctx.kernel_time()
        dominant, linfo = launch_summary(ctx)
        cells_rank = sum(out[k].cells for k in range(P))
        bad = sum(1 for k in range(P) if out[k].status != L.ST_OK)
        dt_max, cells_all, bad_all = reduce_step_stats(dt, float(cells_rank), float(bad), device="cpu" if share_gpu else "cuda")
        m = dict(P=P, first=first, stride=stride, dt_max=dt_max, cells_all=cells_all, bad_all=bad_all, cells_rank=cells_rank,
                 kernel_ms=kernel_ms, launches=launches, t_setup=t_setup, gcups=cells_all * steps / dt_max / 1e9,
                 keys=None, kernel=dominant, launch_info=linfo)
        if rank == 0 and mode == args.scaling:
            m["gpu_keys"] = lambda n: [out[k].key() for k in range(n)]
            m["_keep"] = (sset, tasks, out)
        else:
            sset.close()
        return m

    m = measure(args.scaling, args.steps, args.warmup)
    weak = None
    if world > 1 and args.scaling == "strong":
        w = measure("weak", 1, 1)
        weak = {"value": w["gcups"], "unit": "GCUPS", "pairs_per_gpu_per_step": w["P"], "steps": 1,
                "ms_per_step": w["dt_max"] * 1e3}

    if rank == 0:
        steps = args.steps
        launches = m["launches"]
        avg_launch_s = (m["kernel_ms"] / 1e3) / max(1, launches)
        cells_per_launch = m["cells_rank"] * steps / max(1, launches)
        achieved = cells_per_launch * B_ALG / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        kname = m["kernel"] or "?"   # the library's own account of what it launched (gamdp_ctx_launch_info), not a mirror of its planner
        traffic_bytes, traffic_src, traffic_commit, traffic_hash = measured_traffic(m["P"], length, band, launches == steps, kname)
        line = {
            "metric": "GCUPS", "value": m["gcups"], "unit": "GCUPS", "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": m["dt_max"] / steps * 1e3, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": ("int32 (fast blocks: exact packed-f16 offsets from per-lane int32 bases, two tasks per register; "
                      "results bit-identical to the reference's int64 DP)") if kname.startswith(("k_align_p", "k_align_o"))
                     else "int32 (results bit-identical to the reference's int64 DP)",
            "data": "synthetic",
            "config": {"workload": "synthetic %d bp x ~%d bp contig pairs, 5%% divergence, band %d "
                                   "(BASELINE.json config 5 generator), find_alignment incl. traceback summary"
                                   % (length, length, band),
                       "pairs_total_per_step": int(args.pairs if args.scaling == "strong" else args.pairs * world),
                       "pairs_on_rank0_per_step": m["P"], "cells_per_pair": m["cells_rank"] // max(1, m["P"]),
                       "parallelism": ("one fixed pair list dealt over %d GPU(s) by gamdp_partition_lpt, no collective" % world)
                       if args.scaling == "strong" else ("%d GPU(s) x their own pairs, no collective" % world),
                       "failed_pairs": int(m["bad_all"]),
                       "one_time_setup_s": round(m["t_setup"], 3)},  # generate + pack + upload the sequences (not timed)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (traffic_bytes / avg_launch_s / 1e9) if traffic_bytes and avg_launch_s > 0 else None,
                         "traffic_bytes_per_launch": traffic_bytes, "traffic_measured_in_this_run": False,
                         # were the replayed counters collected on the sources this run uses?  (a hash of gam_ngs_amd/csrc +
                         # include/, recorded with every profile set: tools/srchash.py)
                         "profile_matches_source": (traffic_hash == source_hash()) if traffic_src else None,
                         "source_hash": source_hash(),
                         "traffic_source": ("replayed from %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                            "workload, collected at commit %s), divided by this run's kernel time"
                                            % (traffic_src, traffic_commit or "unrecorded")) if traffic_src else None,
                         "algorithmic_bytes_per_launch": cells_per_launch * B_ALG,
                         "kernel": kname, "kernel_ms_per_launch": avg_launch_s * 1e3,
                         "launches": int(launches), "algorithmic_bytes_per_cell": B_ALG,
                         "valu": valu_record(m["P"], length, band, kname, m["gcups"]) if world == 1 else None},
        }
        line["launch_info"] = m["launch_info"]
        if weak is not None:
            line["weak"] = weak
        if m["bad_all"]:
            if not os.environ.get("GAMDP_BENCH_IGNORE_FAILED"):   # (timing experiments with unusable results)
                raise SystemExit("bench.py: %d pairs came back without an alignment" % int(m["bad_all"]))
            line["results_unusable"] = True
            print("bench.py: WARNING: %d pairs came back without an alignment (GAMDP_BENCH_IGNORE_FAILED): NOT a valid measurement"
                  % int(m["bad_all"]), file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            n_cpu = min(args.cpu_pairs or 512, m["P"])   # ~26 G cell updates: 10 - 15 s of the reference on this box's cores
            # the sample is spread over the whole list (the list order is also the order in which a launch pairs tasks
            # up); only the oracle's own bench entry ("port", no reference build on this box) needs a contiguous range
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import _oracle as O
            pos = strided_sample(m["P"], n_cpu) if O.ref() is not None else list(range(n_cpu))
            ids = [m["first"] + k * m["stride"] for k in pos]
            rec, cpu_keys = cpu_baseline(length, band, ids)
            line["cpu_baseline"] = rec
            all_keys = m["gpu_keys"](m["P"])
            gpu_keys = [all_keys[k] for k in pos]
            diff = [j for j in range(len(pos)) if tuple(gpu_keys[j]) != tuple(cpu_keys[j])]
            if diff:
                k = diff[0]
                raise SystemExit("bench.py: GPU result of pair %d differs from the CPU %s: %r vs %r (%d of %d differ)"
                                 % (ids[k], rec["kind"], gpu_keys[k], cpu_keys[k], len(diff), len(pos)))
            line["verified_pairs"] = len(pos)   # status, begin, score, #matches, length, first/last match, identity
            line["verified_sample"] = "every %d-th pair of the list" % max(1, (m["P"] - 1) // max(1, len(pos) - 1))
        if world > 1 and args.verify_pairs and not args.no_cpu_baseline:
            # N > 1: the line is certified too -- a sample spread over RANK 0's share against the CPU path (the baseline
            # itself, a timing of the host cores, stays an N = 1 record: N ranks would time the same cores N times)
            pos = strided_sample(m["P"], min(args.verify_pairs, m["P"]))
            ids = [m["first"] + k * m["stride"] for k in pos]
            rec, cpu_keys = cpu_baseline(length, band, ids)
            all_keys = m["gpu_keys"](m["P"])
            diff = [j for j, k in enumerate(pos) if tuple(all_keys[k]) != tuple(cpu_keys[j])]
            if diff:
                j = diff[0]
                raise SystemExit("bench.py: rank 0: GPU result of pair %d differs from the CPU %s: %r vs %r (%d of %d differ)"
                                 % (ids[j], rec["kind"], all_keys[pos[j]], cpu_keys[j], len(diff), len(pos)))
            line["verified_pairs"] = len(pos)
            line["verified_against"] = rec["kind"]
            line["verified_sample"] = "every %d-th pair of rank 0's share (pairs %d, %d, ...)" % (
                max(1, (m["P"] - 1) // max(1, len(pos) - 1)), m["first"], m["first"] + m["stride"])
        if arena_bytes:
            line["config"]["arena_bytes_per_rank"] = arena_bytes
        if share_gpu and world > 1:
            line["config"]["note"] = "BENCH_SHARE_GPU=1: %d ranks on ONE GPU over gloo -- a test of the N > 1 path, not a multi-GPU measurement" % world
        if not args.no_band150 and world == 1 and band == 512 and "_keep" in m:
            line["band150"] = band150_record(ctx, m, length, verify=0 if args.no_cpu_baseline else 128)
        if not args.no_proxy and world == 1 and band == 512 and args.scaling == "strong" and args.pairs >= 64:
            if "_keep" in m:
                m["_keep"][0].close()   # the headline's sequences: make room
            for ng in (8, 4):
                line["strong%d_proxy" % ng] = strong_proxy_record(ctx, ng, args.pairs, length, band, m["gcups"],
                                                                  verify=0 if args.no_cpu_baseline else 128)
        if not args.no_mixed150 and world == 1 and band == 512:
            line["mixed150"] = mixed150_record(ctx, verify=0 if args.no_cpu_baseline else 256)
        if not args.no_l1 and world == 1:
            import bench_l1
            line["l1"] = bench_l1.run(ctx, genome=args.l1_genome, cpu=not args.no_cpu_baseline)
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
