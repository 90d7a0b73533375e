#!/usr/bin/env python3
"""Benchmark of the hot path on BASELINE.json's workload: GCUPS of the banded semi-global DP on
synthetic 50 kb x 50 kb contig pairs (5 % divergence, band 512; generator of SURVEY.md 8d).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pairs P]

One "step" = one gamdp_align_batch call over this rank's P pairs (fill + end-cell search + traceback
summary for every pair), sequences already packed and resident in HBM.  For N > 1 the driver starts
one process per GPU with torch.distributed.run; every rank aligns its own P pairs (the pair list is
statically partitioned, no data-path collective) -> weak scaling; value = total cells / max-over-ranks time.

The printed JSON line also carries
  roofline      algorithmic HBM bytes (0.2507 B per cell update) of one kernel launch
                divided by that launch's HIP-event duration, against the 8 TB/s HBM peak;
  cpu_baseline  the reference's own find_alignment (oracle/_ref, kind "reference") or, where that build
                is absent, our C restatement (oracle/, kind "port"), timed on this box's host cores on a
                bounded sample of the same pairs.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = 0.2507          # algorithmic HBM bytes per cell update (SURVEY.md 8d / BASELINE.md section 4)
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(length, band, first_pair, budget_pairs):
    """Times the CPU path on `budget_pairs` pairs with all host cores. Test infrastructure (oracle/) is
    used here only as the thing being timed for the reported baseline -- never by the product path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    from gam_ngs_amd import api
    # the reference allocates a 410 MB matrix per in-flight 50 kb pair: cap the pool so the host stays safe
    threads = min(os.cpu_count() or 1, 16)
    ref = O.ref()
    if ref is not None and hasattr(ref, "gamref_bench_pairs"):
        pairs = [api.synth_pair(first_pair + k, length) for k in range(budget_pairs)]
        a = [api.decode(m).encode() for m, _ in pairs]
        b = [api.decode(s).encode() for _, s in pairs]
        arr_a = (C.c_char_p * budget_pairs)(*a)
        arr_b = (C.c_char_p * budget_pairs)(*b)
        la = (C.c_uint64 * budget_pairs)(*[len(x) for x in a])
        lb = (C.c_uint64 * budget_pairs)(*[len(x) for x in b])
        ref.gamref_bench_pairs.restype = C.c_uint64
        ref.gamref_bench_pairs.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(C.c_char_p),
                                           C.POINTER(C.c_uint64), C.c_uint64, C.c_uint64, C.c_int, C.c_void_p]
        t0 = time.time()
        cells = ref.gamref_bench_pairs(arr_a, la, arr_b, lb, budget_pairs, band, threads, None)
        dt = time.time() - t0
        kind = "reference"
    else:
        t0 = time.time()
        cells = O.oracle().gamdp_oracle_bench_pairs(first_pair, budget_pairs, length, band, threads, None)
        dt = time.time() - t0
        kind = "port"
    return {"value": cells / dt / 1e9, "unit": "GCUPS", "cores": threads, "kind": kind,
            "sample": "%d of the same synthetic %d bp pairs (band %d), %d threads, %.1f s" %
                      (budget_pairs, length, band, threads, dt)}


def kernel_name(band):
    """The k_align instantiation the library picks for N-free contigs of this band (gamdp_host.cpp pick_kernel)."""
    n = "true" if os.environ.get("GAMDP_DIAG_FORCE_N") else "false"
    if band == 512:
        return "k_align<17,4,%s>" % n
    if band == 150:
        return "k_align<5,0,%s>" % n
    y = 2 * band + 1
    c = next(c for c in (2, 3, 5, 9, 17) if y <= c * 64)
    return "k_align<%d,-1,true>" % c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=100000,
                    help="pairs per GPU per step (default: all 100 k pairs of BASELINE config 5)")
    ap.add_argument("--len", type=int, default=50000)
    ap.add_argument("--band", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=0, help="pairs in the CPU-baseline sample (0 = 2 per core)")
    args = ap.parse_args()

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
    from gam_ngs_amd import api, lib as L

    ctx = gam.Context(local_rank)
    P, length, band = args.pairs, args.len, args.band
    first = rank * P  # static partition of the pair list: rank r owns pairs [r*P, (r+1)*P)
    # pairs are generated + packed inside the library (same generator as gamdp_synth_pair / the oracle) and
    # uploaded once: sequence 2k = master, 2k+1 = slave of pair first+k
    t_setup = time.perf_counter()
    sset = gam.SequenceSet.synthetic(ctx, first, P, length)
    t_setup = time.perf_counter() - t_setup
    tasks = (L.Task * P)()
    for k in range(P):
        t = tasks[k]
        t.a_id, t.b_id, t.band = 2 * k, 2 * k + 1, band
        t.begin_a, t.end_a, t.begin_b, t.end_b = 0, length - 1, 0, sset.lengths[2 * k + 1] - 1
    out = (L.Result * P)()

    def step():
        rc = ctx.lib.gamdp_align_batch(ctx.handle, sset.handle, sset.handle, tasks, P, out, None)
        if rc != 0:
            raise SystemExit("gamdp_align_batch failed: %d %s" % (rc, ctx.lib.gamdp_last_error(ctx.handle)))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    ctx.kernel_time(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    kernel_ms, launches = ctx.kernel_time()

    cells_rank = sum(out[k].cells for k in range(P))
    bad = sum(1 for k in range(P) if out[k].status != L.ST_OK)
    from gam_ngs_amd import shard
    dt_max, cells_all, bad_all = shard.reduce_step_stats(dt, float(cells_rank), float(bad),
                                                         device="cpu" if share_gpu else "cuda")

    if rank == 0:
        gcups = cells_all * args.steps / dt_max / 1e9
        avg_launch_s = (kernel_ms / 1e3) / max(1, launches)
        cells_per_launch = cells_rank * args.steps / max(1, launches)
        achieved = cells_per_launch * B_ALG / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        # HBM bytes per launch from the PMC counters (FETCH_SIZE/WRITE_SIZE need their own rocprofv3 passes, so
        # they are taken from the committed profile of this same workload; null when the workload differs)
        traffic = traffic_bytes = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            w = tj["workload"]
            if (w["pairs_per_launch"], w["len"], w["band"]) == (P, length, band) and launches == args.steps:
                traffic_bytes = tj["hbm_bytes_per_launch"]
                traffic = traffic_bytes / avg_launch_s / 1e9
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "GCUPS", "value": gcups, "unit": "GCUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "synthetic %d bp x ~%d bp contig pairs, 5%% divergence, band %d "
                                   "(BASELINE.json config 5 generator), find_alignment incl. traceback summary"
                                   % (length, length, band),
                       "pairs_per_gpu_per_step": P, "cells_per_pair": cells_rank // P,
                       "parallelism": "pair list statically partitioned over %d GPU(s), no collective" % world,
                       "failed_pairs": int(bad_all),
                       "one_time_setup_s": round(t_setup, 3)},  # generate + pack + upload the sequences (not timed)
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_bytes_per_launch": traffic_bytes,
                         "algorithmic_bytes_per_launch": cells_per_launch * B_ALG,
                         "kernel": kernel_name(args.band), "kernel_ms_per_launch": avg_launch_s * 1e3,
                         "launches": int(launches), "algorithmic_bytes_per_cell": B_ALG},
        }
        if not args.no_cpu_baseline:
            n_cpu = args.cpu_pairs or 32 * min(os.cpu_count() or 1, 16)
            line["cpu_baseline"] = cpu_baseline(length, band, first, min(n_cpu, P))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
