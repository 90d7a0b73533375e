#!/usr/bin/env python3
"""Secondary measurement: throughput of the merge-block driver (gamdp_align_merge_blocks, band 150) on
synthetic merge blocks shaped like gam-merge's (tests/_l1cases.py).  Not the headline metric."""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _l1cases  # noqa: E402
import gam_ngs_amd as gam  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--distinct", type=int, default=500, help="distinct scenarios (repeated to reach --n)")
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    rng = random.Random(1)
    base = [_l1cases.scenario(rng, rng.choice(["overlap", "overlap_rev", "contained", "reverse_order", "wrong_vote"]))
            for _ in range(args.distinct)]
    ctx = gam.Context(0)
    masters = gam.SequenceSet(ctx, [s["master"].encode() for s in base])
    slaves = gam.SequenceSet(ctx, [s["slave"].encode() for s in base])
    pb = gam.PctgBuilder(ctx, masters, slaves)

    def make():
        return [gam.MergeBlock(i % args.distinct, i % args.distinct, [gam.Block(*b) for b in base[i % args.distinct]["blocks"]],
                               *base[i % args.distinct]["tails"]) for i in range(args.n)]
    pb.alignMergeBlocks(make())  # warm-up
    ctx.kernel_time(reset=True)
    t_make = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tm = time.perf_counter()
        batch = make()
        t_make += time.perf_counter() - tm
        mbs = pb.alignMergeBlocks(batch)
    dt = (time.perf_counter() - t0 - t_make) / args.steps
    print("python object construction %.1f ms/step (excluded)" % (t_make / args.steps * 1e3))
    kms, kl = ctx.kernel_time()
    cells = sum(m.cells for m in mbs)
    ndp = sum(m.n_dp for m in mbs)
    print("merge blocks/step %d  DP calls %d  cells %.3e  ok %d  step %.1f ms (kernels %.1f ms in %d launches)  -> %.1f GCUPS, %.0f merge blocks/s"
          % (args.n, ndp, cells, sum(m.align_ok for m in mbs), dt * 1e3, kms / args.steps, kl // args.steps, cells / dt / 1e9, args.n / dt))


if __name__ == "__main__":
    main()
