#!/usr/bin/env python3
"""Large seeded parity campaign: GPU (C ABI) vs CPU oracle on random windowed cases, all bands / kernel
variants, with and without edit strings.  Exit code 1 on the first mismatch (prints the case)."""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _cases  # noqa: E402
import _oracle as O  # noqa: E402
from _gpu import oracle_for, run_cases  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=40)
    ap.add_argument("--per-seed", type=int, default=1000)
    ap.add_argument("--long", type=int, default=0, help="additional 3-9 kb related pairs (bands 64..512, N, windows, force flags)")
    ap.add_argument("--first-seed", type=int, default=0, help="continue a campaign: seeds first .. first + seeds - 1 (and another set of long pairs)")
    ap.add_argument("--wide", action="store_true", help="bands beyond the systolic kernels only (k_align_w, round 6): 544 - 4 100")
    args = ap.parse_args()
    band_sets = [(0, 1, 2, 5, 8, 20, 150), (3, 31, 32, 63, 64, 95, 96), (127, 128, 150, 159, 160, 161, 287, 288, 289),
                 (512, 300, 543, 511, 513), (150,), (512,), (7, 40, 70, 100, 200, 256, 400)]
    if args.wide:
        band_sets = [(544, 545, 600, 1000), (2048, 777, 1500), (4100, 550, 3000)]
    total = ok = 0
    for seed in range(args.first_seed, args.first_seed + args.seeds):
        rng = random.Random(77000 + seed)
        bands = band_sets[seed % len(band_sets)]
        max_len = rng.choice([120, 400, 400, 1200, 3000])
        cases = _cases.cases(880000 + seed, args.per_seed if max_len <= 1200 else args.per_seed // 4, max_len=max_len, bands=bands)
        for want_ops in (True, False):
            res = run_cases(cases, want_ops=want_ops)
            for cs, r in zip(cases, res):
                o, ops = oracle_for(cs, want_ops)
                if o.status == O.INVALID:
                    continue
                total += 1
                ok += o.status == O.OK
                if r.key() != o.key() or (want_ops and r.ops != ops):
                    print("MISMATCH seed", seed, "want_ops", want_ops, cs, r.key(), o.key())
                    return 1
        print("seed %d bands %s max_len %d ok (%d checked so far, %d with alignments)" % (seed, bands, max_len, total, ok), flush=True)
    if args.long:
        from concurrent.futures import ThreadPoolExecutor
        rng = random.Random(4242 + args.first_seed)
        cases = []
        for i in range(args.long):
            n = rng.randint(3000, 9000)
            a, b = _cases.related_pair(rng, n, n_frac=rng.choice([0.0, 0.0, 0.005, 0.02]), div=rng.choice([0.5, 1.0, 1.0, 2.0]))
            band = rng.choice([600, 1000, 2048, 544]) if args.wide else rng.choice([150, 150, 512, 512, 64, 300])
            la, lb = len(a), len(b)
            if rng.random() < 0.5:
                ba, bb = rng.randint(0, 600), rng.randint(0, 600)
                ea, eb = la - 1 - rng.randint(0, 600), lb - 1 - rng.randint(0, 600)
            else:
                ba, bb, ea, eb = 0, 0, la - 1, lb - 1
            cases.append(dict(a=a.encode(), b=b.encode(), band=band, begin_a=ba, end_a=ea, begin_b=bb, end_b=eb,
                              fs=rng.random() < 0.3, fe=rng.random() < 0.3))
        for want_ops in (True, False):
            res = run_cases(cases, want_ops=want_ops)
            with ThreadPoolExecutor(16) as ex:
                oracles = list(ex.map(lambda cs: oracle_for(cs, want_ops), cases))
            for cs, r, (o, ops) in zip(cases, res, oracles):
                total += 1
                if r.key() != o.key() or (want_ops and r.ops != ops):
                    print("MISMATCH long", {k: v for k, v in cs.items() if k not in ("a", "b")}, r.key(), o.key())
                    return 1
        print("long cases ok (%d)" % len(cases))
    print("campaign passed: %d comparisons" % total)
    return 0


if __name__ == "__main__":
    sys.exit(main())
