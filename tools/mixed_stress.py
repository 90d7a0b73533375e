#!/usr/bin/env python3
"""Driver-shaped batches with MANY force_start / force_end calls and short contigs, against the oracle (GPU box, from the repo root):
    python3 tools/mixed_stress.py [first_seed] [seeds] [pairs] [force_frac]
The committed test (tests/test_gpu_mixed_batch.py) has one seed with 10 % tail calls; this is the same comparison over other seeds and
mixes -- the packed top blocks of force_start calls (two v_pk_max per cell) and calls that start deep inside the band's left triangle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _mixed
import _oracle as O
import gam_ngs_amd as gam
from gam_ngs_amd import lib as L
from test_gpu_mixed_batch import run_batch, oracle_keys

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ff = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
c = gam.Context(0)
total = 0
for seed in range(first, first + seeds):
    seqs, calls = _mixed.mixed_batch(seed, pairs, 8, force_frac=ff, len_lo=200, len_hi=12000)
    sset = gam.SequenceSet(c, seqs, ascii=False)
    out = run_batch(c, sset, calls)
    kernels = sorted({r["kernel"] for r in c.launch_info()})
    want = oracle_keys(seqs, calls)
    bad = [i for i in range(len(calls)) if tuple(out[i].key()) != tuple(want[i])]
    if bad:
        print("MISMATCH seed", seed, len(bad), calls[bad[0]], out[bad[0]].key(), want[bad[0]]); sys.exit(1)
    total += len(calls)
    print("seed %d: %d calls (%d force_start, %d force_end, %d inside the triangle) bit-exact; kernels %s" % (
        seed, len(calls), sum(cl["fs"] for cl in calls), sum(cl["fe"] for cl in calls), sum(1 for cl in calls if cl["begin_a"] < 150), kernels), flush=True)
    sset.close()
print("mixed stress passed: %d comparisons" % total)
