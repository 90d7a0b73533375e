#!/usr/bin/env python3
"""The `mixed150` record of bench.py on its own (a driver-shaped batch of 100 000 band-150 calls through the planner's own choice):
    python3 tools/mixed150.py [--pairs 12500] [--calls-per-pair 8] [--verify 0]
A/B by environment: GAMDP_NO_PACKED_TOP_MIXED=1 (round 4's rule: packed top blocks only for a shared begin_a), GAMDP_NO_PACKED_TOP=1."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import gam_ngs_amd as gam

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=12500)
ap.add_argument("--calls-per-pair", type=int, default=8)
ap.add_argument("--verify", type=int, default=0)
ap.add_argument("--steps", type=int, default=3)
a = ap.parse_args()
ctx = gam.Context(0)
rec = bench.mixed150_record(ctx, n_pairs=a.pairs, calls_per_pair=a.calls_per_pair, steps=a.steps, verify=a.verify)
print(json.dumps(rec))
