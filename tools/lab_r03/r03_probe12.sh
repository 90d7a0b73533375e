#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# A/B on one box: diagnostics builds with the packed stream before / after the reordering (alternating, two passes).
# libgamdp_diag_oldpair.so = `make -C gam_ngs_amd/csrc variant NAME=oldpair` with kernel_pair.inc of the commit before (not kept in the tree).
# Result: no difference beyond noise (13 105 - 13 116 / 10 430 - 10 820 / 10 440 - 10 500 GCUPS at 100 000 / 12 500 / 4 096 pairs, both).
B="timeout -s KILL 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy"
for pass in 1 2; do
for lib in diag_oldpair diag; do
for P in 100000 12500 4096; do
 GAMDP_LIB=$PWD/gam_ngs_amd/libgamdp_$lib.so $B --pairs $P 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-14s pairs %6d gcups %.0f kernel_ms %.2f'%('$lib', $P, d['value'], d['roofline']['kernel_ms_per_launch']))
"
done; done; done
