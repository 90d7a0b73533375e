#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3, probe 5: is it the memory traffic of the packed fill that holds a SIMD with one or two wavefronts back?  (fill only)
export GAMDP_DIAG_SKIP_TRACEBACK=1
for lib in diag diag_nostore diag_nomem diag_src; do for P in 2048 4096 40960; do
GAMDP_LIB=$PWD/gam_ngs_amd/libgamdp_$lib.so python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --pairs $P 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-14s fill only %6d pairs kernel_ms %.2f'%('$lib', d['config']['pairs_total_per_step'], d['roofline']['kernel_ms_per_launch']))
"; done; done
