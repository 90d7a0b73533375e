#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
D=$PWD/gam_ngs_amd/libgamdp_diag.so
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --pairs 40960"
p() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 kernel_ms %.2f'%d['roofline']['kernel_ms_per_launch'])
"; }
GAMDP_LIB=$D $B 2>&1 | p full
GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 $B 2>&1 | p fill
GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 GAMDP_DIAG_COUNT_MAT=1 $B 2>&1 | p fill+mat
GAMDP_LIB=$D GAMDP_DIAG_COUNT_MAT=1 python tools/count_materialise.py 512 4096 50000 2>&1 | tail -5
