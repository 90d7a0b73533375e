#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3, probe 11: the packed stream without its two hazard nops per row-time -- headline and short launches
B="timeout -s KILL 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy"
for P in 100000 12500 4096 2048; do
 $B --pairs $P 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('pairs %6d gcups %.0f kernel_ms %.2f ms_step %.2f'%($P, d['value'], d['roofline']['kernel_ms_per_launch'], d['ms_per_step']))
"
done
