#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
for cfg in "" "GAMDP_NO_DF5=1" "GAMDP_L1_ONE_WAVE=1" "GAMDP_L1_ONE_WAVE=1 GAMDP_NO_DF5=1"; do
  for G in 2900000 30000000; do
    env $cfg GAMDP_DIAG_TIMING=1 timeout -s KILL 200 python bench_l1.py --genome $G --steps 3 --verify 0 2>&1 | grep -E "chain: kernel|^\{" | tail -2 | python -c "
import sys, json
ls = sys.stdin.read().strip().split('\n')
k = [l for l in ls if 'chain: kernel' in l]
d = json.loads([l for l in ls if l.startswith('{')][-1])
print('$cfg', $G, 'ms_per_step %.2f' % d['ms_per_step'], k[-1].split('kernel')[1].split(',')[0] if k else '')
"
  done
done
