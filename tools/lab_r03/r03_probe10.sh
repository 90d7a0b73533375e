#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3, probe 10: the one-task band-150 kernel with direction-free fast blocks (default) and without (GAMDP_NO_DF5=1), where batches use it
B="timeout -s KILL 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --band 150"
run() { name=$1; shift; "$@" 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-28s gcups %.0f kernel_ms %.2f ms_step %.2f kernel %s'%('$name', d['value'], d['roofline']['kernel_ms_per_launch'], d['ms_per_step'], d['roofline']['kernel']))
"; }
for cfg in "X=1" "GAMDP_NO_DF5=1"; do
  run "4096x50k $cfg" env $cfg $B --pairs 4096
  run "1024x50k $cfg" env $cfg $B --pairs 1024
  run "400kx2k $cfg" env $cfg $B --pairs 400000 --len 2000
  run "20kx5k $cfg" env $cfg $B --pairs 20000 --len 5000
done
