#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3, probe 13: what the int32 top blocks (pos <= 0 region) cost: windows from base 0 against windows from base 600 / 200
B="timeout -s KILL 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy"
run() { name=$1; shift; "$@" 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-30s gcups %.0f kernel_ms %.2f ms_step %.2f kernel %s'%('$name', d['value'], d['roofline']['kernel_ms_per_launch'], d['ms_per_step'], d['roofline']['kernel']))
"; }
run "b512 40960x50k from 0" $B --pairs 40960
run "b512 40960x50k from 600" env GAMDP_BENCH_BEGIN=600 $B --pairs 40960
run "b512 200kx5k from 0" $B --pairs 200000 --len 5000
run "b512 200kx5k from 600" env GAMDP_BENCH_BEGIN=600 $B --pairs 200000 --len 5000
run "b150 400kx5k from 0" $B --band 150 --pairs 400000 --len 5000
run "b150 400kx5k from 200" env GAMDP_BENCH_BEGIN=200 $B --band 150 --pairs 400000 --len 5000
run "b150 400kx2k from 0" $B --band 150 --pairs 400000 --len 2000
run "b150 400kx2k from 200" env GAMDP_BENCH_BEGIN=200 $B --band 150 --pairs 400000 --len 2000
