#!/bin/bash
# round 6: the shapes of tools/sweep_r05.sh once more (host side of a batch call reworked), and whole launch against pieces for the
# call-heavy ones (GAMDP_CHUNK_MIN=999999999: never in pieces).   tools/sweep_r06.sh <tag> [quick]
set -u
: "${GRAFT_REPO_ROOT:?run under gpurun}"
TAG=${1:-r06_sweep}; QUICK=${2:-}
OUT=gpurun_out/$TAG; mkdir -p $OUT
B="timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --no-mixed150"
run() { name=$1; shift; $B "$@" > $OUT/$name.log 2>&1; python3 - $OUT/$name.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); ok=True; print("%-26s gcups %6.0f kernel_ms %7.1f ms_step %7.1f launches %d kernel %s"%(sys.argv[2], d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"], d["roofline"]["launches"], d["roofline"]["kernel"]))
if not ok: print(sys.argv[2], "FAILED", open(sys.argv[1]).read()[-400:])
PY
}
run b150_400k_5k --band 150 --len 5000 --pairs 400000
GAMDP_CHUNK_MIN=999999999 run b150_400k_5k_whole --band 150 --len 5000 --pairs 400000
run b150_400k_2k --band 150 --len 2000 --pairs 400000
GAMDP_CHUNK_MIN=999999999 run b150_400k_2k_whole --band 150 --len 2000 --pairs 400000
run b150_131k_5k --band 150 --len 5000 --pairs 131072
GAMDP_CHUNK_MIN=999999999 run b150_131k_5k_whole --band 150 --len 5000 --pairs 131072
run b512_200k_5k --len 5000 --pairs 200000
GAMDP_CHUNK_MIN=999999999 run b512_200k_5k_whole --len 5000 --pairs 200000
[ -n "$QUICK" ] && exit 0
run b150_100k_50k --band 150
run b150_100k_20k --band 150 --len 20000
run b512_100k_20k --len 20000
run b512_12500 --pairs 12500
run b150_4096_50k --band 150 --pairs 4096
run b150_12500_50k --band 150 --pairs 12500
run b64_50k --band 64
run b100_50k --band 100
run b256_50k --band 256
run b500_50k --band 500
