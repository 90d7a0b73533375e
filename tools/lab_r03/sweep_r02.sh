#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
mkdir -p gpurun_out/r02_sweep
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150"
run() { name=$1; shift; $B "$@" > gpurun_out/r02_sweep/$name.log 2>&1; python - gpurun_out/r02_sweep/$name.log $name <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("%-22s gcups %.0f kernel_ms %.1f ms_step %.1f kernel %s"%(sys.argv[2], d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"], d["roofline"]["kernel"]))
PY
}
run b512_100k_50k
run b512_50000 --pairs 50000
run b512_25000 --pairs 25000
run b512_12500 --pairs 12500
run b512_4096 --pairs 4096
run b512_100k_20k --len 20000
run b512_200k_5k --len 5000 --pairs 200000
run b150_100k_50k --band 150
run b150_100k_20k --band 150 --len 20000
run b150_400k_5k --band 150 --len 5000 --pairs 400000
run b150_400k_2k --band 150 --len 2000 --pairs 400000
run b150_4096_50k --band 150 --pairs 4096
run b64_50k --band 64
run b256_50k --band 256
GAMDP_NO_PAIR=1 run b512_nopair
python bench_l1.py --genome 30000000 --steps 3 --verify 0 > gpurun_out/r02_sweep/l1_30mb.log 2>&1; tail -c 900 gpurun_out/r02_sweep/l1_30mb.log
python bench_l1.py --genome 2900000 --steps 5 --verify 0 > gpurun_out/r02_sweep/l1_2p9mb.log 2>&1; tail -c 900 gpurun_out/r02_sweep/l1_2p9mb.log
