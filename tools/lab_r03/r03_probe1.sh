#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3, probe 1: issue rates at 1-4 wavefronts per SIMD (micro-benchmark 6) and the short-launch baseline of the library
mkdir -p gpurun_out/r03_probe1
./tools/valu_rate6 > gpurun_out/r03_probe1/valu_rate6.log 2>&1
cat gpurun_out/r03_probe1/valu_rate6.log
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150"
run() { name=$1; shift; $B "$@" > gpurun_out/r03_probe1/$name.log 2>&1; python - gpurun_out/r03_probe1/$name.log $name <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("%-22s gcups %.0f kernel_ms %.1f ms_step %.1f kernel %s"%(sys.argv[2], d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"], d["roofline"]["kernel"]))
PY
}
run b512_100k
run b512_25000 --pairs 25000
run b512_12500 --pairs 12500
run b512_8192 --pairs 8192
run b512_4096 --pairs 4096
run b512_2048 --pairs 2048
GAMDP_NO_PAIR=1 run np_b512_12500 --pairs 12500
GAMDP_NO_PAIR=1 run np_b512_4096 --pairs 4096
GAMDP_NO_PAIR=1 run np_b512_2048 --pairs 2048
GAMDP_NO_PAIR=1 run np_b512_1024 --pairs 1024
