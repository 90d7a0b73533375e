#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# A/B of library builds on one workload: tools/lab_r03/ab_libs.sh "<bench args>" lib1.so lib2.so ...   ("-" = the product library)
ARGS="$1"; shift
mkdir -p gpurun_out/ablibs
for lib in "$@"; do
 if [ "$lib" = "-" ]; then unset GAMDP_LIB; tag=prod; else export GAMDP_LIB=$PWD/$lib; tag=$(basename $lib .so); fi
 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 $ARGS > gpurun_out/ablibs/$tag.log 2>&1
 python - gpurun_out/ablibs/$tag.log $tag <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("%-20s gcups %.0f kernel_ms %.1f ms_step %.1f"%(sys.argv[2], d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"]))
PY
done
