#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# fill / fill + strips (diagnostics builds): tools/lab_r03/ab_fillmat.sh "<bench args>" libdiag1.so ...
ARGS="$1"; shift
mkdir -p gpurun_out/abfm
for lib in "$@"; do
 export GAMDP_LIB=$PWD/$lib; tag=$(basename $lib .so)
 for mode in fill fillmat full; do
  unset GAMDP_DIAG_SKIP_TRACEBACK GAMDP_DIAG_COUNT_MAT
  if [ $mode != full ]; then export GAMDP_DIAG_SKIP_TRACEBACK=1; fi
  if [ $mode = fillmat ]; then export GAMDP_DIAG_COUNT_MAT=1; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 $ARGS > gpurun_out/abfm/${tag}_$mode.log 2>&1
  python - gpurun_out/abfm/${tag}_$mode.log ${tag}_$mode <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("%-24s gcups %.0f kernel_ms %.1f ms_step %.1f"%(sys.argv[2], d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"]))
PY
 done
done
