#!/bin/bash
# round 4 A/B: kernel time of variants of the diagnostics build, full / fill only / fill + the strips a walk down the middle asks for
#   tools/ab_r04.sh <out-tag> <band> <pairs> <lib-suffix>...      ("diag" = the plain diagnostics build)
set -u
: "${GRAFT_REPO_ROOT:?run under gpurun}"
TAG=$1; BAND=$2; PAIRS=$3; shift 3
OUT=gpurun_out/$TAG; mkdir -p $OUT
for lib in "$@"; do
  for mode in full fill fillmat; do
    unset GAMDP_DIAG_SKIP_TRACEBACK GAMDP_DIAG_COUNT_MAT
    [ $mode = fill ] && export GAMDP_DIAG_SKIP_TRACEBACK=1
    [ $mode = fillmat ] && export GAMDP_DIAG_SKIP_TRACEBACK=1 GAMDP_DIAG_COUNT_MAT=1
    L=gam_ngs_amd/libgamdp_$lib.so
    [ -f $L ] || { echo "missing $L"; continue; }
    GAMDP_LIB=$PWD/$L python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --band $BAND --pairs $PAIRS > $OUT/${lib}_$mode.log 2>&1
    python - $OUT/${lib}_$mode.log $lib $mode <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); ok=True
        print("%-12s %-8s kernel_ms %8.2f  gcups %7.0f"%(sys.argv[2], sys.argv[3], d["roofline"]["kernel_ms_per_launch"], d["value"]))
if not ok: print(sys.argv[2], sys.argv[3], "FAILED", open(sys.argv[1]).read()[-300:])
PY
  done
done
