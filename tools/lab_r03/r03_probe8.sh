#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3, probe 8: a lone wavefront on a band-150 call (what the longest chain of a merge-block launch is): fill vs walk
mkdir -p gpurun_out/r03_probe8
D=$PWD/gam_ngs_amd/libgamdp_diag.so
for P in 256 1024; do
  B="timeout -s KILL 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --band 150 --pairs $P"
  GAMDP_LIB=$D $B > gpurun_out/r03_probe8/${P}_full.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 $B > gpurun_out/r03_probe8/${P}_fill.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_FORCE_N=1 $B > gpurun_out/r03_probe8/${P}_full_n.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_FORCE_N=1 GAMDP_DIAG_SKIP_TRACEBACK=1 $B > gpurun_out/r03_probe8/${P}_fill_n.log 2>&1
done
for f in gpurun_out/r03_probe8/*.log; do python - "$f" <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); ok=True; print("%-30s gcups %.0f kernel_ms %.3f ms_step %.3f %s"%(sys.argv[1].split('/')[-1], d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"], d["roofline"].get("kernel")))
if not ok: print(sys.argv[1], "no line:", open(sys.argv[1]).read()[-300:])
PY
done
