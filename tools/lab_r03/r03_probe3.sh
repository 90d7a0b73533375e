#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3, probe 3: where a wavefront's time goes at 1 and at 4 wavefronts per SIMD -- pipelined fill vs source order
mkdir -p gpurun_out/r03_probe3
for P in 2048 40960; do
for lib in diag diag_src; do
  D=$PWD/gam_ngs_amd/libgamdp_$lib.so
  B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --pairs $P"
  GAMDP_LIB=$D $B > gpurun_out/r03_probe3/${lib}_${P}_full.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 $B > gpurun_out/r03_probe3/${lib}_${P}_fill.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 GAMDP_DIAG_COUNT_MAT=1 $B > gpurun_out/r03_probe3/${lib}_${P}_fillmat.log 2>&1
done; done
for f in gpurun_out/r03_probe3/*.log; do python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("%-50s gcups %.0f kernel_ms %.2f ms_step %.2f"%(sys.argv[1].split('/')[-1], d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"]))
PY
done
