#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# A/B timing of the band-512 kernels on the GPU box (diagnostics builds): two tasks per wavefront in packed f16 (default)
# against one task per wavefront (GAMDP_NO_PAIR=1); full / fill only / fill + strips.  libgamdp_diag_w5.so = the pair
# kernel compiled for 5 waves per SIMD (hand-built for this comparison; the host side still plans 4 per SIMD).
mkdir -p gpurun_out/ab512
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --pairs ${PAIRS:-40960}"
for mode in pair pair_w5 nopair; do
  D=$PWD/gam_ngs_amd/libgamdp_diag.so
  unset GAMDP_NO_PAIR
  if [ $mode = nopair ]; then export GAMDP_NO_PAIR=1; fi
  if [ $mode = pair_w5 ]; then D=$PWD/gam_ngs_amd/libgamdp_diag_w5.so; [ -f $D ] || continue; fi
  GAMDP_LIB=$D $B > gpurun_out/ab512/${mode}_full.log 2> gpurun_out/ab512/${mode}_full.err
  GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 $B > gpurun_out/ab512/${mode}_fill.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 GAMDP_DIAG_COUNT_MAT=1 $B > gpurun_out/ab512/${mode}_fillmat.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_COUNT_MAT=1 python tools/count_materialise.py 512 20480 50000 > gpurun_out/ab512/${mode}_count.log 2>&1
  GAMDP_LIB=$D GAMDP_DIAG_COUNT_MAT=1 python tools/count_materialise.py 512 64 50000 >> gpurun_out/ab512/${mode}_count.log 2>&1
done
unset GAMDP_NO_PAIR
for f in gpurun_out/ab512/*.log; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("   gcups %.0f kernel_ms %.1f ms_step %.1f"%(d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"]))
    elif 'band' in l: print("  ", l.rstrip()[:230])
PY
done
