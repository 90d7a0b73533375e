#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# A/B timing of the band-150 throughput kernels on the GPU box (diagnostics build): eight tasks per wavefront (two quads,
# packed f16; default for >= 32 768 N-free tasks) against four (GAMDP_NO_PAIR=1); full / fill only / fill + strips
mkdir -p gpurun_out/ab150
D=$PWD/gam_ngs_amd/libgamdp_diag.so
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --band 150"
for mode in octo quad; do
 unset GAMDP_NO_PAIR
 if [ $mode = quad ]; then export GAMDP_NO_PAIR=1; fi
 for len in 50000 5000; do
  if [ $len = 50000 ]; then P=100000; X=""; else P=400000; X="GAMDP_QUAD_MIN=32768"; fi
  env $X GAMDP_LIB=$D $B --len $len --pairs $P > gpurun_out/ab150/${mode}_full_$len.log 2>&1
  env $X GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 $B --len $len --pairs $P > gpurun_out/ab150/${mode}_fill_$len.log 2>&1
  env $X GAMDP_LIB=$D GAMDP_DIAG_SKIP_TRACEBACK=1 GAMDP_DIAG_COUNT_MAT=1 $B --len $len --pairs $P > gpurun_out/ab150/${mode}_fillmat_$len.log 2>&1
 done
 GAMDP_LIB=$D GAMDP_DIAG_COUNT_MAT=1 python tools/count_materialise.py 150 65536 50000 > gpurun_out/ab150/${mode}_count.log 2>&1
 GAMDP_LIB=$D GAMDP_DIAG_COUNT_MAT=1 GAMDP_QUAD_MIN=1 python tools/count_materialise.py 150 64 50000 >> gpurun_out/ab150/${mode}_count.log 2>&1
done
unset GAMDP_NO_PAIR
for f in gpurun_out/ab150/*.log; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("   gcups %.0f kernel_ms %.1f ms_step %.1f"%(d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"]))
    elif 'band' in l: print("  ", l.rstrip()[:230])
PY
done
