#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# round 3, probe 9: what holds a lone direction-free band-150 wavefront back (timing only: the variants' results are unusable).
# Needs a tree built with -DGAMDP_DF5=1 and the variants `make -C gam_ngs_amd/csrc variant NAME=nostore FLAGS='-DGAMDP_DF5=1 -DGAMDP_EXP_DF_NOSTORE'`,
# NAME=noaread FLAGS='-DGAMDP_DF5=1 -DGAMDP_EXP_DF_NOAREAD', NAME=noboth with both.  Measured: 3.245 / 2.839 / 3.248 / 2.779 ms per 50 000 rows.
for lib in diag diag_nostore diag_noaread diag_noboth; do
  GAMDP_LIB=$PWD/gam_ngs_amd/libgamdp_$lib.so GAMDP_DIAG_SKIP_TRACEBACK=1 timeout -s KILL 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --no-proxy --band 150 --pairs 256 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib fill only: kernel_ms %.3f'%d['roofline']['kernel_ms_per_launch'])
"
done
