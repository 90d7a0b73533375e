#!/bin/bash
# (lab notes: how a number quoted in DESIGN.md was measured; run under gpurun from the repo root)
set -eu
: "${GRAFT_REPO_ROOT:?run under gpurun (GRAFT_REPO_ROOT = the repo copy on the GPU box)}"
# Band 150, eight tasks per wavefront: how much of the kernel is the tail of the last round?  100 000 tasks are 3.05 rounds
# of 8 x 4 096; 98 304 are exactly 3.  Optional $1 = another build of the library to compare (e.g. 2-lane strips).
mkdir -p gpurun_out/tail150
B="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-l1 --no-band150 --band 150 --len 50000"
for lib in "" "$1"; do
 tag=prod; if [ -n "$lib" ]; then tag=alt; export GAMDP_LIB=$PWD/$lib; fi
 for P in 32768 65536 98304 100000 114688 131072; do
  $B --pairs $P > gpurun_out/tail150/${tag}_$P.log 2>&1
 done
 if [ -z "$1" ]; then break; fi
done
unset GAMDP_LIB
for f in gpurun_out/tail150/*.log; do echo $f; python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print("   gcups %.0f kernel_ms %.1f ms_step %.1f"%(d["value"], d["roofline"]["kernel_ms_per_launch"], d["ms_per_step"]))
PY
done
