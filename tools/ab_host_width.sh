#!/bin/bash
# A/B of the host pool's loop width on the driver-shaped batch (same box, one process per setting):  tools/ab_host_width.sh [outdir]
OUT=${1:-gpurun_out/ab_host_width}
mkdir -p $OUT
for rep in 1 2; do
for w in 16 32 48 8; do
  GAMDP_HOST_WIDTH=$w python3 tools/mixed150.py --steps 6 > $OUT/w${w}_$rep.json 2> $OUT/w${w}_$rep.err
done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/w*.json")):
    r = json.load(open(f))
    print(f.split("/")[-1], "%.2f ms per step, kernels %.2f, %d GCUPS" % (r["ms_per_step"], r["kernel_ms_per_step"], r["gcups"]))
PY
